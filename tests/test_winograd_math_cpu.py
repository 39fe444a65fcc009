"""The minimal-filtering identities the fp32 Winograd kernels rely on (qwen3-tts_amd/csrc/kernels_wino.hip, hipvoc.cpp:
Loader::conv / Loader::transconv), checked in float64 with numpy: no GPU, no library.  The matrices below are the ones the
loader (G) and the kernels (B^T as the 24-instruction input transform, A^T as the output transform, the point split of
wino_ps_kernel) implement; a change there that breaks the algebra fails here first."""
import numpy as np

# F(2,7), points 0, +1, -1, +2, -2, +1/2, -1/2, inf -- hipvoc.cpp: Gm
G27 = np.array([
    [1, 0, 0, 0, 0, 0, 0],
    [-2 / 9] * 7,
    [-2 / 9, 2 / 9, -2 / 9, 2 / 9, -2 / 9, 2 / 9, -2 / 9],
    [1 / 90, 2 / 90, 4 / 90, 8 / 90, 16 / 90, 32 / 90, 64 / 90],
    [1 / 90, -2 / 90, 4 / 90, -8 / 90, 16 / 90, -32 / 90, 64 / 90],
    [64 / 90, 32 / 90, 16 / 90, 8 / 90, 4 / 90, 2 / 90, 1 / 90],
    [64 / 90, -32 / 90, 16 / 90, -8 / 90, 4 / 90, -2 / 90, 1 / 90],
    [0, 0, 0, 0, 0, 0, 1]], dtype=np.float64)


def input_transform_f27(d):
    """kernels_wino.hip: transform() -- the same 24 operations, in the same association."""
    d0, d1, d2, d3, d4, d5, d6, d7 = d
    e12 = -4.25 * d4 + (d2 + d6); o12 = -4.25 * d3 + (d1 + d5)
    e34 = 0.25 * d2 + (-1.25 * d4 + d6); o34 = 4.0 * d5 + (-5.0 * d3 + d1)          # 2 x odd part
    e56 = 4.0 * d2 + (-5.0 * d4 + d6); o56 = 0.25 * d5 + (-1.25 * d3 + d1)          # odd part / 2
    return np.array([5.25 * (d4 - d2) + (d0 - d6), e12 + o12, e12 - o12, 0.5 * o34 + e34, -0.5 * o34 + e34,
                     2.0 * o56 + e56, -2.0 * o56 + e56, 5.25 * (d3 - d5) + (d7 - d1)])


def output_transform_f27(m):
    y0 = ((m[1] + m[2]) + (m[3] + m[4])) + ((m[5] + m[6]) + m[0])
    y1 = (2.0 * (m[3] - m[4]) + (m[1] - m[2])) + (0.5 * (m[5] - m[6]) + m[7])
    return y0, y1


def test_f27_is_a_7_tap_filter_at_two_positions():
    rng = np.random.default_rng(0)
    for _ in range(50):
        g = rng.standard_normal(7)            # taps: y[p] = sum_j g[j] x[p - 6 + j]  (causal k = 7, dilation 1)
        d = rng.standard_normal(8)            # x[p - 6 .. p + 1]
        m = (G27 @ g) * input_transform_f27(d)
        y0, y1 = output_transform_f27(m)
        assert abs(y0 - g @ d[0:7]) < 1e-12 and abs(y1 - g @ d[1:8]) < 1e-12


def test_f27_point_split_partial_sums():
    """wino_ps_kernel: point group 0 holds M0 M1 M2 M7, group 1 M3 M4 M5 M6; each keeps one partial sum and sends the other."""
    rng = np.random.default_rng(1)
    m = rng.standard_normal(8)
    y0, y1 = output_transform_f27(m)
    keep0, send0 = (m[1] + m[2]) + m[0], (m[1] - m[2]) + m[7]
    keep1, send1 = 2.0 * (m[3] - m[4]) + 0.5 * (m[5] - m[6]), (m[3] + m[4]) + (m[5] + m[6])
    assert abs((keep0 + send1) - y0) < 1e-12 and abs((keep1 + send0) - y1) < 1e-12


def test_f27_right_edge_needs_no_mask():
    """The last input (d7 = x[p + D]) reaches V7 only and M7 reaches y[p + D] only: garbage there cannot touch a stored y[p]."""
    rng = np.random.default_rng(2)
    g, d = rng.standard_normal(7), rng.standard_normal(8)
    d_bad = d.copy(); d_bad[7] = 1e30
    v, v_bad = input_transform_f27(d), input_transform_f27(d_bad)
    assert np.array_equal(v[:7], v_bad[:7])
    y0, _ = output_transform_f27((G27 @ g) * v)
    y0_bad, _ = output_transform_f27((G27 @ g) * v_bad)
    assert y0 == y0_bad


def test_f32_two_tap_filter_at_three_positions():
    """F(3,2) of the transposed convs: y[q] = g0 x[q] + g1 x[q + 1] for three consecutive q from four products."""
    rng = np.random.default_rng(3)
    for _ in range(50):
        g0, g1 = rng.standard_normal(2)
        d = rng.standard_normal(4)
        u = np.array([g0, (g0 + g1) / 2, (g0 - g1) / 2, g1])
        v = np.array([d[0] - d[2], d[1] + d[2], d[2] - d[1], d[3] - d[1]])
        m = u * v
        y = [(m[1] + m[2]) + m[0], m[1] - m[2], (m[1] + m[2]) + m[3]]
        for q in range(3):
            assert abs(y[q] - (g0 * d[q] + g1 * d[q + 1])) < 1e-12


def test_f27_tile_numbering_covers_every_position_once():
    """Tile T of a unit is the output pair (p, p + D), p = 2 D (T / D) + T % D: every position exactly once, for every dilation."""
    for D in (1, 3, 9):
        n_tiles = 6 * D
        seen = []
        for T in range(n_tiles):
            p = 2 * D * (T // D) + T % D
            seen += [p, p + D]
        assert sorted(seen) == list(range(2 * n_tiles))
