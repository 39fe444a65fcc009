"""Pins the CPU oracle (oracle/voc_oracle.c) to the golden fixtures generated from the published
transformers building blocks (tests/golden/make_golden.py).  CPU-only."""
import numpy as np
import pytest

from conftest import load_golden, oracle_for, snr_db

UNIT_FIXTURES = ["small_T5", "small_T3_edge", "odd_T9", "full_T3"]
# fp32 vs fp32 with different summation order: 60+ layers, activations O(1)
STAGE_TOL = 1e-4
WAV_TOL = 2e-5


@pytest.mark.parametrize("name", UNIT_FIXTURES)
def test_unit_waveform_matches_golden(vo, synth, name):
    g = load_golden(name)
    o = oracle_for(vo, synth, str(g["cfg"]), int(g["wseed"]))
    wav = o.run_unit(g["codes"])
    cfg = getattr(synth, str(g["cfg"]))
    assert wav.size == cfg.out_len(g["codes"].shape[0]) == g["wav"].size
    assert np.abs(wav - g["wav"]).max() <= WAV_TOL
    assert snr_db(g["wav"], wav) >= 80.0
    assert np.abs(g["wav"]).max() < 1.0, "fixture must not sit on the clamp"


@pytest.mark.parametrize("name", ["small_T5", "odd_T9"])
def test_every_stage_matches_golden(vo, synth, name):
    g = load_golden(name)
    o = oracle_for(vo, synth, str(g["cfg"]), int(g["wseed"]))
    assert o.n_stages == int(g["n_stages"])
    for s in range(o.n_stages):
        a = o.stage(g["codes"], s)
        assert tuple(a.shape) == tuple(g[f"s{s}_shape"][-2:]), s
        got = a.reshape(-1)[g[f"s{s}_idx"]]
        assert np.abs(got - g[f"s{s}_val"]).max() <= STAGE_TOL, f"stage {s}"


def test_dequant_is_exact_against_golden(vo, synth):
    """Stage 0 (RVQ gather-sum + 1x1 projections).  The sums are exact; the projection differs from
    torch's conv only by summation order."""
    g = load_golden("small_T3_edge")
    o = oracle_for(vo, synth, str(g["cfg"]), int(g["wseed"]))
    a = o.stage(g["codes"], 0).reshape(-1)[g["s0_idx"]]
    assert np.abs(a - g["s0_val"]).max() <= 2e-6


@pytest.mark.parametrize("name", ["small_chunks_10x4", "small_chunks_8x4"])
def test_chunk_loop_matches_golden(vo, synth, name):
    """Vocoder::synthesize semantics (vocoder.rs:60-85): zero-code padding, truncate, concat."""
    g = load_golden(name)
    o = oracle_for(vo, synth, str(g["cfg"]), int(g["wseed"]))
    wav = o.synthesize(g["codes"], int(g["max_tokens"]))
    assert wav.size == g["wav"].size
    assert np.abs(wav - g["wav"]).max() <= WAV_TOL


def test_length_rule(vo, synth):
    """len = 1920*T - 555 for the (2,2)x(8,5,4,3) topology (SURVEY.md 8a length rule)."""
    o = oracle_for(vo, synth, "SMALL", 0)
    for T in (1, 2, 12, 61, 63, 64):
        assert o.out_len(T) == 1920 * T - 555
    assert o.out_len(64) == 122325


def test_chunk_independence_and_pad_influence(vo, synth):
    """Units are independent (frames 0..3 decode identically whatever follows) and the tail of a
    partial unit depends on the pad frames being code 0 (vocoder.rs:62,70)."""
    o = oracle_for(vo, synth, "SMALL", 0)
    cfg = synth.SMALL
    codes = synth.make_codes(7, 42, cfg)
    a = o.synthesize(codes[:4], 4)
    b = o.synthesize(codes, 4)
    assert np.array_equal(a, b[: a.size])
    tail = o.synthesize(codes[4:], 4)               # 3 frames padded with code 0
    assert np.array_equal(tail, b[a.size:])
    assert tail.size == 3 * 1920
    padded = np.zeros((4, cfg.num_quantizers), dtype=np.int64)
    padded[:3] = codes[4:]
    unit0 = o.run_unit(padded)
    padded[3] = 5                                   # a different pad code changes the kept tail
    unit5 = o.run_unit(padded)
    assert np.array_equal(unit0[: 3 * 1920], tail)
    assert not np.array_equal(unit0[: 3 * 1920], unit5[: 3 * 1920])
    # ... but only through the one-step lookahead of the trans-convs, never the early samples
    assert np.array_equal(unit0[:1920], unit5[:1920])


def test_pcm16_matches_golden(vo):
    g = load_golden("pcm16")
    assert np.array_equal(vo.pcm16(g["x"]), g["y"])
    assert vo.pcm16(np.array([np.nan], dtype=np.float32))[0] == 0      # Rust: NaN as i16 == 0


def test_codes_from_le_bytes(vo):
    """mod.rs:513-516: i64 little-endian."""
    codes = np.array([0, 1, 2047, 2 ** 40 + 3, -1], dtype=np.int64)
    assert np.array_equal(vo.codes_from_le(codes.astype("<i8").tobytes()), codes)


def test_out_of_range_code_is_an_error(vo, synth):
    o = oracle_for(vo, synth, "SMALL", 0)
    codes = synth.make_codes(2, 1, synth.SMALL)
    codes[1, 3] = synth.SMALL.codebook_size
    with pytest.raises(RuntimeError):
        o.run_unit(codes)


def test_transconv_trim_policies_match_transformers(vo):
    """Both trim policies of the decoder's transposed convs (SURVEY.md 8a asks for the switch: which one the shipped
    vocoder.onnx has cannot be verified offline) against transformers: Qwen3OmniMoeCausalTransConvNet (both sides, the
    default) and Mimi's causal MimiConvTranspose1d (right only).  Fixture: tools/gen_transconv_trim_golden.py."""
    g = load_golden("transconv_trim")
    for name in ("a", "b", "c"):
        x, w, b, s = g[f"{name}_x"], g[f"{name}_w"], g[f"{name}_b"], int(g[f"{name}_stride"])
        for policy, right in (("both", False), ("right", True)):
            y = vo.transconv(x, w, b, s, right)
            ref = g[f"{name}_{policy}"]
            assert y.shape == ref.shape, (name, policy, y.shape, ref.shape)
            assert ref.size == 0 or np.abs(y - ref).max() <= 1e-5, (name, policy)      # (one input, both sides: empty)
        L = x.shape[1]
        assert g[f"{name}_both"].shape[1] == (L - 1) * s and g[f"{name}_right"].shape[1] == L * s


def test_right_only_trim_gives_exactly_1920_samples_per_frame(vo, synth):
    cfg = synth.SMALL_TRIM_RIGHT
    o = vo.Oracle(synth.make_weights(cfg, 0), {k: float(v) for k, v in cfg.metadata().items()})
    for T in (1, 3, 4):
        assert o.out_len(T) == cfg.out_len(T) == 1920 * T
        assert o.run_unit(synth.make_codes(T, T, cfg)).size == 1920 * T
    # the chunk loop then keeps every sample of every unit (vocoder.rs:79-80: the .min() is a no-op)
    codes = synth.make_codes(10, 5, cfg)
    assert o.synthesize(codes, 4).size == 10 * 1920
