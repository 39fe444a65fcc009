"""The encoder oracle (oracle/enc_oracle.c, restating /root/reference/src/speech_tokenizer.rs) against:
  * golden vectors from an independent implementation of the upstream architecture (transformers' MimiModel with
    the same seeded weights; tools/gen_encoder_golden.py, run in the build container only) -- the Rust reference
    itself cannot be built here (no cargo), so against IT parity stays unpinned;
  * known-answer and property tests of every piece the reference spells out (ELU, causal / strided padding and
    frame count, the sliding window, first-minimum nearest-codeword search, the 16-codebook cut, resampling).
"""
import numpy as np
import pytest

from conftest import enc_oracle_for, enc_weights_for, load_golden


def stage_ids(cfg):
    nb = len(cfg.ratios)
    return {"seanet": 1 + 2 * nb, "transformer": 1 + 2 * nb + cfg.num_layers, "latent": 2 + 2 * nb + cfg.num_layers,
            "proj_sem": 3 + 2 * nb + cfg.num_layers, "proj_ac": 4 + 2 * nb + cfg.num_layers}


def test_oracle_matches_upstream_mimi_golden(eo, synth):
    g = load_golden("encoder_mimi_small")
    cfg = synth.ENC_SMALL
    o = enc_oracle_for(eo, synth, "ENC_SMALL", 0)
    ids = stage_ids(cfg)
    for name in ("seanet", "transformer", "latent"):
        got = o.stage(g["audio"], ids[name])
        assert got.shape == g[name].shape
        assert np.abs(got - g[name]).max() <= 1e-4 * max(1.0, np.abs(g[name]).max()), name
    codes = o.encode(g["audio"])
    assert codes.dtype == np.int64 and np.array_equal(codes, g["codes"])


@pytest.mark.parametrize("cfg_name", ["ENC_SMALL", "ENC_ODD"])
def test_frame_count_floors_at_every_strided_conv(eo, synth, cfg_name):
    cfg = getattr(synth, cfg_name)
    o = enc_oracle_for(eo, synth, cfg_name, 0)
    for n in (1920, 1921, 3839, 3840, 5000, 1920 * 5 + 1919):
        assert o.n_frames(n) == cfg.n_frames(n) == n // 1920
        assert o.encode(synth.make_audio(n, n)).shape == (n // 1920, min(16, cfg.n_semantic + cfg.n_acoustic))
    with pytest.raises(RuntimeError):
        o.encode(synth.make_audio(1919, 0))          # less than one frame: the last strided conv has no window


def test_codebook_cut_is_16_or_all(eo, synth):
    assert enc_oracle_for(eo, synth, "ENC_SMALL", 0).use_q == 16      # 1 + 17 available (speech_tokenizer.rs:664-667)
    assert enc_oracle_for(eo, synth, "ENC_ODD", 0).use_q == 7         # 1 + 6 available


def test_encoder_is_causal(eo, synth):
    """Left-only padding everywhere and a causal attention mask: frames of a prefix do not change when audio is
    appended (what makes unit-wise / streaming encoding possible)."""
    o = enc_oracle_for(eo, synth, "ENC_SMALL", 0)
    a = synth.make_audio(1920 * 9, 4)
    full = o.encode(a)
    assert np.array_equal(o.encode(a[: 1920 * 5]), full[:5])
    lat = stage_ids(synth.ENC_SMALL)["latent"]
    assert np.array_equal(o.stage(a[: 1920 * 5 + 777], lat), o.stage(a, lat)[:, :5])


def test_sliding_window_binds(eo, synth):
    """context=5 in ENC_SMALL: transformer frame t must not see frames <= t - 5 (speech_tokenizer.rs:346-362)."""
    cfg = synth.ENC_SMALL
    o = enc_oracle_for(eo, synth, "ENC_SMALL", 0)
    W = dict(enc_weights_for(synth, "ENC_SMALL", 0))
    wide = eo.EncOracle(W, {**{k: float(v) for k, v in cfg.metadata().items()}, "context": 1000.0})
    a = synth.make_audio(1920 * 8, 6)
    st = stage_ids(cfg)["transformer"]
    x, y = o.stage(a, st), wide.stage(a, st)
    assert np.array_equal(x[:, :5], y[:, :5])          # the first 5 positions see everything either way
    assert np.abs(x[:, 5:] - y[:, 5:]).max() > 1e-4


def test_nearest_codeword_is_first_minimum_and_truly_nearest(eo, synth):
    cfg = synth.ENC_SMALL
    W = dict(enc_weights_for(synth, "ENC_SMALL", 0))
    # duplicate codewords: rows 9 and 30 of the semantic codebook become copies of row 4
    p = "encoder.quantizer.semantic_residual_vector_quantizer.layers.0.codebook."
    es, us = W[p + "embed_sum"].copy(), W[p + "cluster_usage"].copy()
    for r in (9, 30):
        es[r] = es[4]; us[r] = us[4]
    W[p + "embed_sum"], W[p + "cluster_usage"] = es, us
    o = eo.EncOracle(W, {k: float(v) for k, v in cfg.metadata().items()})
    a = synth.make_audio(1920 * 40, 8)
    ids = stage_ids(cfg)
    codes = o.encode(a)
    assert not np.isin(codes[:, 0], (9, 30)).any()       # never the later duplicates
    # every decision is the true nearest neighbour (float64 brute force), up to ties within rounding
    for group, col0, n_layers, st in ((0, 0, 1, ids["proj_sem"]), (1, 1, 15, ids["proj_ac"])):
        res = o.stage(a, st).astype(np.float64).T            # [T, cb_dim]
        for q in range(n_layers):
            emb = o.embed(group, q).astype(np.float64)
            d = ((res[:, None, :] - emb[None]) ** 2).sum(-1)
            pick = codes[:, col0 + q]
            gap = d[np.arange(len(pick)), pick] - d.min(axis=1)
            assert gap.max() <= 1e-4, (group, q, gap.max())
            res = res - emb[pick]
    # quantize() on the latent reproduces encode()
    assert np.array_equal(o.quantize(o.stage(a, ids["latent"])), codes)
    # distances() is what the search minimises
    x = o.stage(a, ids["proj_sem"])[:, 3]
    assert int(np.argmin(o.distances(0, 0, x))) == codes[3, 0]


def test_elu_and_padding_known_answers(eo, synth):
    """A hand-built one-block encoder whose stages can be written down: initial conv = identity tap, residual branch
    zeroed, down conv = sum of its window (so ELU, the left pad of k - s and the floor show up exactly)."""
    cfg = synth.EncoderConfig(n_filters=2, ratios=(4,), dimension=2, num_layers=0, num_heads=1, dim_feedforward=4,
                              codebook_size=4, codebook_dim=2, n_acoustic=1, downsample_stride=2)
    W = {k: np.zeros_like(v) for k, v in synth.make_encoder_weights(cfg, 0).items()}
    e = "encoder.encoder.layers"
    W[f"{e}.0.conv.weight"][0, 0, 6] = 1.0           # channel 0 = x (the last tap is the current sample)
    W[f"{e}.0.conv.weight"][1, 0, 6] = -1.0          # channel 1 = -x
    W[f"{e}.3.conv.weight"][0, 0, :] = 1.0           # down conv (k=8, s=4): window sum of elu(channel 0)
    W[f"{e}.3.conv.weight"][1, 1, :] = 1.0
    for k in W:
        if k.endswith("cluster_usage"):
            W[k][:] = 1.0
    o = eo.EncOracle(W, {k: float(v) for k, v in cfg.metadata().items()})
    x = np.array([0.5, -1.0, 2.0, -0.25, 1.0, 1.0, -3.0, 0.0, 4.0, 9.0], dtype=np.float32)   # 10 samples -> 2 windows
    s0 = o.stage(x, 0)
    assert np.array_equal(s0[0], x) and np.array_equal(s0[1], -x)
    assert np.array_equal(o.stage(x, 1), s0)         # zero residual branch
    elu = lambda v: np.where(v > 0, v, np.expm1(np.minimum(v, 0))).astype(np.float32)
    d = o.stage(x, 2)
    assert d.shape == (4, 2)                         # floor(10 / 4) outputs, 2 * C channels
    pad = np.concatenate([np.zeros(4, np.float32), elu(x)])      # left pad k - s = 4
    want = np.array([pad[0:8].sum(dtype=np.float64), pad[4:12].sum(dtype=np.float64)])
    assert np.allclose(d[0], want, rtol=0, atol=1e-6)
    padn = np.concatenate([np.zeros(4, np.float32), elu(-x)])
    assert np.allclose(d[1], [padn[0:8].sum(dtype=np.float64), padn[4:12].sum(dtype=np.float64)], atol=1e-6)


def test_resample_matches_the_reference_formula(eo):
    rng = np.random.default_rng(0)
    x = rng.standard_normal(1000).astype(np.float32)
    for fr, to in ((16000, 24000), (44100, 24000), (48000, 24000), (24000, 24000), (8000, 24000)):
        got = eo.resample(x, fr, to)
        ratio = fr / to
        n = int(len(x) / ratio)
        i = np.arange(n, dtype=np.float64) * ratio
        idx = i.astype(np.int64)
        frac = i - idx
        nxt = np.minimum(idx + 1, len(x) - 1)
        want = np.where(idx + 1 < len(x), x[idx].astype(np.float64) * (1 - frac) + x[nxt].astype(np.float64) * frac,
                        x[np.minimum(idx, len(x) - 1)].astype(np.float64)).astype(np.float32)
        assert got.shape == want.shape and np.array_equal(got, want), (fr, to)
