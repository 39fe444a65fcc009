"""How much of the gates' margin the reduced-product arithmetic keeps when the weights are not one friendly seed.

The fp32 mode runs its k = 7 convs as Winograd F(2,7) (points 0, +-1, +-2, +-1/2, inf: ~5x the direct kernel's rounding
error) and its transposed convs as F(3,2); the bf16x3 mode drops the mid*mid term of every product; the bf16 mode rounds
every GEMM operand.  All three were gated on ONE Gaussian weight seed.  Here, at FULL size, through the C ABI, against
the fp32 CPU oracle (the checker), one 10 s utterance each (2 reference units: 64 frames + 61 padded with code 0,
src/worker/vocoder.rs:60-85), in every mode:

  * three seeds of the default family and one of the "stress" family (synth.FAMILY_KNOBS: heavy-tailed conv weights,
    SnakeBeta alpha / beta in [-1, 0.9], O(1) per-channel DC on the residual stream ahead of every k = 7 conv,
    per-channel gains over 1/8 .. 8, LayerScale / gamma up to 1): the ABSOLUTE gates hold (fp32 and bf16x3: 1e-3 /
    60 dB; the bf16 mode's stated tolerance is a statement about the default family -- on the other families its figures
    are printed, not gated: its error scales with the network's conditioning, below);
  * the "hostile" family (the same with alpha / beta up to 1.5, i.e. e^alpha up to 4.5), which VERDICT r2 asked for, turns
    out to be an ILL-CONDITIONED NETWORK rather than a test of the arithmetic: perturbing its weights by half an fp32 ulp
    moves the oracle's own waveform by ~1e-3 / 60 dB (measured below as the `floor`; default family: 4e-6 / 104 dB), the
    error grows 4-5x per decoder block with the direct kernels as with Winograd (profiles/r03_margin_stages_*.txt).  No
    fp32 implementation holds 1e-3 / 60 dB against another there, so it is judged against what the direct kernels
    (HIPVOC_WINO=0) do on the same weights: Winograd may cost at most 3x their error;
  * the margin to each mode's gate is printed (pytest -s; tools/r03_*.sh keep it under profiles/).

Gates (BASELINE.json north_star): fp32 and bf16x3 max-abs <= 1e-3 and SNR >= 60 dB; bf16 (stated separately, DESIGN.md
section 7): max-abs <= 3e-2 and SNR >= 30 dB.
"""
import os
import tempfile

import numpy as np
import pytest

from conftest import snr_db

pytestmark = pytest.mark.gpu

GATES = {0: (1e-3, 60.0), 2: (1e-3, 60.0), 1: (3e-2, 30.0)}       # dtype -> (max-abs, SNR dB)
MODE = {0: "fp32 (Winograd)", 1: "bf16", 2: "bf16x3", 10: "fp32 (direct)"}
STAGE_MAX_ABS = 1e-3
CASES = [("gauss", 0), ("gauss", 1), ("gauss", 2), ("stress", 0), ("hostile", 0)]


@pytest.fixture(scope="module", autouse=True)
def _torch_hip_first():
    import torch
    if torch.cuda.is_available():
        torch.cuda.init()
    yield


def _utterance_units(synth, cfg, seed):
    c = synth.make_codes(125, seed, cfg)
    u = np.zeros((2, 64, 16), dtype=np.int64)
    u[0] = c[:64]
    u[1, :61] = c[64:]
    return c, u


@pytest.fixture(scope="module")
def results(pkg, synth, vo):
    """Every (family, seed) x mode once: max-abs error and SNR of the 10 s utterance against the fp32 oracle; for the
    non-default families also the direct fp32 kernels (key 10) and the network's own half-ulp sensitivity (key "floor")."""
    cfg = synth.FULL
    md = {k: float(v) for k, v in cfg.metadata().items()}
    out = {}
    for fam, seed in CASES:
        W = synth.make_weights(cfg, seed, fam)
        path = os.path.join(tempfile.gettempdir(), f"hipvoc_margin_{fam}_{seed}.safetensors")
        synth.save_safetensors(path, W, cfg.metadata())
        codes, units = _utterance_units(synth, cfg, 1000 + seed)
        o = vo.Oracle(W, md)
        ref = o.synthesize(codes, 64)
        del o
        assert float((np.abs(ref) >= 1.0).mean()) < 1e-3, "the clamp would hide errors"
        modes = [0, 2, 1] + ([10] if fam != "gauss" else [])
        for dtype in modes:
            if dtype == 10:
                os.environ["HIPVOC_WINO"] = "0"          # read by hipvoc_load, per handle
            try:
                v = pkg.HipVocoder.load(path, device=0, unit_frames=64, dtype=dtype % 10)
            finally:
                os.environ.pop("HIPVOC_WINO", None)
            got = v.synthesize(codes)
            v.close()
            assert got.shape == ref.shape
            out[(fam, seed, dtype)] = (float(np.abs(got - ref).max()), float(snr_db(ref, got)), float(ref.std()))
        if fam != "gauss":
            # the network's own sensitivity: the oracle again, every weight moved by at most half an fp32 ulp
            rng = np.random.default_rng(5)
            W2 = {k: (a * (1 + rng.uniform(-6e-8, 6e-8, size=a.shape).astype(np.float32))).astype(np.float32) for k, a in W.items()}
            o2 = vo.Oracle(W2, md)
            ref2 = o2.synthesize(codes, 64)
            del o2, W2
            out[(fam, seed, "floor")] = (float(np.abs(ref2 - ref).max()), float(snr_db(ref, ref2)), float(ref.std()))
        os.remove(path)
    return out


def test_margin_over_seeds_and_families(results):
    print()
    print(f"{'family':8s} {'seed':>4s} {'mode':22s} {'max-abs':>10s} {'gate/err':>9s} {'SNR dB':>8s} {'dB over gate':>12s} {'ref rms':>8s}")
    order = {0: 0, 10: 1, 2: 2, 1: 3, "floor": 4}
    for (fam, seed, dtype), (err, snr, rms) in sorted(results.items(), key=lambda kv: (CASES.index((kv[0][0], kv[0][1])), order[kv[0][2]])):
        if dtype == "floor":
            print(f"{fam:8s} {seed:4d} {'oracle, weights +-ulp/2':22s} {err:10.3e} {'':9s} {snr:8.1f}")
            continue
        ga, gs = GATES[dtype % 10]
        print(f"{fam:8s} {seed:4d} {MODE[dtype]:22s} {err:10.3e} {ga / err:9.1f} {snr:8.1f} {snr - gs:12.1f} {rms:8.3f}")
    for (fam, seed, dtype), (err, snr, rms) in results.items():
        if dtype == "floor" or fam == "hostile":
            continue
        if dtype == 1 and fam != "gauss":
            continue                                    # the bf16 tolerance is stated for the default family (docstring)
        ga, gs = GATES[dtype % 10]
        assert err <= ga and snr >= gs, (fam, seed, MODE[dtype], err, snr)


def test_winograd_costs_at_most_3x_the_direct_kernels_where_the_network_is_ill_conditioned(results):
    """On the stress and hostile families the Winograd forms may lose at most 3x in max-abs error and 10 dB against the
    direct kernels on the same weights; and the hostile network's own half-ulp sensitivity is reported beside them."""
    for fam in ("stress", "hostile"):
        w_err, w_snr, _ = results[(fam, 0, 0)]
        d_err, d_snr, _ = results[(fam, 0, 10)]
        f_err, f_snr, _ = results[(fam, 0, "floor")]
        print(f"\n{fam}: Winograd {w_err:.3e} / {w_snr:.1f} dB, direct {d_err:.3e} / {d_snr:.1f} dB, network floor {f_err:.3e} / {f_snr:.1f} dB")
        assert w_err <= 3.0 * d_err and w_snr >= d_snr - 10.0, (fam, w_err, d_err, w_snr, d_snr)
    # the hostile family is ill-conditioned by itself: its floor is within a factor of a few of the gate
    assert results[("hostile", 0, "floor")][0] > 1e-4


def test_full_size_padded_unit_stage_by_stage(pkg, synth, vo):
    """The 61-frame unit of a 10 s utterance as the reference decodes it -- 64 frames, the last three code 0
    (vocoder.rs:62,70) -- at FULL size, every stage against the oracle (fp32 mode)."""
    cfg = synth.FULL
    W = synth.make_weights(cfg, 0)
    path = os.path.join(tempfile.gettempdir(), "hipvoc_test_FULL_0.safetensors")
    if not os.path.exists(path):
        synth.save_safetensors(path, W, cfg.metadata())
    v = pkg.HipVocoder.load(path, device=0, unit_frames=64)
    o = vo.Oracle(W, {k: float(v_) for k, v_ in cfg.metadata().items()})
    _, units = _utterance_units(synth, cfg, 1000)
    codes = units[1]
    assert (codes[61:] == 0).all() and codes[:61].any()
    assert v.n_stages == o.n_stages
    worst = (0.0, -1)
    for s in range(v.n_stages):
        ref = o.stage(codes, s)
        got = v.stage(codes, s, ref.size + 8)
        assert got.shape == ref.shape, s
        e = float(np.abs(got - ref).max())
        worst = max(worst, (e, s))
        assert e <= STAGE_MAX_ABS, f"stage {s}: {e}"
    print(f"\nFULL T=64 padded unit: worst stage error {worst[0]:.2e} at stage {worst[1]} of {v.n_stages}")
    v.close()
