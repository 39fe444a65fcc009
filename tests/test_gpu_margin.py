"""How much of the gates' margin the reduced-product arithmetic keeps when the weights are not one friendly seed.

The fp32 mode runs its k = 7 convs as Winograd F(2,7) (points 0, +-1, +-2, +-1/2, inf: ~5x the direct kernel's rounding
error) and its transposed convs as F(3,2); the bf16x3 mode drops the mid*mid term of every product; the bf16 mode rounds
every GEMM operand.  All three were gated on ONE Gaussian weight seed.  Here, at FULL size, through the C ABI, against
the fp32 CPU oracle (the checker):

  * three seeds of the default family and one of the "hostile" family (synth.make_weights(..., family="hostile"):
    heavy-tailed conv weights, SnakeBeta alpha / beta in [-1, 1.5], O(1) per-channel DC on the residual stream ahead of
    every k = 7 conv, per-channel gains over 1/8 .. 8, LayerScale / gamma up to 1), one 10 s utterance each (2 reference
    units: 64 frames + 61 padded with code 0, src/worker/vocoder.rs:60-85), in every mode;
  * the margin to each mode's gate is printed (pytest -s shows it; tools/prof_r03.sh keeps it under profiles/), and a
    family that loses more than 10x of the first seed's margin fails with the layer named: the per-stage taps of the
    hostile model are compared with HIPVOC_WINO=0's (the direct kernels) to say whether Winograd is what lost it;
  * one FULL-size per-stage comparison of the PADDED 61-frame unit at T = 64 (until now only T = 3 probes existed at this
    size): every stage within the stage tolerance.

Gates (BASELINE.json north_star): fp32 and bf16x3 max-abs <= 1e-3 and SNR >= 60 dB; bf16 (stated separately, DESIGN.md
section 7): max-abs <= 3e-2 and SNR >= 30 dB.
"""
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest

from conftest import snr_db

pytestmark = pytest.mark.gpu

GATES = {0: (1e-3, 60.0), 2: (1e-3, 60.0), 1: (3e-2, 30.0)}       # dtype -> (max-abs, SNR dB)
MODE = {0: "fp32 (Winograd)", 1: "bf16", 2: "bf16x3"}
STAGE_MAX_ABS = 1e-3
CASES = [("gauss", 0), ("gauss", 1), ("gauss", 2), ("hostile", 0)]


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
    """Every (family, seed) x mode once: max-abs error and SNR of the 10 s utterance against the fp32 oracle."""
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
        assert float((np.abs(ref) >= 1.0).mean()) < 1e-3, "the clamp would hide errors"
        for dtype in (0, 2, 1):
            v = pkg.HipVocoder.load(path, device=0, unit_frames=64, dtype=dtype)
            got = v.synthesize(codes)
            v.close()
            assert got.shape == ref.shape
            out[(fam, seed, dtype)] = (float(np.abs(got - ref).max()), float(snr_db(ref, got)), float(ref.std()))
        del o
        if (fam, seed) != ("hostile", 0):
            os.remove(path)
    return out


def test_margin_over_seeds_and_a_hostile_family(results):
    print()
    print(f"{'family':8s} {'seed':>4s} {'mode':16s} {'max-abs':>10s} {'gate/err':>9s} {'SNR dB':>8s} {'dB over gate':>12s} {'ref rms':>8s}")
    for (fam, seed, dtype), (err, snr, rms) in sorted(results.items(), key=lambda kv: (kv[0][2], kv[0][0], kv[0][1])):
        ga, gs = GATES[dtype]
        print(f"{fam:8s} {seed:4d} {MODE[dtype]:16s} {err:10.3e} {ga / err:9.1f} {snr:8.1f} {snr - gs:12.1f} {rms:8.3f}")
    for (fam, seed, dtype), (err, snr, rms) in results.items():
        ga, gs = GATES[dtype]
        assert err <= ga and snr >= gs, (fam, seed, MODE[dtype], err, snr)


def test_no_family_loses_more_than_10x_of_the_margin(results, pkg, synth, vo):
    """Margin = gate / max-abs error (and dB over the SNR gate).  If the hostile family or another seed keeps less than a
    tenth of seed 0's max-abs margin in a mode, name the first stage whose error grows and say whether the direct kernels
    (HIPVOC_WINO=0) restore it."""
    lost = []
    for dtype in (0, 2, 1):
        base_err, base_snr, _ = results[("gauss", 0, dtype)]
        for fam, seed in CASES[1:]:
            err, snr, _ = results[(fam, seed, dtype)]
            if err > 10.0 * base_err or snr < base_snr - 20.0:
                lost.append((fam, seed, dtype, err / base_err, base_snr - snr))
    if not lost:
        return
    report = [f"{fam} seed {seed} {MODE[d]}: max-abs error x{k:.1f}, SNR -{ds:.1f} dB against gauss seed 0" for fam, seed, d, k, ds in lost]
    # which layer: per-stage taps of the hostile model, Winograd against direct kernels (a child process: the switch is
    # read from the environment when the library loads a handle)
    probe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools", "margin_stages.py")
    for wino in ("1", "0"):
        r = subprocess.run([sys.executable, probe, "hostile", "0"], capture_output=True, text=True, timeout=900,
                           env=dict(os.environ, HIPVOC_WINO=wino))
        report.append(f"--- per-stage errors, HIPVOC_WINO={wino}\n{r.stdout[-3000:]}{r.stderr[-500:]}")
    pytest.fail("margin lost:\n" + "\n".join(report))


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
