"""GPU tests of the split-operand mode (HIPVOC_DTYPE_BF16X3) through the C ABI.

Every conv-GEMM operand is carried as two bf16 planes (hi = bf16(v), mid = bf16(v - hi): 16 mantissa bits) and
every product is three bf16 MFMAs (hi*hi + hi*mid + mid*hi) accumulated in fp32; everything else is the fp32 mode.
The mode is held to the FP32 gates of BASELINE.json's north_star, not to a tolerance of its own:
waveform max-abs <= 1e-3 and SNR >= 60 dB against the fp32 CPU oracle, every stage within the fp32 stage tolerance,
RVQ dequantisation bit-exact.  (Measured on the seeded full model: 3.2e-5 / 88 dB; the fp32 mode: 5.9e-6 / 104 dB.)
"""
import os
import tempfile

import numpy as np
import pytest

from conftest import oracle_for, snr_db, weights_for

pytestmark = pytest.mark.gpu

BF16X3 = 2
WAV_MAX_ABS = 1e-3          # the fp32 gates
WAV_SNR_DB = 60.0
STAGE_MAX_ABS = 1e-3


@pytest.fixture(scope="module", autouse=True)
def _torch_hip_first():
    import torch
    if torch.cuda.is_available():
        torch.cuda.init()
    yield


_VOC = {}


def voc_x3(pkg, synth, cfg_name, unit):
    key = (cfg_name, unit)
    if key not in _VOC:
        cfg = getattr(synth, cfg_name)
        path = os.path.join(tempfile.gettempdir(), f"hipvoc_test_{cfg_name}_0.safetensors")
        if not os.path.exists(path):
            synth.save_safetensors(path, weights_for(synth, cfg_name, 0), cfg.metadata())
        _VOC[key] = pkg.HipVocoder.load(path, device=0, unit_frames=unit, dtype=BF16X3)
    return _VOC[key]


@pytest.mark.parametrize("cfg_name,T", [("SMALL", 8), ("ODD", 9), ("SMALL", 5), ("FULL", 3)])
def test_x3_every_stage_within_the_fp32_stage_tolerance(pkg, synth, vo, cfg_name, T):
    cfg = getattr(synth, cfg_name)
    v = voc_x3(pkg, synth, cfg_name, T)
    o = oracle_for(vo, synth, cfg_name, 0)
    codes = synth.make_codes(T, 21 + T, cfg)
    assert v.n_stages == o.n_stages
    for s in (1000, 0):          # raw dequant sums, projected dequant output: bit-exact in every mode
        ref = o.stage(codes, s)
        got = v.stage(codes, s, ref.size + 8)
        assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), s
    used = set()
    for s in range(1, v.n_stages):
        ref = o.stage(codes, s)
        got = v.stage(codes, s, ref.size + 8)
        assert got.shape == ref.shape, s
        scale = max(1.0, float(np.abs(ref).max()))
        assert np.abs(got - ref).max() <= STAGE_MAX_ABS * scale, (s, float(np.abs(got - ref).max()), scale)
    v.set_profile(True)
    v.synthesize(codes)
    used = {k.split("|")[1].split("<")[0] for k in v.last_profile()}
    v.set_profile(False)
    assert "conv_gemm_bf16x3_kernel" in used and "conv_gemm_bf16_kernel" not in used and "conv_gemm_kernel" not in used, used


def test_x3_full_utterance_meets_the_fp32_gates(pkg, synth, vo):
    """BASELINE configs[1]'s utterance (125 frames -> units of 64 + 61, full model) against the fp32 oracle, fp32 gates."""
    cfg = synth.FULL
    v = voc_x3(pkg, synth, "FULL", 64)
    o = oracle_for(vo, synth, "FULL", 0)
    codes = synth.make_codes(125, 1000, cfg)
    got = v.synthesize(codes)
    ref = o.synthesize(codes, 64)
    assert got.size == ref.size == 122325 + 61 * 1920
    err = np.abs(got - ref).max()
    snr = snr_db(ref, got)
    assert err <= WAV_MAX_ABS, err
    assert snr >= WAV_SNR_DB, snr
    assert snr >= 80.0, f"the split should leave ~2^-16 per product, measured 88 dB: {snr}"


def test_x3_ragged_batch_bitwise_and_deterministic(pkg, synth, vo):
    cfg = synth.SMALL
    v = voc_x3(pkg, synth, "SMALL", 4)
    o = oracle_for(vo, synth, "SMALL", 0)
    seqs = [synth.make_codes(n, 50 + n, cfg) for n in (1, 4, 5, 10, 3)]
    outs = v.synthesize_batch(seqs)
    for c, got in zip(seqs, outs):
        one = v.synthesize(c)
        assert np.array_equal(one.view(np.uint32), got.view(np.uint32))
        ref = o.synthesize(c, 4)
        assert got.size == ref.size
        assert np.abs(got - ref).max() <= WAV_MAX_ABS and snr_db(ref, got) >= WAV_SNR_DB
    assert v.synthesize(np.zeros((0, 16), dtype=np.int64)).size == 0
    vf = voc_x3(pkg, synth, "FULL", 64)
    codes = synth.make_codes(64, 5, synth.FULL)
    a, b = vf.run(codes), vf.run(codes)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


def test_x3_full_size_unit_independence(pkg, synth):
    """Units decode identically alone, in a batch, and in any group position (size-independent property, full model)."""
    cfg = synth.FULL
    v = voc_x3(pkg, synth, "FULL", 64)
    stream = synth.make_codes(64 * 5 + 12, 4242, cfg)
    whole = v.synthesize(stream)
    assert whole.size == 5 * 122325 + 12 * 1920 and np.all(np.abs(whole) <= 1.0)
    alone = v.synthesize(stream[3 * 64: 4 * 64])
    assert np.array_equal(alone, whole[3 * 122325: 4 * 122325])
    utts = [synth.make_codes(125, 1000 + u, cfg) for u in range(3)]
    for got, c in zip(v.synthesize_batch(utts), utts):
        assert np.array_equal(got, v.synthesize(c))


@pytest.mark.parametrize("cfg_name,unit,frames", [("FULL", 64, 125), ("ODD", 9, 20), ("SMALL", 8, 13)])
def test_x3_small_launch_kernels_bitwise(pkg, synth, cfg_name, unit, frames, monkeypatch):
    """The small-launch forms (gemm1_bf16x3_kernel, convs_bf16x3_kernel) make conv_gemm_bf16x3_kernel's MFMA chains:
    forced on everywhere vs off, the waveform is the same bit for bit."""
    cfg = getattr(synth, cfg_name)
    path = os.path.join(tempfile.gettempdir(), f"hipvoc_test_{cfg_name}_0.safetensors")
    if not os.path.exists(path):
        synth.save_safetensors(path, weights_for(synth, cfg_name, 0), cfg.metadata())
    codes = synth.make_codes(frames, 31, cfg)
    outs = []
    for wgs in ("0", "1000000"):
        monkeypatch.setenv("HIPVOC_GEMM1_WGS", wgs)
        monkeypatch.setenv("HIPVOC_CONVS_WGS", wgs)
        v = pkg.HipVocoder.load(path, device=0, unit_frames=unit, dtype=BF16X3)
        v.set_profile(True)
        outs.append(v.synthesize(codes))
        for name in ("gemm1_bf16x3_kernel", "convs_bf16x3_kernel")[: 2 if cfg_name == "FULL" else 1]:
            used = any(name in k for k in v.last_profile())
            assert used == (wgs != "0"), (wgs, name, sorted(v.last_profile()))
        del v
    assert outs[0].size and np.array_equal(outs[0].view(np.uint32), outs[1].view(np.uint32))
