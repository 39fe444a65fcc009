"""GPU tests of the bf16 conv-GEMM mode (HIPVOC_DTYPE_BF16, BASELINE configs[4]) through the C ABI.

The mode rounds the operands of every conv / linear / transposed conv to bf16 (nearest even) and accumulates
in fp32; RVQ, norms, attention, SnakeBeta, residual adds and the final C->1 conv stay fp32.  Two checkers:
  * the oracle in its bf16-emulating mode (oracle/voc_oracle.c, bf16_gemm=1: same rounding points, fp32
    products and sums): pins the arithmetic -- the first GEMM stages agree to fp32 round-off, later stages only
    differ through rounding-boundary flips (a 1e-7 difference before a bf16 rounding becomes a 4e-3 one after);
  * the fp32 oracle: the STATED TOLERANCE of the mode, separate from the fp32 gates (1e-3 / 60 dB):
        waveform SNR >= 30 dB and max-abs error <= 3e-2 of full scale
    (measured on the seeded random-weight full model: 35.9 dB / 1.2e-2; SURVEY.md 8d proposed 35 dB / 2e-2 "to be
    fixed after first measurement").  Parity vs the shipped ONNX graph stays unpinned.
"""
import os
import tempfile

import numpy as np
import pytest

from conftest import oracle_for, snr_db, weights_for

pytestmark = pytest.mark.gpu

BF16 = 1
WAV_SNR_DB = 30.0
WAV_MAX_ABS = 3e-2


@pytest.fixture(scope="module", autouse=True)
def _torch_hip_first():
    import torch
    if torch.cuda.is_available():
        torch.cuda.init()
    yield


_VOC = {}
_OB = {}


def voc_bf16(pkg, synth, cfg_name, unit):
    key = (cfg_name, unit)
    if key not in _VOC:
        cfg = getattr(synth, cfg_name)
        path = os.path.join(tempfile.gettempdir(), f"hipvoc_test_{cfg_name}_0.safetensors")
        if not os.path.exists(path):
            synth.save_safetensors(path, weights_for(synth, cfg_name, 0), cfg.metadata())
        _VOC[key] = pkg.HipVocoder.load(path, device=0, unit_frames=unit, dtype=BF16)
    return _VOC[key]


def oracle_bf16(vo, synth, cfg_name):
    if cfg_name not in _OB:
        cfg = getattr(synth, cfg_name)
        md = {k: float(v) for k, v in cfg.metadata().items()}
        md["bf16_gemm"] = 1.0
        _OB[cfg_name] = vo.Oracle(weights_for(synth, cfg_name, 0), md)
    return _OB[cfg_name]


@pytest.mark.parametrize("cfg_name,T", [("SMALL", 8), ("ODD", 9), ("FULL", 3)])
def test_bf16_dequant_stays_bit_exact_and_first_gemms_match_emulation(pkg, synth, vo, cfg_name, T):
    cfg = getattr(synth, cfg_name)
    v = voc_bf16(pkg, synth, cfg_name, T)
    ob = oracle_bf16(vo, synth, cfg_name)
    codes = synth.make_codes(T, 21 + T, cfg)
    for s in (1000, 0):          # raw dequant sums, projected dequant output: fp32 in every mode
        ref = ob.stage(codes, s)
        got = v.stage(codes, s, ref.size + 8)
        assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), s
    ref = ob.stage(codes, 1)     # pre_conv: the first bf16 GEMM, identical inputs on both sides
    got = v.stage(codes, 1, ref.size + 8)
    assert np.abs(got - ref).max() <= 1e-5


@pytest.mark.parametrize("cfg_name,T", [("SMALL", 8), ("ODD", 9), ("FULL", 3)])
def test_bf16_every_stage_tracks_the_emulation(pkg, synth, vo, cfg_name, T):
    """Every stage is at least as close to the fp32 oracle as the bf16-emulating oracle is (3 dB slack), and
    within 30 dB of the emulation itself (flips only)."""
    cfg = getattr(synth, cfg_name)
    v = voc_bf16(pkg, synth, cfg_name, T)
    ob = oracle_bf16(vo, synth, cfg_name)
    o32 = oracle_for(vo, synth, cfg_name, 0)
    codes = synth.make_codes(T, 21 + T, cfg)
    assert v.n_stages == ob.n_stages
    for s in range(1, v.n_stages):
        rb = ob.stage(codes, s)
        r32 = o32.stage(codes, s)
        got = v.stage(codes, s, rb.size + 8)
        assert got.shape == rb.shape
        assert snr_db(rb, got) >= 30.0, (s, snr_db(rb, got))
        assert snr_db(r32, got) >= snr_db(r32, rb) - 3.0, (s, snr_db(r32, got), snr_db(r32, rb))


def test_bf16_full_utterance_within_stated_tolerance(pkg, synth, vo):
    """configs[1]'s utterance (125 frames, full model) in bf16 mode against the fp32 oracle."""
    cfg = synth.FULL
    v = voc_bf16(pkg, synth, "FULL", 64)
    o32 = oracle_for(vo, synth, "FULL", 0)
    codes = synth.make_codes(125, 1000, cfg)
    got = v.synthesize(codes)
    ref = o32.synthesize(codes, 64)
    assert got.size == ref.size
    err = np.abs(got - ref).max()
    snr = snr_db(ref, got)
    assert err <= WAV_MAX_ABS, err
    assert snr >= WAV_SNR_DB, snr


def test_bf16_chunk_loop_ragged_and_batch_bitwise(pkg, synth, vo):
    """Chunk loop / padding / truncation are mode-independent host logic; a batch equals separate calls bit for bit."""
    cfg = synth.SMALL
    v = voc_bf16(pkg, synth, "SMALL", 4)
    o32 = oracle_for(vo, synth, "SMALL", 0)
    seqs = [synth.make_codes(n, 50 + n, cfg) for n in (1, 4, 5, 10, 3)]
    outs = v.synthesize_batch(seqs)
    for c, got in zip(seqs, outs):
        one = v.synthesize(c)
        assert np.array_equal(one.view(np.uint32), got.view(np.uint32))
        ref = o32.synthesize(c, 4)
        assert got.size == ref.size
        assert snr_db(ref, got) >= WAV_SNR_DB
        assert np.abs(got - ref).max() <= WAV_MAX_ABS
    assert v.synthesize(np.zeros((0, 16), dtype=np.int64)).size == 0


def test_bf16_deterministic(pkg, synth):
    cfg = synth.FULL
    v = voc_bf16(pkg, synth, "FULL", 64)
    codes = synth.make_codes(64, 5, cfg)
    a = v.run(codes)
    b = v.run(codes)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


@pytest.mark.parametrize("cfg_name,unit,frames", [("SMALL", 4, 10), ("FULL", 8, 8), ("PAD8", 3, 7)])
def test_bf16_packed_dataflow_equals_fp32_exchange_bitwise(pkg, synth, cfg_name, unit, frames):
    """Production bf16 plans exchange GEMM-to-GEMM tensors of the decoder stack as packed bf16 with the consumer's
    SnakeBeta applied by the producer (ConvGemmArgs::xb / yb); HIPVOC_PACKED=0 keeps them fp32 and rounds in the
    consumer.  Same fp32 values rounded at the same point: the waveforms must be identical.  (With the residual stream in fp32
    and an fp32 input of the final conv on both sides, HIPVOC_RES_BF16=0 HIPVOC_FIN_BF16=0: the fp32 exchange has neither, see
    test_bf16_residual_stream_in_bf16.)"""
    cfg = getattr(synth, cfg_name)
    path = os.path.join(tempfile.gettempdir(), f"hipvoc_test_{cfg_name}_0.safetensors")
    if not os.path.exists(path):
        synth.save_safetensors(path, weights_for(synth, cfg_name, 0), cfg.metadata())
    codes = synth.make_codes(frames, 11, cfg)
    old = os.environ.get("HIPVOC_PACKED")
    old_rb = os.environ.get("HIPVOC_RES_BF16")
    old_fb = os.environ.get("HIPVOC_FIN_BF16")
    try:
        os.environ["HIPVOC_RES_BF16"] = "0"
        os.environ["HIPVOC_FIN_BF16"] = "0"
        os.environ["HIPVOC_PACKED"] = "0"
        plain = pkg.HipVocoder.load(path, device=0, unit_frames=unit, dtype=BF16)
        os.environ["HIPVOC_PACKED"] = "1"
        packed = pkg.HipVocoder.load(path, device=0, unit_frames=unit, dtype=BF16)
    finally:
        for k, o in (("HIPVOC_PACKED", old), ("HIPVOC_RES_BF16", old_rb), ("HIPVOC_FIN_BF16", old_fb)):
            if o is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = o
    a = plain.synthesize(codes)
    b = packed.synthesize(codes)
    assert a.size == b.size and a.size > 0
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), float(np.abs(a - b).max())


def test_channel_counts_that_pad_the_row_tiles(pkg, synth, vo):
    """PAD8: 320 / 160 / 80 / 40 decoder channels -- multiples of 8 (packed exchange on) whose 32-row tiles are partly
    padding, with fused (80, 40) and un-fused (320, 160) residual units: fp32 within the fp32 gates, bf16 within its
    stated tolerance, stage by stage against the emulation."""
    cfg = synth.PAD8
    W = weights_for(synth, "PAD8", 0)
    path = os.path.join(tempfile.gettempdir(), "hipvoc_test_PAD8_0.safetensors")
    synth.save_safetensors(path, W, cfg.metadata())
    md = {k: float(v) for k, v in cfg.metadata().items()}
    o32, ob = vo.Oracle(W, md), vo.Oracle(W, {**md, "bf16_gemm": 1.0})
    codes = synth.make_codes(7, 3, cfg)
    v32 = pkg.HipVocoder.load(path, device=0, unit_frames=3)
    got, ref = v32.synthesize(codes), o32.synthesize(codes, 3)
    assert got.size == ref.size and np.abs(got - ref).max() <= 1e-3 and snr_db(ref, got) >= 60.0
    vb = pkg.HipVocoder.load(path, device=0, unit_frames=3, dtype=BF16)
    c3 = codes[:3]
    for s in range(1, vb.n_stages):
        rb = ob.stage(c3, s)
        g = vb.stage(c3, s, rb.size + 8)
        assert g.shape == rb.shape and snr_db(rb, g) >= 30.0, (s, snr_db(rb, g))
    gb = vb.synthesize(codes)
    assert snr_db(ref, gb) >= WAV_SNR_DB and np.abs(gb - ref).max() <= WAV_MAX_ABS


@pytest.mark.parametrize("cfg_name,unit,frames", [("FULL", 64, 125), ("ODD", 9, 20), ("SMALL", 8, 13)])
def test_bf16_gemm1_bitwise(pkg, synth, cfg_name, unit, frames, monkeypatch):
    """bf16 mode: the small-launch 1-tap GEMM kernel (gemm1_bf16_kernel, 128 channels per stage) against
    conv_gemm_bf16_kernel on the same inputs -- the same MFMA chains, so the same bits."""
    cfg = getattr(synth, cfg_name)
    path = os.path.join(tempfile.gettempdir(), f"hipvoc_test_{cfg_name}_0.safetensors")
    if not os.path.exists(path):
        synth.save_safetensors(path, synth.make_weights(cfg, 0), cfg.metadata())
    codes = synth.make_codes(frames, 31, cfg)
    outs = []
    for wgs in ("0", "1000000"):
        monkeypatch.setenv("HIPVOC_GEMM1_WGS", wgs)
        monkeypatch.setenv("HIPVOC_CONVS_WGS", wgs)             # and conv7 (convs_bf16_kernel, packed and fp32 input)
        v = pkg.HipVocoder.load(path, device=0, unit_frames=unit, dtype=1)
        v.set_profile(True)
        outs.append(v.synthesize(codes))
        # (only the full-size model has packed conv7 inputs and an un-fused dec0 conv7 wide enough for both forms)
        for name in ("gemm1_bf16_kernel", "convs_bf16_kernel<true>", "convs_bf16_kernel<false>")[: 3 if cfg_name == "FULL" else 1]:
            used = any(name in k for k in v.last_profile())
            assert used == (wgs != "0"), (wgs, name, sorted(v.last_profile()))
        del v
    assert outs[0].size and np.array_equal(outs[0].view(np.uint32), outs[1].view(np.uint32))


@pytest.mark.parametrize("unit,frames,n_seq", [(64, 125, 1), (7, 33, 3), (64, 64, 9)])
def test_bf16_block3_walk_equals_three_unit_launches_bitwise(pkg, synth, unit, frames, n_seq, monkeypatch):
    """bf16 mode, block 3 (96 channels): resblock3_bf16_kernel runs the block's three residual units in ONE launch that
    walks along time (kernels_blk3.hip) -- per output element the same MFMA chains, SnakeBeta evaluations, roundings and
    residual adds as three launches of the fused per-unit kernel (HIPVOC_BLK3=0), so the waveforms must be identical bit
    for bit: whatever the number of spans a unit is cut into (a span that does not start at position 0 rebuilds its
    history from the tile to its left), for a lone utterance, short units and a pass of several sequences."""
    cfg = synth.FULL
    path = os.path.join(tempfile.gettempdir(), "hipvoc_test_FULL_0.safetensors")
    if not os.path.exists(path):
        synth.save_safetensors(path, weights_for(synth, "FULL", 0), cfg.metadata())
    seqs = [synth.make_codes(frames, 77 + i, cfg) for i in range(n_seq)]
    monkeypatch.setenv("HIPVOC_BLK3", "0")
    ref_v = pkg.HipVocoder.load(path, device=0, unit_frames=unit, dtype=BF16)
    ref_v.set_profile(True)
    ref = ref_v.synthesize_batch(seqs)
    assert not any("resblock3" in k for k in ref_v.last_profile())
    del ref_v
    monkeypatch.setenv("HIPVOC_BLK3", "1")
    v = pkg.HipVocoder.load(path, device=0, unit_frames=unit, dtype=BF16)
    for spans in (None, "1", "2", "7", "100000"):
        if spans is None:
            monkeypatch.delenv("HIPVOC_BLK3_SPANS", raising=False)
        else:
            monkeypatch.setenv("HIPVOC_BLK3_SPANS", spans)       # read at every launch
        v.set_profile(spans is None)
        got = v.synthesize_batch(seqs)
        if spans is None:
            assert any("resblock3_bf16_kernel" in k for k in v.last_profile()), sorted(v.last_profile())
            assert not any("b3_resunit" in k for k in v.last_profile())
        for a, b in zip(ref, got):
            assert a.size == b.size and a.size > 0
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), (spans, float(np.abs(a - b).max()))
    del v


@pytest.mark.parametrize("cfg_name,unit,frames", [("FULL", 64, 125), ("FULL", 5, 13), ("SMALL", 8, 13)])
def test_bf16_fused_units_in_4_wave_workgroups_bitwise(pkg, synth, cfg_name, unit, frames, monkeypatch):
    """bf16 mode: the packed fused residual units (blocks 2 / 3) in 4-wave workgroups of 128 positions
    (conv_gemm_bf16_w4_kernel) against the 8-wave form (HIPVOC_BF16_W4=0): the same MFMA chain per output element, so the
    same bits.  (HIPVOC_BLK3=0 keeps block 3 on the per-unit kernel so that both tile heights are covered.)"""
    cfg = getattr(synth, cfg_name)
    path = os.path.join(tempfile.gettempdir(), f"hipvoc_test_{cfg_name}_0.safetensors")
    if not os.path.exists(path):
        synth.save_safetensors(path, weights_for(synth, cfg_name, 0), cfg.metadata())
    codes = synth.make_codes(frames, 41, cfg)
    monkeypatch.setenv("HIPVOC_BLK3", "0")
    monkeypatch.setenv("HIPVOC_BF16_RING_UNIT", "0")              # (block 2's units otherwise run on resunit_bf16_ring_kernel)
    outs = []
    for w4 in ("0", "1"):
        monkeypatch.setenv("HIPVOC_BF16_W4", w4)                  # read at every launch
        v = pkg.HipVocoder.load(path, device=0, unit_frames=unit, dtype=BF16)
        v.set_profile(True)
        outs.append(v.synthesize(codes))
        if cfg_name == "FULL":
            used = sorted(k for k in v.last_profile() if "w4_kernel" in k)
            assert (len(used) == 2) == (w4 == "1"), (w4, sorted(v.last_profile()))
        del v
    assert outs[0].size and np.array_equal(outs[0].view(np.uint32), outs[1].view(np.uint32))


@pytest.mark.parametrize("unit,frames,n_seq", [(64, 125, 1), (64, 64, 9), (7, 33, 40), (3, 11, 70)])
def test_bf16_conv7_ring_kernel_bitwise(pkg, synth, unit, frames, n_seq, monkeypatch):
    """bf16 mode, blocks 0 / 1: the conv7 kernels that stage both operands by LDS-DMA through a ring (kernels_bf16_ring.hip) --
    conv7_bf16_ring_kernel (HIPVOC_BF16_RING=1, the default: 4-wave workgroups of 256 positions, four weight slots) and
    conv7_bf16_ring8_kernel (=2: one 8-wave workgroup of 512 positions per CU, eight slots, operand reads across the stage barrier) --
    against conv_gemm_bf16_kernel (=0): the same MFMA chain per output element and the same epilogue arithmetic (the ring kernel's lean
    packed-only epilogue and, HIPVOC_BF16_RING_EPI=0, the generic one), so the same bits: a lone
    utterance (first tiles with causal zero padding, ragged last tiles), a full pass, and passes of many short units (several
    units per launch, tiles shorter than a workgroup).  HIPVOC_CONVS_WGS=0 keeps the small-launch conv7 kernel out of the way
    so that every size runs the kernels under test."""
    cfg = synth.FULL
    path = os.path.join(tempfile.gettempdir(), "hipvoc_test_FULL_0.safetensors")
    if not os.path.exists(path):
        synth.save_safetensors(path, weights_for(synth, "FULL", 0), cfg.metadata())
    seqs = [synth.make_codes(frames, 177 + i, cfg) for i in range(n_seq)]
    monkeypatch.setenv("HIPVOC_CONVS_WGS", "0")
    outs = []
    for ring, name in (("0", None), ("1", "conv7_bf16_ring_kernel"), ("1g", "conv7_bf16_ring_kernel"), ("2", "conv7_bf16_ring8_kernel<4,8,2>")):
        monkeypatch.setenv("HIPVOC_BF16_RING", ring[0])           # read at every launch
        monkeypatch.setenv("HIPVOC_BF16_RING_EPI", "0" if ring == "1g" else "1")      # the generic epilogue instead of the lean packed-only one
        v = pkg.HipVocoder.load(path, device=0, unit_frames=unit, dtype=BF16)
        v.set_profile(True)
        outs.append(v.synthesize_batch(seqs))
        used = sorted(k for k in v.last_profile() if "conv7_bf16_ring" in k)
        assert len(used) == (2 if name else 0) and all(name in k for k in used), (ring, sorted(v.last_profile()))      # b0_conv7 and b1_conv7
        del v
    for other in outs[1:]:
        for a, b in zip(outs[0], other):
            assert a.size == b.size and a.size > 0
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), float(np.abs(a - b).max())


@pytest.mark.parametrize("trim_right,unit,frames,n_seq", [(False, 64, 125, 1), (False, 64, 64, 9), (False, 5, 13, 50), (True, 64, 125, 2), (True, 3, 11, 70)])
def test_bf16_ring_kernel_general_form_bitwise(pkg, synth, trim_right, unit, frames, n_seq, monkeypatch):
    """bf16 mode: conv_bf16_ring_kernel<TN, NTAPS, EPI, RW> (kernels_bf16_ring.hip: the 4-wave ring kernel with counted waits, any
    tile height) on the transposed convs of blocks 0 - 2 -- octet-interleaved rows (strides 8 and 4) and phase-tile rows
    (stride 5: tile height 5, three weight slots) -- and on the conv7 layers (HIPVOC_BF16_RING=3),
    against conv_gemm_bf16_oi / _pt_kernel and conv_gemm_bf16_kernel: the same MFMA chains and epilogues, so the same bits.
    Both trim policies: the right-only one reads one position LEFT of each output (causal zero padding in the first tile), the
    default one position to the right.  HIPVOC_CONVS_WGS=0 keeps the small-launch conv7 kernel out of the way."""
    import dataclasses
    cfg = dataclasses.replace(synth.FULL, transconv_trim_right_only=True) if trim_right else synth.FULL
    path = os.path.join(tempfile.gettempdir(), f"hipvoc_test_FULL_0{'_trimr' if trim_right else ''}.safetensors")
    if not os.path.exists(path):
        synth.save_safetensors(path, weights_for(synth, "FULL", 0), cfg.metadata())
    seqs = [synth.make_codes(frames, 277 + i, cfg) for i in range(n_seq)]
    monkeypatch.setenv("HIPVOC_CONVS_WGS", "0")
    outs = []
    for ring, tc in (("0", "0"), ("3", "1")):
        monkeypatch.setenv("HIPVOC_BF16_RING", ring)              # read at every launch
        monkeypatch.setenv("HIPVOC_BF16_RING_TCONV", tc)
        v = pkg.HipVocoder.load(path, device=0, unit_frames=unit, dtype=BF16)
        assert v.unit_samples == (unit * 1920 if trim_right else unit * 1920 - 555)
        v.set_profile(True)
        outs.append(v.synthesize_batch(seqs))
        used = sorted(k.split("|")[1] for k in v.last_profile() if "conv_bf16_ring_kernel" in k)
        want = ["conv_bf16_ring_kernel<3,2,3,4>", "conv_bf16_ring_kernel<4,2,2,4>", "conv_bf16_ring_kernel<4,2,2,4>", "conv_bf16_ring_kernel<4,7,0,4>",
                "conv_bf16_ring_kernel<4,7,0,4>", "conv_bf16_ring_kernel<5,2,3,3>"] if ring != "0" else []
        if (unit < 64 or len(seqs) * ((frames + unit - 1) // unit) < 9) and want:      # (small launches: the first blocks' trans-convs run at a lowered tile height, on the old kernels)
            assert set(used) >= {"conv_bf16_ring_kernel<4,7,0,4>"} and set(used) <= set(want), (ring, sorted(v.last_profile()))
        else:
            assert used == want, (ring, sorted(v.last_profile()))
        del v
    for a, b in zip(*outs):
        assert a.size == b.size and a.size > 0
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), float(np.abs(a - b).max())


@pytest.mark.parametrize("unit,frames,n_seq", [(64, 125, 1), (64, 64, 9), (7, 33, 40), (2, 7, 64)])
def test_bf16_resunit_ring_kernel_bitwise(pkg, synth, unit, frames, n_seq, monkeypatch):
    """bf16 mode, block 2 (192 channels): the whole residual unit on the ring (kernels_bf16_ring.hip: 4-wave workgroups, two per CU,
    conv7 and 1x1-conv weights as one stream of LDS-DMA slots: resunit_bf16_ring_kernel<6>, 128 positions per workgroup) against
    conv_gemm_bf16_kernel<6,true,true> (HIPVOC_BF16_RING_UNIT=0): the same MFMA chains, SnakeBeta evaluations, roundings and
    residual add per output element, so the same bits -- a lone utterance, a full pass, passes of many short units."""
    cfg = synth.FULL
    path = os.path.join(tempfile.gettempdir(), "hipvoc_test_FULL_0.safetensors")
    if not os.path.exists(path):
        synth.save_safetensors(path, weights_for(synth, "FULL", 0), cfg.metadata())
    seqs = [synth.make_codes(frames, 377 + i, cfg) for i in range(n_seq)]
    outs = []
    for mode, name in (("0", None), ("1", "resunit_bf16_ring_kernel<6,true>"), ("1g", "resunit_bf16_ring_kernel<6,true>")):
        monkeypatch.setenv("HIPVOC_BF16_RING_UNIT", mode[0])      # read at every launch
        monkeypatch.setenv("HIPVOC_BF16_RING_EPI", "0" if mode == "1g" else "1")      # the generic fused epilogue instead of the lean one
        v = pkg.HipVocoder.load(path, device=0, unit_frames=unit, dtype=BF16)
        v.set_profile(True)
        outs.append(v.synthesize_batch(seqs))
        used = [k for k in v.last_profile() if "resunit_bf16_ring" in k]
        assert used == ([f"b2_resunit|{name}"] if name else []), (mode, sorted(v.last_profile()))
        del v
    for other in outs[1:]:
        for a, b in zip(outs[0], other):
            assert a.size == b.size and a.size > 0
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), float(np.abs(a - b).max())


@pytest.mark.parametrize("unit,frames,n_seq,stream", [(64, 125, 1, None), (64, 64, 9, None), (64, 64, 9, "15"), (7, 33, 40, "15"), (2, 7, 64, None)])
def test_bf16_conv1_ring_kernel_bitwise(pkg, synth, unit, frames, n_seq, stream, monkeypatch):
    """bf16 mode, blocks 0 / 1: the 1x1 conv that closes an un-fused residual unit on conv1_bf16_ring_kernel<4> (activation tiles and
    weight slots of three channel blocks in flight by LDS-DMA, two 4-wave workgroups per CU, the next tile's residual requested before
    the current tile's epilogue, a lean epilogue) against conv_gemm_bf16_kernel<4,false,true> (HIPVOC_BF16_RING_C1=0): the same MFMA
    chain and epilogue arithmetic per output element, so the same bits -- forced onto small launches too (HIPVOC_BF16_RING_C1_WGS=1),
    with the fp32 residual stream of the default and with the packed bf16 one (HIPVOC_RES_BF16=15)."""
    cfg = synth.FULL
    path = os.path.join(tempfile.gettempdir(), "hipvoc_test_FULL_0.safetensors")
    if not os.path.exists(path):
        synth.save_safetensors(path, weights_for(synth, "FULL", 0), cfg.metadata())
    seqs = [synth.make_codes(frames, 577 + i, cfg) for i in range(n_seq)]
    if stream:
        monkeypatch.setenv("HIPVOC_RES_BF16", stream)
    monkeypatch.setenv("HIPVOC_BF16_RING_C1_WGS", "1")
    outs = []
    for mode in ("0", "1"):
        monkeypatch.setenv("HIPVOC_BF16_RING_C1", mode)           # read at every launch
        v = pkg.HipVocoder.load(path, device=0, unit_frames=unit, dtype=BF16)
        v.set_profile(True)
        outs.append(v.synthesize_batch(seqs))
        used = sorted(k for k in v.last_profile() if "conv1_bf16_ring" in k)
        assert used == (["b0_conv1|conv1_bf16_ring_kernel<4>", "b1_conv1|conv1_bf16_ring_kernel<4>"] if mode == "1" else []), (mode, sorted(v.last_profile()))
        v.close()
    for a, b in zip(outs[0], outs[1]):
        assert a.size == b.size and a.size > 0
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), float(np.abs(a - b).max())


@pytest.mark.parametrize("dtype,n_units", [(BF16, 200), (0, 150)])
def test_two_lanes_of_one_handle_bitwise_and_ordered(pkg, synth, dtype, n_units, monkeypatch):
    """hipvoc_decode_units_device with HIPVOC_LANES=2 (the bf16 mode's default) alternates the passes of one call between two streams
    with a workspace each.  Same kernels on the same inputs: the waveforms are those of one lane, bit for bit -- for a call of several
    passes with a ragged last one, for two calls back to back without a sync in between (the second call's lanes start behind the
    first call's), and the handle's own stream ends behind both lanes (hipvoc_sync alone makes every output readable)."""
    import torch
    cfg = synth.FULL
    path = os.path.join(tempfile.gettempdir(), "hipvoc_test_FULL_0.safetensors")
    if not os.path.exists(path):
        synth.save_safetensors(path, weights_for(synth, "FULL", 0), cfg.metadata())
    T = 8
    codes = np.stack([synth.make_codes(T, 5000 + i, cfg) for i in range(n_units)])
    d_codes = torch.from_numpy(codes).cuda()
    seqs = [synth.make_codes(n, 6000 + i, cfg) for i, n in enumerate([3, 21, 8, 1, 40, 17] * 25)]          # 350 units of 8 frames
    outs, host, pcm = [], [], []
    for lanes in ("1", "2"):
        monkeypatch.setenv("HIPVOC_LANES", lanes)                  # read at load
        v = pkg.HipVocoder.load(path, device=0, unit_frames=T, dtype=dtype)
        stride = (v.unit_samples + 63) // 64 * 64
        d_out = torch.zeros((2, n_units, stride), dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        v.decode_units_device(d_codes.data_ptr(), n_units, d_out[0].data_ptr(), stride)
        v.decode_units_device(d_codes.data_ptr(), n_units - 70, d_out[1].data_ptr(), stride)       # no sync in between
        v.sync()
        outs.append(d_out.cpu().numpy()[:, :, : v.unit_samples].copy())
        # the host entry points take the same lanes (pass k in slot k & 1 on lane k & 1): ragged sequences over several passes
        host.append(np.concatenate(v.synthesize_batch(seqs)))
        pcm.append(b"".join(b for b, _ in v.handle_vocode_batch([q.astype("<i8").tobytes() for q in seqs], [len(q) for q in seqs])))
        v.close()
    assert np.abs(outs[0][0]).max() > 0
    assert np.array_equal(outs[0].view(np.uint32), outs[1].view(np.uint32))
    assert host[0].size > 1800 * sum(len(q) for q in seqs) and np.array_equal(host[0].view(np.uint32), host[1].view(np.uint32))
    assert len(pcm[0]) == 2 * host[0].size and pcm[0] == pcm[1]
    assert np.array_equal(outs[1][0, : n_units - 70].view(np.uint32), outs[1][1, : n_units - 70].view(np.uint32))


@pytest.mark.parametrize("cfg_name,unit,frames", [("FULL", 8, 8), ("FULL", 64, 61), ("SMALL", 10, 10), ("PAD8", 7, 5)])
def test_bf16_residual_stream_in_bf16(pkg, synth, vo, cfg_name, unit, frames, monkeypatch):
    """bf16 mode, packed exchange: the residual stream of every decoder block but the first (HIPVOC_RES_BF16=15: the first too,
    =0: fp32 everywhere as in round 2) is rounded to bf16 wherever it is written and kept as a packed tensor (ConvGemmArgs::rbf); and the final conv filters the packed bf16
    tensor of its activated input (HIPVOC_FIN_BF16=0: fp32).  (a) The debug plans keep the stream as fp32 tensors that hold the rounded values: their last stage is the production
    waveform bit for bit (both forms of every epilogue round at the same point).  (b) What the rounding costs: the two waveforms agree
    to 35 dB or better (40 dB on the full model), and each is inside the mode's gate against the fp32 oracle."""
    cfg = getattr(synth, cfg_name)
    path = os.path.join(tempfile.gettempdir(), f"hipvoc_test_{cfg_name}_0.safetensors")
    if not os.path.exists(path):
        synth.save_safetensors(path, weights_for(synth, cfg_name, 0), cfg.metadata())
    codes = synth.make_codes(frames, 77, cfg)
    padded = np.zeros((unit, codes.shape[1]), dtype=np.int64)
    padded[:frames] = codes
    ref = oracle_for(vo, synth, cfg_name, 0).synthesize(codes, unit)
    outs = {}
    for rb in ("auto", "15", "0"):                                 # the default (every block but the first), every eligible block, none
        if rb == "auto":
            monkeypatch.delenv("HIPVOC_RES_BF16", raising=False)
        else:
            monkeypatch.setenv("HIPVOC_RES_BF16", rb)              # read at load
        monkeypatch.setenv("HIPVOC_FIN_BF16", "0" if rb == "0" else "1")     # ... and the final conv's input packed bf16 / fp32
        v = pkg.HipVocoder.load(path, device=0, unit_frames=unit, dtype=BF16)
        got = v.synthesize(codes)
        whole = v.run(padded)                                       # the whole padded unit, as the last stage holds it
        last = v.stage(padded, v.n_stages - 1, whole.size + 8)
        assert np.array_equal(last.reshape(-1).view(np.uint32), whole.reshape(-1).view(np.uint32)), (rb, float(np.abs(last.reshape(-1) - whole.reshape(-1)).max()))
        assert got.size == ref.size and snr_db(ref, got) >= WAV_SNR_DB and np.abs(got - ref).max() <= WAV_MAX_ABS, (rb, snr_db(ref, got))
        if cfg_name == "FULL":
            v.set_profile(True)
            v.synthesize(codes)
            names = sorted(v.last_profile())
            assert any("resunit_bf16_ring_kernel<6,true>" in k for k in names) == (rb != "0"), names
            assert any("final_conv_packed_kernel<7>" in k for k in names) == (rb != "0"), names
        outs[rb] = got
        v.close()
    print(f"{cfg_name} T={unit}: bf16 stream (default / all blocks) vs fp32 stream {snr_db(outs['0'], outs['auto']):.1f} / {snr_db(outs['0'], outs['15']):.1f} dB; "
          f"vs the fp32 oracle {snr_db(ref, outs['auto']):.2f} / {snr_db(ref, outs['15']):.2f} / {snr_db(ref, outs['0']):.2f} dB")
    assert snr_db(outs["0"], outs["auto"]) >= 35.0 and snr_db(outs["0"], outs["15"]) >= 35.0


def test_bf16_stream_is_the_fp32_value_rounded_once(pkg, synth, monkeypatch):
    """Where the bf16 residual stream starts (block 1's transposed conv on the full model: everything upstream is the same arithmetic
    with the stream on or off) the stage tap with the stream on is EXACTLY the fp32 tap rounded to bf16 (nearest even): the stream
    is rounded once, where it is written, from the value the fp32-stream path stores.  Downstream the two paths differ (that
    is what the rounding costs, test_bf16_residual_stream_in_bf16); block 0's taps are identical (its stream stays fp32)."""
    cfg = synth.FULL
    path = os.path.join(tempfile.gettempdir(), "hipvoc_test_FULL_0.safetensors")
    if not os.path.exists(path):
        synth.save_safetensors(path, weights_for(synth, "FULL", 0), cfg.metadata())
    T = 8
    codes = synth.make_codes(T, 91, cfg)
    monkeypatch.delenv("HIPVOC_RES_BF16", raising=False)
    monkeypatch.delenv("HIPVOC_FIN_BF16", raising=False)
    on = pkg.HipVocoder.load(path, device=0, unit_frames=T, dtype=BF16)
    monkeypatch.setenv("HIPVOC_RES_BF16", "0")
    monkeypatch.setenv("HIPVOC_FIN_BF16", "0")
    off = pkg.HipVocoder.load(path, device=0, unit_frames=T, dtype=BF16)
    cap = 2 * 1024 * 1024
    C1 = cfg.decoder_dim // 4                                       # channels of block 1 (1536 -> 768 -> 384)
    seen_dec0 = seen_b0 = seen_b1 = False
    for s in range(1, on.n_stages):
        a, b = on.stage(codes, s, cap), off.stage(codes, s, cap)
        assert a.shape == b.shape
        seen_dec0 = seen_dec0 or a.shape[0] == cfg.decoder_dim      # decoder[0]'s output: the blocks follow
        if not seen_dec0:
            continue
        if a.shape[0] == cfg.decoder_dim // 2:                      # block 0: trans-conv and units, fp32 stream either way
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), s
            seen_b0 = True
        if a.shape[0] == C1:                                        # block 1's trans-conv output: the first rounded tensor
            u = b.view(np.uint32).astype(np.uint64)
            r = ((u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000).astype(np.uint32)
            assert np.array_equal(a.view(np.uint32), r), s
            assert not np.array_equal(a.view(np.uint32), b.view(np.uint32))
            seen_b1 = True
            break
    assert seen_b0 and seen_b1
    on.close(); off.close()
