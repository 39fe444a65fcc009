"""GPU parity tests: the HIP path, called through the C ABI (ctypes on libhipvoc.so), against the CPU
oracle on the same seeded inputs, against the committed golden fixtures, and -- at full size --
through size-independent properties (unit independence, sharded stitching, determinism).

Tolerances (BASELINE.json north_star): RVQ dequantisation bit-exact; fp32 waveform max-abs <= 1e-3 and
SNR >= 60 dB versus the fp32 CPU path.
"""
import os
import tempfile

import numpy as np
import pytest

from conftest import load_golden, oracle_for, snr_db, weights_for

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _torch_hip_first():
    """torch ships its own HIP runtime: initialise it BEFORE libhipvoc pulls in the system one, the
    same order bench.py uses (device tensors for the device-resident entry point come from torch)."""
    import torch
    if torch.cuda.is_available():
        torch.cuda.init()
    yield

WAV_MAX_ABS = 1e-3
WAV_SNR_DB = 60.0
STAGE_MAX_ABS = 1e-3

_VOC = {}


def voc_for(pkg, synth, cfg_name, seed, unit):
    key = (cfg_name, int(seed), int(unit))
    if key not in _VOC:
        cfg = getattr(synth, cfg_name)
        path = os.path.join(tempfile.gettempdir(), f"hipvoc_test_{cfg_name}_{seed}.safetensors")
        if not os.path.exists(path):
            synth.save_safetensors(path, weights_for(synth, cfg_name, seed), cfg.metadata())
        _VOC[key] = pkg.HipVocoder.load(path, device=0, unit_frames=unit)
    return _VOC[key]


def pad16(codes, cfg):
    """The wire/ABI layout is [n,16]; configs here all use 16 codebooks."""
    assert cfg.num_quantizers == 16
    return codes


# ------------------------------------------------------------------------------------ dequantisation
@pytest.mark.parametrize("cfg_name", ["SMALL", "ODD", "FULL"])
def test_rvq_dequant_bit_exact(pkg, synth, vo, cfg_name):
    cfg = getattr(synth, cfg_name)
    T = 7
    v = voc_for(pkg, synth, cfg_name, 0, T)
    o = oracle_for(vo, synth, cfg_name, 0)
    codes = synth.make_codes(T, 11, cfg)
    codes[0, :] = 0
    codes[1, :] = cfg.codebook_size - 1
    codes[2, 0] = 3                         # cluster_usage == 0 -> clamp(1e-5) path
    codes[3, :] = 7 if cfg.codebook_size > 7 else 0   # a usage < 1e-5 row (every 7th+ entry)
    sums_ref = o.stage(codes, 1000)
    sums = v.stage(codes, 1000, sums_ref.size + 8)
    assert sums.shape == sums_ref.shape
    assert np.array_equal(sums.view(np.uint32), sums_ref.view(np.uint32)), "dequant sums differ bitwise"
    q_ref = o.stage(codes, 0)
    q = v.stage(codes, 0, q_ref.size + 8)
    assert np.array_equal(q.view(np.uint32), q_ref.view(np.uint32)), "projected dequant output differs bitwise"


# ------------------------------------------------------------------------------------ every stage
@pytest.mark.parametrize("cfg_name,T", [("SMALL", 5), ("ODD", 9), ("SMALL", 8)])
def test_every_stage_matches_oracle(pkg, synth, vo, cfg_name, T):
    cfg = getattr(synth, cfg_name)
    v = voc_for(pkg, synth, cfg_name, 0, T)
    o = oracle_for(vo, synth, cfg_name, 0)
    codes = synth.make_codes(T, 21 + T, cfg)
    assert v.n_stages == o.n_stages
    for s in range(v.n_stages):
        ref = o.stage(codes, s)
        got = v.stage(codes, s, ref.size + 8)
        assert got.shape == ref.shape, s
        assert np.abs(got - ref).max() <= STAGE_MAX_ABS, f"stage {s}"


# ------------------------------------------------------------------------------------ golden fixtures
@pytest.mark.parametrize("name", ["small_T5", "small_T3_edge", "odd_T9", "full_T3"])
def test_unit_matches_golden(pkg, synth, name):
    g = load_golden(name)
    T = g["codes"].shape[0]
    v = voc_for(pkg, synth, str(g["cfg"]), int(g["wseed"]), T)
    wav = v.run(g["codes"])
    assert wav.size == g["wav"].size
    assert np.abs(wav - g["wav"]).max() <= WAV_MAX_ABS
    assert snr_db(g["wav"], wav) >= WAV_SNR_DB
    for s in range(int(g["n_stages"])):
        shp = tuple(int(x) for x in g[f"s{s}_shape"][-2:])
        got = v.stage(g["codes"], s, shp[0] * shp[1] + 8)
        assert got.shape == shp
        assert np.abs(got.reshape(-1)[g[f"s{s}_idx"]] - g[f"s{s}_val"]).max() <= STAGE_MAX_ABS, f"stage {s}"


@pytest.mark.parametrize("name", ["small_chunks_10x4", "small_chunks_8x4"])
def test_chunk_loop_matches_golden(pkg, synth, name):
    g = load_golden(name)
    v = voc_for(pkg, synth, str(g["cfg"]), int(g["wseed"]), int(g["max_tokens"]))
    wav = v.synthesize(g["codes"])
    assert wav.size == g["wav"].size
    assert np.abs(wav - g["wav"]).max() <= WAV_MAX_ABS
    assert snr_db(g["wav"], wav) >= WAV_SNR_DB


# ------------------------------------------------------------------------------------ ragged / edge inputs
def test_ragged_lengths_and_batch_equals_separate_calls(pkg, synth, vo):
    cfg = synth.SMALL
    unit = 4
    v = voc_for(pkg, synth, "SMALL", 0, unit)
    o = oracle_for(vo, synth, "SMALL", 0)
    lens = [0, 1, 3, 4, 5, 8, 9, 37]      # empty, sub-unit, exact unit, unit+1, multiples, > one pass (group=8)
    seqs = [synth.make_codes(n, 300 + n, cfg) for n in lens]
    batch = v.synthesize_batch(seqs)
    for n, c, got in zip(lens, seqs, batch):
        single = v.synthesize(c, n)
        assert np.array_equal(single, got), f"batch != single for n={n}"
        if n == 0:
            assert got.size == 0
            continue
        full, rem = divmod(n, unit)
        assert got.size == full * cfg.out_len(unit) + rem * 1920
        ref = o.synthesize(c, unit)
        assert np.abs(got - ref).max() <= WAV_MAX_ABS
        assert snr_db(ref, got) >= WAV_SNR_DB


def test_error_codes_and_recovery(pkg, synth):
    cfg = synth.SMALL
    v = voc_for(pkg, synth, "SMALL", 0, 4)
    good = synth.make_codes(6, 5, cfg)
    bad = good.copy()
    bad[4, 9] = cfg.codebook_size
    with pytest.raises(pkg.HipVocError) as e:
        v.synthesize(bad)
    assert e.value.code == -2
    neg = good.copy()
    neg[0, 0] = -1
    with pytest.raises(pkg.HipVocError) as e:
        v.synthesize(neg)
    assert e.value.code == -2
    # capacity too small
    import ctypes as C
    out = np.empty(100, dtype=np.float32)
    n = C.c_size_t(0)
    rc = v._L.hipvoc_decode(v._h, good.ctypes.data, 6, out.ctypes.data, out.size, C.byref(n))
    assert rc == -3
    # the handle keeps serving (FFI never aborts: mod.rs:246-258 turns errors into error responses)
    a = v.synthesize(good)
    assert a.size == cfg.out_len(4) + 2 * 1920


def test_device_resident_path_flags_bad_codes(pkg, synth):
    import torch
    cfg = synth.SMALL
    v = voc_for(pkg, synth, "SMALL", 0, 4)
    codes = synth.make_codes(4, 5, cfg).reshape(1, 4, 16)
    codes[0, 2, 3] = cfg.codebook_size + 5
    d_codes = torch.from_numpy(codes).cuda()
    stride = (v.unit_samples + 63) // 64 * 64
    d_out = torch.empty((1, stride), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    v.decode_units_device(d_codes.data_ptr(), 1, d_out.data_ptr(), stride)
    with pytest.raises(pkg.HipVocError) as e:
        v.sync()
    assert e.value.code == -2
    v.sync()     # flag is cleared


def test_pcm16_path_bit_exact(pkg, synth, vo):
    """handle_vocode: LE i64 bytes in, LE i16 PCM out; conversion on the GPU must equal mod.rs:521-524."""
    cfg = synth.SMALL
    v = voc_for(pkg, synth, "SMALL", 0, 4)
    codes = synth.make_codes(9, 8, cfg)
    wav = v.synthesize(codes)
    pcm_bytes, n = v.handle_vocode(codes.astype("<i8").tobytes(), 9)
    assert n == wav.size and len(pcm_bytes) == 2 * n
    assert np.array_equal(np.frombuffer(pcm_bytes, dtype="<i2"), vo.pcm16(wav))


def test_pad_frames_are_code_zero(pkg, synth):
    """A 3-frame tail must equal the first 3*1920 samples of the unit padded with code 0 explicitly."""
    cfg = synth.SMALL
    v = voc_for(pkg, synth, "SMALL", 0, 4)
    c = synth.make_codes(3, 9, cfg)
    padded = np.zeros((4, 16), dtype=np.int64)
    padded[:3] = c
    assert np.array_equal(v.synthesize(c), v.run(padded)[: 3 * 1920])


def test_window_binds_when_unit_is_longer_than_window(pkg, synth, vo):
    """ODD has sliding_window 5; a 23-frame unit exercises the kv > q - window mask and 2 query blocks... """
    cfg = synth.ODD
    v = voc_for(pkg, synth, "ODD", 0, 40)
    o = oracle_for(vo, synth, "ODD", 0)
    codes = synth.make_codes(40, 3, cfg)
    ref = o.run_unit(codes)
    got = v.run(codes)
    assert np.abs(got - ref).max() <= WAV_MAX_ABS
    assert snr_db(ref, got) >= WAV_SNR_DB


@pytest.mark.parametrize("store", ["F16", "BF16"])
def test_half_precision_checkpoints_load_like_their_fp32_values(pkg, synth, vo, store, tmp_path):
    """Upstream checkpoints come as F32, F16 or BF16 (csrc/safetensors.cpp converts on load); the decode is then the
    fp32 decode of exactly those rounded values."""
    cfg = synth.SMALL
    W = weights_for(synth, "SMALL", 0)
    if store == "F16":
        Wr = {k: v.astype(np.float16).astype(np.float32) for k, v in W.items()}
    else:
        Wr = {k: synth.from_bf16_bits(synth.to_bf16_bits(v)).reshape(v.shape) for k, v in W.items()}
    path = str(tmp_path / f"small_{store}.safetensors")
    synth.save_safetensors(path, W, cfg.metadata(), dtype=store)
    assert os.path.getsize(path) < 0.6 * sum(v.size * 4 for v in W.values())
    v = pkg.HipVocoder.load(path, device=0, unit_frames=6)
    o = vo.Oracle(Wr, {k: float(x) for k, x in cfg.metadata().items()})
    codes = synth.make_codes(6, 77, cfg)
    sums_ref = o.stage(codes, 1000)
    sums = v.stage(codes, 1000, sums_ref.size + 8)
    assert np.array_equal(sums.view(np.uint32), sums_ref.view(np.uint32))
    ref = o.run_unit(codes)
    got = v.run(codes)
    assert np.abs(got - ref).max() <= WAV_MAX_ABS
    assert snr_db(ref, got) >= WAV_SNR_DB
    v.close()


# ------------------------------------------------------------------------------------ full size
def test_full_single_utterance_matches_oracle(pkg, synth, vo):
    """BASELINE configs[1]: one 10 s utterance (125 frames -> units of 64 + 61), fp32, full model."""
    cfg = synth.FULL
    v = voc_for(pkg, synth, "FULL", 0, 64)
    o = oracle_for(vo, synth, "FULL", 0)
    codes = synth.make_codes(125, 1000, cfg)
    got = v.synthesize(codes)
    ref = o.synthesize(codes, 64)
    assert got.size == ref.size == 122325 + 61 * 1920
    err = np.abs(got - ref).max()
    assert err <= WAV_MAX_ABS, err
    assert snr_db(ref, got) >= WAV_SNR_DB
    assert np.abs(ref).max() < 1.0 and ref.std() > 0.02, "reference must be a non-trivial, unclipped signal"


def test_full_size_properties_batch_and_sharded_stream(pkg, synth):
    """Size-independent properties at BASELINE sizes (no oracle): determinism, unit independence
    (a unit decodes identically in any batch slot / group), and seam-exact stitching of a long
    stream decoded as 1, 2, 4 and 8 contiguous shards (configs[3] semantics on one GPU)."""
    cfg = synth.FULL
    v = voc_for(pkg, synth, "FULL", 0, 64)
    sh = pkg.sharding
    stream = synth.make_codes(64 * 19 + 12, 4242, cfg)          # 20 units: 19 full + a 12-frame tail
    whole = v.synthesize(stream)
    assert whole.size == 19 * 122325 + 12 * 1920
    assert np.array_equal(whole, v.synthesize(stream)), "non-deterministic"
    assert np.all(np.abs(whole) <= 1.0)
    for world in (2, 4, 8):
        parts = [sh.decode_stream_shard(v.synthesize, stream, 64, r, world) for r in range(world)]
        assert np.array_equal(np.concatenate(parts), whole), f"stitching differs at world={world}"
    # unit independence: unit 5 alone == unit 5 inside the stream
    alone = v.synthesize(stream[5 * 64: 6 * 64])
    assert np.array_equal(alone, whole[5 * 122325: 6 * 122325])
    # batch of utterances == separate calls
    utts = [synth.make_codes(125, 1000 + u, cfg) for u in range(5)]
    for got, c in zip(v.synthesize_batch(utts), utts):
        assert np.array_equal(got, v.synthesize(c))


def test_right_only_transconv_trim_policy(pkg, synth, vo):
    """The other trans-conv trim policy (metadata transconv_trim_right_only=1; SURVEY.md 8a: which one the shipped
    vocoder.onnx has is unverifiable offline): every stage and the chunk loop against the oracle, exactly 1920 samples
    per frame, in fp32 and in the bf16 mode (packed exchange)."""
    cfg = synth.SMALL_TRIM_RIGHT
    W = synth.make_weights(cfg, 0)
    path = os.path.join(tempfile.gettempdir(), "hipvoc_test_SMALL_TRIM_RIGHT.safetensors")
    synth.save_safetensors(path, W, cfg.metadata())
    md = {k: float(v) for k, v in cfg.metadata().items()}
    o = vo.Oracle(W, md)
    v = pkg.HipVocoder.load(path, device=0, unit_frames=4)
    assert v.unit_samples == 4 * 1920
    codes = synth.make_codes(4, 77, cfg)
    for s in range(1, v.n_stages):
        ref = o.stage(codes, s)
        got = v.stage(codes, s, ref.size + 8)
        assert got.shape == ref.shape, s
        assert np.abs(got - ref).max() <= 1e-3 * max(1.0, float(np.abs(ref).max())), s
    for n in (1, 4, 7, 10):
        c = synth.make_codes(n, 90 + n, cfg)
        got, ref = v.synthesize(c), o.synthesize(c, 4)
        assert got.size == ref.size == n * 1920
        assert np.abs(got - ref).max() <= 1e-3 and snr_db(ref, got) >= 60.0
    vb = pkg.HipVocoder.load(path, device=0, unit_frames=4, dtype=1)
    ob = vo.Oracle(W, {**md, "bf16_gemm": 1.0})
    c = synth.make_codes(10, 5, cfg)
    gb = vb.synthesize(c)
    assert gb.size == 10 * 1920
    assert snr_db(o.synthesize(c, 4), gb) >= 30.0 and snr_db(ob.synthesize(c, 4), gb) >= 30.0


@pytest.mark.parametrize("cfg_name,unit,frames", [("FULL", 64, 125), ("ODD", 9, 20), ("SMALL", 8, 13)])
def test_gemm1_bitwise(pkg, synth, cfg_name, unit, frames, monkeypatch):
    """The small-launch form of the 1-tap GEMMs (kernels_gemm1.hip: 64 channels per stage, 128 positions per workgroup)
    makes the same MFMA chains as conv_gemm_kernel: with it disabled (HIPVOC_GEMM1_WGS=0) the waveform is the same,
    bit for bit -- which is what lets the choice depend on the launch size."""
    cfg = getattr(synth, cfg_name)
    voc_for(pkg, synth, cfg_name, 0, unit)                      # makes the checkpoint file
    path = os.path.join(tempfile.gettempdir(), f"hipvoc_test_{cfg_name}_0.safetensors")
    codes = synth.make_codes(frames, 31, cfg)
    outs = []
    for wgs in ("0", "1000000"):
        monkeypatch.setenv("HIPVOC_GEMM1_WGS", wgs)
        monkeypatch.setenv("HIPVOC_CONVS_WGS", wgs)             # and of the multi-tap convs (convs_kernel)
        v = pkg.HipVocoder.load(path, device=0, unit_frames=unit)
        v.set_profile(True)
        outs.append(v.synthesize(codes))
        for name in ("gemm1_kernel", "convs_kernel<"):
            used = any(name in k for k in v.last_profile())
            assert used == (wgs != "0"), (wgs, name, sorted(v.last_profile()))
        del v
    assert outs[0].size and np.array_equal(outs[0].view(np.uint32), outs[1].view(np.uint32))


def test_winograd_conv7_against_the_direct_kernels(pkg, synth, vo, monkeypatch):
    """fp32 k = 7 convs of decoder[0] and all four blocks run as Winograd F(2,7) on the matrix cores (kernels_wino.hip): fewer
    multiplications, different rounding.  With it off (HIPVOC_WINO=0) the direct implicit-GEMM kernels run instead.  Both
    must meet the fp32 gates against the oracle on a 10 s utterance (edge tiles: the 61-frame tail, every dilation), the
    two waveforms must agree far inside the gate, and the Winograd kernels must really be the ones that ran."""
    cfg = synth.FULL
    voc_for(pkg, synth, "FULL", 0, 64)                         # makes the checkpoint file
    path = os.path.join(tempfile.gettempdir(), "hipvoc_test_FULL_0.safetensors")
    codes = synth.make_codes(125, 1000, cfg)
    ref = oracle_for(vo, synth, "FULL", 0).synthesize(codes, 64)
    out = {}
    for wino in ("0", "1"):
        monkeypatch.setenv("HIPVOC_WINO", wino)
        v = pkg.HipVocoder.load(path, device=0, unit_frames=64)
        v.set_profile(True)
        out[wino] = v.synthesize(codes)
        used = sorted(k for k in v.last_profile() if "wino_conv7_kernel" in k or "wino_ps_kernel" in k)
        assert bool(used) == (wino == "1"), used
        if wino == "1":
            steps = {k.split("|")[0] for k in used}
            assert steps == {"dec0_conv7", "b0_conv7", "b1_conv7", "b2_conv7", "b3_resunit"}, steps    # b3: whole units (wino_ps_kernel<D,3,true>)
            assert {k.split("<")[1].split(",")[0] for k in used} == {"1", "3", "9"}          # every dilation
        err = float(np.abs(out[wino] - ref).max())
        assert err <= WAV_MAX_ABS and snr_db(ref, out[wino]) >= WAV_SNR_DB, (wino, err, snr_db(ref, out[wino]))
        v.close()
    assert float(np.abs(out["0"] - out["1"]).max()) <= 1e-4
    assert snr_db(out["0"], out["1"]) >= 75.0


@pytest.mark.gpu
def test_winograd_workgroup_shapes_agree(pkg, synth, monkeypatch):
    """The k = 7 convs have five workgroup shapes (HIPVOC_WINO_RT, kernels_wino.hip: 8 waves x 8 points, 4 waves x 2 row tiles,
    4 waves x 32 rows, and the two point-split forms) and block 3's residual units two forms (the fused Winograd unit and, with
    HIPVOC_WINO_UNIT=0, the fused direct kernel).  Same arithmetic per output up to the order of the output transform's sums:
    every shape must give the default's waveform to within fp32 round-off, and must really run the kernel it names."""
    cfg = synth.FULL
    voc_for(pkg, synth, "FULL", 0, 64)                         # makes the checkpoint file
    path = os.path.join(tempfile.gettempdir(), "hipvoc_test_FULL_0.safetensors")
    codes = synth.make_codes(125, 1000, cfg)

    def run(env):
        for k, val in env.items():
            monkeypatch.setenv(k, val)
        v = pkg.HipVocoder.load(path, device=0, unit_frames=64)
        v.set_profile(True)
        out = v.synthesize(codes)
        names = sorted({k.split("|")[1] for k in v.last_profile()})
        v.close()
        for k in env:
            monkeypatch.delenv(k)
        return out, names

    ref, names = run({})
    assert any(n.startswith("wino_ps_kernel<1,3,true") for n in names), names         # block 3: fused Winograd units
    want = {"1": "wino_conv7_kernel<1,false,1,2>", "2": "wino_conv7_kernel<1,false,2,1>", "4": "wino_conv7_kernel<1,false,1,1>",
            "3": "wino_ps_kernel<1,2,false,4,16>", "5": "wino_ps_kernel<1,2,false,2,8>"}
    for mode, kname in want.items():
        out, names = run({"HIPVOC_WINO_RT": mode})
        assert kname in names, (mode, names)
        assert float(np.abs(out - ref).max()) <= 2e-5 and snr_db(ref, out) >= 90.0, (mode, float(np.abs(out - ref).max()), snr_db(ref, out))
    out, names = run({"HIPVOC_WINO_UNIT_WG": "4"})             # the fused unit in 4-wave workgroups (two-phase exchange)
    assert "wino_ps_kernel<1,3,true,2,8>" in names, names
    assert float(np.abs(out - ref).max()) <= 2e-5 and snr_db(ref, out) >= 90.0, (float(np.abs(out - ref).max()), snr_db(ref, out))
    out, names = run({"HIPVOC_WINO_UNIT": "0"})
    assert "conv_gemm_kernel<3,true,4,false>" in names and not any(n.startswith("wino_ps_kernel<1,3,true") for n in names), names
    assert float(np.abs(out - ref).max()) <= 5e-5 and snr_db(ref, out) >= 85.0, (float(np.abs(out - ref).max()), snr_db(ref, out))


@pytest.mark.gpu
def test_gemm_dma_bitwise(pkg, synth, monkeypatch):
    """gemm_dma_kernel (the large 1-tap GEMMs with both operands staged by LDS-DMA) runs the same MFMA chain per output as
    conv_gemm_kernel and gemm1_kernel: with it off (HIPVOC_GEMMD=0) the waveform is the same bit for bit, at a large launch
    (64 units: it is selected for the 1x1 convs) and at a small one (2 units: its small-launch form gemm_dmas_kernel against
    gemm1_kernel)."""
    cfg = synth.FULL
    voc_for(pkg, synth, "FULL", 0, 64)
    path = os.path.join(tempfile.gettempdir(), "hipvoc_test_FULL_0.safetensors")
    codes = [synth.make_codes(125, 3000 + i, cfg) for i in range(6)]
    outs = {}
    for on in ("1", "0"):
        monkeypatch.setenv("HIPVOC_GEMMD", on)
        v = pkg.HipVocoder.load(path, device=0, unit_frames=64)
        v.set_profile(True)
        big = v.synthesize_batch(codes) if hasattr(v, "synthesize_batch") else [v.synthesize(c) for c in codes]
        names = {k.split("|")[1] for k in v.last_profile()}
        small = v.synthesize(codes[0])
        small_names = {k.split("|")[1] for k in v.last_profile()}
        v.close()
        outs[on] = (big, small, names, small_names)
    assert any(n.startswith("gemm_dma_kernel") for n in outs["1"][2]), outs["1"][2]
    assert not any(n.startswith("gemm_dma_kernel") for n in outs["0"][2]), outs["0"][2]
    # the small launch: gemm_dmas_kernel (deep LDS-DMA pipeline) for the transformer's projections, gemm1_kernel with it off
    assert any(n.startswith("gemm_dmas_kernel") for n in outs["1"][3]), outs["1"][3]
    assert not any(n.startswith("gemm_dma") for n in outs["0"][3]) and any(n.startswith("gemm1_kernel") for n in outs["0"][3]), outs["0"][3]
    for x, y in zip(outs["1"][0], outs["0"][0]):
        assert np.array_equal(np.asarray(x).view(np.uint32), np.asarray(y).view(np.uint32))
    assert np.array_equal(outs["1"][1].view(np.uint32), outs["0"][1].view(np.uint32))
    assert np.array_equal(np.asarray(outs["1"][0][0]).view(np.uint32), outs["1"][1].view(np.uint32))      # batch == single


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [0, 1, 2])
def test_layout_and_placement_switches_do_not_change_a_bit(pkg, synth, monkeypatch, dtype):
    """Choices that move work between kernels or permute rows without touching any output element's arithmetic must leave the
    waveform bit-identical, in every dtype mode, for a 10 s utterance (small launches) and a 6-utterance batch:
      HIPVOC_FIN_POST=0          the final SnakeBeta inside final_conv_kernel instead of the last residual unit's epilogue;
      HIPVOC_TCONV_OI=0          bf16 / bf16x3: phase-major rows instead of the octet-interleaved / phase-tile orders (their own
                                 kernel instances -- which must really have run);
      HIPVOC_WINO_TCONV_SMALL=0  fp32: the trans-convs' large-launch workgroup shapes on a small launch too;
      HIPVOC_WINO_TCONV_W4=0     fp32: blocks 1/2's trans-convs in 8-wave workgroups of 16-channel blocks instead of 4-wave / 8-channel ones;
      HIPVOC_WINO_PS_SMALL=0     fp32: two row tiles per wave in the conv7 kernels on small launches too (default there: one);
      HIPVOC_GEMMDS_NW=1, HIPVOC_GEMM1_DMA=1   fp32: one-wave workgroups of gemm_dmas_kernel; gemm1_kernel for the layers the
                                 LDS-DMA kernels otherwise take;
      HIPVOC_GEMMD_EPI=0         fp32: the generic epilogue in gemm_dma_kernel instead of the lean one of the residual units' 1x1 convs."""
    cfg = synth.FULL
    voc_for(pkg, synth, "FULL", 0, 64)                         # makes the checkpoint file
    path = os.path.join(tempfile.gettempdir(), "hipvoc_test_FULL_0.safetensors")
    codes = [synth.make_codes(125, 4000 + i, cfg) for i in range(6)]

    def run(env):
        for k, val in env.items():
            monkeypatch.setenv(k, val)
        v = pkg.HipVocoder.load(path, device=0, unit_frames=64, dtype=dtype)
        v.set_profile(True)
        small = v.synthesize(codes[0])
        names = {k.split("|")[1] for k in v.last_profile()}
        big = v.synthesize_batch(codes) if hasattr(v, "synthesize_batch") else [v.synthesize(c) for c in codes]
        names |= {k.split("|")[1] for k in v.last_profile()}
        v.close()
        for k in env:
            monkeypatch.delenv(k)
        return small, big, names

    ref_small, ref_big, ref_names = run({})
    if dtype == 0:
        assert any(n.startswith("gemm_dmas_kernel<1,2,8>") for n in ref_names), ref_names
        assert any(n.startswith("wino_tconv_kernel<false,1,2,") for n in ref_names), ref_names       # a 64-row small-launch shape
        assert "wino_tconv_kernel<false,2,1,4,8>" in ref_names, ref_names
        assert "wino_ps_kernel<1,1,false,2,8>" in ref_names, ref_names                                # decoder[0] / block 0 on one utterance
    else:
        # (the octet-interleaved / phase-tile row orders: the old kernels' instances or, bf16 mode, the ring kernel's EPI = 2 / 3 ones)
        is_oi = lambda n: "_oi_kernel<" in n or (n.startswith("conv_bf16_ring_kernel<") and n.split(",")[2] == "2")
        is_pt = lambda n: "_pt_kernel<" in n or (n.startswith("conv_bf16_ring_kernel<") and n.split(",")[2] == "3")
        assert any(is_oi(n) for n in ref_names) and any(is_pt(n) for n in ref_names), ref_names
    switches = [{"HIPVOC_FIN_POST": "0"}]
    switches += [{"HIPVOC_GEMMD_EPI": "0"}] if dtype == 0 else []     # fp32: gemm_dma_kernel's generic epilogue instead of the lean residual one
    switches += [{"HIPVOC_WINO_TCONV_SMALL": "0"}, {"HIPVOC_WINO_TCONV_W4": "0"}, {"HIPVOC_WINO_PS_SMALL": "0"}, {"HIPVOC_GEMMDS_NW": "1"}, {"HIPVOC_GEMM1_DMA": "1"}] if dtype == 0 else [{"HIPVOC_TCONV_OI": "0"}]
    for env in switches:
        small, big, names = run(env)
        if "HIPVOC_TCONV_OI" in env:
            assert not any(is_oi(n) or is_pt(n) for n in names), names
        if "HIPVOC_WINO_TCONV_SMALL" in env:
            assert not any(n.startswith("wino_tconv_kernel<false,1,2,") for n in names), names
        if "HIPVOC_WINO_TCONV_W4" in env:
            assert not any(n.endswith(",8>") and n.startswith("wino_tconv_kernel") for n in names) and any(n.startswith("wino_tconv_kernel<false,2,2,4,") for n in names), names
        if "HIPVOC_WINO_PS_SMALL" in env:
            assert not any(n.startswith("wino_ps_kernel<") and ",1,false,2,8>" in n for n in names), names
        if "HIPVOC_GEMMDS_NW" in env:
            assert any(n.startswith("gemm_dmas_kernel<1,1,8>") for n in names), names
        assert np.array_equal(small.view(np.uint32), ref_small.view(np.uint32)), env
        for x, y in zip(big, ref_big):
            assert np.array_equal(np.asarray(x).view(np.uint32), np.asarray(y).view(np.uint32)), env
    assert np.array_equal(np.asarray(ref_big[0]).view(np.uint32), ref_small.view(np.uint32))           # batch == single


# ------------------------------------------------------------------------------------ the static archive on the GPU
def test_static_archive_host_decodes_the_golden_chunks(pkg, synth, tmp_path):
    """libhipvoc.a linked into a plain-C host as build.rs would link it (tests/native/static_link_host.c): one
    hipvoc_decode of the golden 10-frame sequence in 4-frame units.  Every kernel object's fat binary has to have
    registered through the archive for this to produce the golden waveform."""
    import subprocess
    host = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "qwen3-tts_amd", "static_link_host")
    assert os.path.exists(host), "built by `make -C qwen3-tts_amd/csrc`"
    g = load_golden("small_chunks_10x4")
    cfg = getattr(synth, str(g["cfg"]))
    wpath = str(tmp_path / "w.safetensors")
    synth.save_safetensors(wpath, weights_for(synth, str(g["cfg"]), int(g["wseed"])), cfg.metadata())
    cpath, opath = str(tmp_path / "codes.bin"), str(tmp_path / "out.bin")
    g["codes"].astype("<i8").tofile(cpath)
    r = subprocess.run([host, "decode", wpath, cpath, str(g["codes"].shape[0]), str(int(g["max_tokens"])), opath],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    wav = np.fromfile(opath, dtype="<f4")
    assert wav.size == g["wav"].size
    assert np.abs(wav - g["wav"]).max() <= WAV_MAX_ABS
    assert snr_db(g["wav"], wav) >= WAV_SNR_DB
    # and bit-equal to the shared library's result on the same inputs
    v = voc_for(pkg, synth, str(g["cfg"]), int(g["wseed"]), int(g["max_tokens"]))
    assert np.array_equal(v.synthesize(g["codes"]).view(np.uint32), wav.view(np.uint32))
