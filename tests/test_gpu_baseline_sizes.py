"""GPU tests at the sizes BASELINE.json's configs state (configs[2..4] were only exercised by the bench):

  configs[2]  256 x 10 s utterances through hipvoc_decode_batch (512 units = 8 passes of 64): bit-equal to the same
              utterances decoded one at a time; oracle parity on sampled units (fp32 gates 1e-3 / 60 dB).
  configs[3]  one 7 500-frame stream = 118 units (crosses the 64-unit pass boundary; 12-frame tail): decoded whole,
              bit-equal to its 2/4/8-shard decodes stitched, and to the device-resident entry point the bench times.
  configs[4]  one GPU's share of the 4 096-utterance node batch (512 utterances) in the bf16 conv-GEMM mode: batch ==
              one at a time bit for bit; the mode's stated tolerance (30 dB / 3e-2) against the fp32 oracle on a sample.
All through the C ABI (ctypes on libhipvoc.so).
"""
import os
import tempfile

import numpy as np
import pytest

from conftest import oracle_for, snr_db, weights_for

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _torch_hip_first():
    import torch
    if torch.cuda.is_available():
        torch.cuda.init()
    yield


@pytest.fixture(scope="module")
def full_path(synth):
    path = os.path.join(tempfile.gettempdir(), "hipvoc_test_FULL_0.safetensors")
    if not os.path.exists(path):
        synth.save_safetensors(path, weights_for(synth, "FULL", 0), synth.FULL.metadata())
    return path


def test_config2_batch_of_256_utterances(pkg, synth, vo, full_path):
    cfg = synth.FULL
    v = pkg.HipVocoder.load(full_path, device=0, unit_frames=64)
    utts = [synth.make_codes(125, 1000 + u, cfg) for u in range(256)]
    batch = v.synthesize_batch(utts)
    assert len(batch) == 256 and all(b.size == 122325 + 61 * 1920 for b in batch)
    # one at a time: a different launch size and pass composition for every unit, same bits
    for u in list(range(0, 256, 17)) + [63, 64, 255]:
        assert np.array_equal(batch[u].view(np.uint32), v.synthesize(utts[u]).view(np.uint32)), u
    again = v.synthesize_batch(utts[128:])                     # the same units in other pass slots
    assert all(np.array_equal(a, b) for a, b in zip(again, batch[128:]))
    assert max(float(np.abs(b).max()) for b in batch) <= 1.0
    # GPU-side i16 of the same batch (the VocodeBatch request) == the reference's conversion of the f32 samples
    pcm = v.handle_vocode_batch([u.astype("<i8").tobytes() for u in utts[:70]], [125] * 70)
    for (pb, n), b in zip(pcm, batch[:70]):
        assert n == b.size and np.array_equal(np.frombuffer(pb, dtype="<i2"), vo.pcm16(b))
    o = oracle_for(vo, synth, "FULL", 0)
    for u, k in ((0, 0), (97, 1), (255, 0)):                   # full units and a padded 61-frame tail
        seg = utts[u][k * 64:(k + 1) * 64]
        ref = o.synthesize(seg, 64)
        got = batch[u][k * 122325: k * 122325 + ref.size]
        assert float(np.abs(got - ref).max()) <= 1e-3 and snr_db(ref, got) >= 60.0, (u, k)
    v.close()


def test_config3_ten_minute_stream_sharded_and_stitched(pkg, synth, vo, full_path):
    import torch
    cfg = synth.FULL
    sh = pkg.sharding
    v = pkg.HipVocoder.load(full_path, device=0, unit_frames=64)
    stream = synth.make_codes(7500, 4242, cfg)
    spans = sh.unit_spans(7500, 64)
    assert len(spans) == 118 and spans[-1] == (7488, 7500)
    whole = v.synthesize(stream)
    assert whole.size == 117 * 122325 + 12 * 1920               # 600 s minus the 555-sample trim of each full unit
    for world in (2, 4, 8):
        parts = [sh.decode_stream_shard(v.synthesize, stream, 64, r, world) for r in range(world)]
        assert sum(hi - lo for lo, hi in (sh.shard_range(118, r, world) for r in range(world))) == 118
        assert np.array_equal(np.concatenate(parts).view(np.uint32), whole.view(np.uint32)), f"stitching differs at world={world}"
    # the device-resident entry point (what bench.py --workload longform times), 118 units in two passes
    codes = np.zeros((118, 64, 16), dtype=np.int64)
    for i, (a, b) in enumerate(spans):
        codes[i, : b - a] = stream[a:b]
    stride = (v.unit_samples + 63) // 64 * 64
    d_codes = torch.from_numpy(codes).cuda()
    d_out = torch.empty((118, stride), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    v.decode_units_device(d_codes.data_ptr(), 118, d_out.data_ptr(), stride)
    v.sync()
    g = d_out.cpu().numpy()
    dev = np.concatenate([g[i, : min((b - a) * 1920, v.unit_samples)] for i, (a, b) in enumerate(spans)])
    assert np.array_equal(dev.view(np.uint32), whole.view(np.uint32))
    # oracle parity where the pass boundary and the tail are: units 63, 64 and the 12-frame unit 117
    o = oracle_for(vo, synth, "FULL", 0)
    for i in (63, 64, 117):
        a, b = spans[i]
        ref = o.synthesize(stream[a:b], 64)
        got = whole[i * 122325: i * 122325 + ref.size]
        assert float(np.abs(got - ref).max()) <= 1e-3 and snr_db(ref, got) >= 60.0, i
    v.close()


def test_config4_one_gpus_share_of_the_4096_batch_bf16(pkg, synth, vo, full_path):
    cfg = synth.FULL
    v = pkg.HipVocoder.load(full_path, device=0, unit_frames=64, dtype=1)
    utts = [synth.make_codes(125, 1000 + 512 * 3 + u, cfg) for u in range(512)]      # rank 3's utterances in bench.py's numbering
    batch = v.synthesize_batch(utts)
    assert len(batch) == 512 and all(b.size == 122325 + 61 * 1920 for b in batch)
    for u in (0, 31, 32, 300, 511):
        assert np.array_equal(batch[u].view(np.uint32), v.synthesize(utts[u]).view(np.uint32)), u
    o = oracle_for(vo, synth, "FULL", 0)
    ref = o.synthesize(utts[300][:64], 64)
    got = batch[300][: ref.size]
    assert snr_db(ref, got) >= 30.0 and float(np.abs(got - ref).max()) <= 3e-2
    v.close()
