"""hipvoc_decode_units_device_noalloc (include/hipvoc.h): the entry point that allocates nothing -- its workspace is the
library's file-scope `static __device__` scratch array -- against hipvoc_decode_units_device on the same device-resident
codes, through the C ABI.  Integer bar of the build contract: the two entry points run the same kernels and launch plan,
so the waveforms must be equal BIT FOR BIT.  Two calls in a row, the second with a smaller input (the array keeps the
first call's contents); then the refusal of a pass that does not fit the array.
"""
import os
import tempfile

import numpy as np
import pytest

from conftest import weights_for

pytestmark = pytest.mark.gpu


def _load(pkg, synth, cfg_name, unit, dtype=0):
    cfg = getattr(synth, cfg_name)
    path = os.path.join(tempfile.gettempdir(), f"hipvoc_test_{cfg_name}_0.safetensors")
    if not os.path.exists(path):
        synth.save_safetensors(path, weights_for(synth, cfg_name, 0), cfg.metadata())
    return pkg.HipVocoder.load(path, device=0, unit_frames=unit, dtype=dtype)


def _both(v, codes_units):
    """codes_units [n, T, 16] -> (reference entry point's output, no-alloc entry point's output), host arrays."""
    import torch
    n = codes_units.shape[0]
    stride = (v.unit_samples + 63) // 64 * 64
    d_codes = torch.from_numpy(np.ascontiguousarray(codes_units)).cuda()
    d_ref = torch.full((n, stride), 7.0, dtype=torch.float32, device="cuda")
    d_got = torch.full((n, stride), -7.0, dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    v.decode_units_device(d_codes.data_ptr(), n, d_ref.data_ptr(), stride)
    v.sync()
    v.decode_units_device_noalloc(d_codes.data_ptr(), n, d_got.data_ptr(), stride)
    v.sync()
    L = v.unit_samples
    return d_ref.cpu().numpy()[:, :L], d_got.cpu().numpy()[:, :L]


def _units(synth, cfg, n, T, seed, tail=None):
    c = np.zeros((n, T, 16), dtype=np.int64)
    for u in range(n):
        c[u] = synth.make_codes(T, seed + u, cfg)
    if tail is not None:
        c[-1, tail:] = 0                     # a ragged last unit, padded with code 0 as vocoder.rs:62,70 does
    return c


@pytest.mark.parametrize("dtype", [0, 1, 2])
def test_noalloc_equals_allocating_entry_point_small(pkg, synth, dtype):
    import torch
    torch.cuda.init()
    cfg = synth.SMALL
    v = _load(pkg, synth, "SMALL", 8, dtype)
    for n, seed, tail in ((11, 500, 5), (3, 700, None), (1, 900, 2)):      # second and third calls are smaller
        ref, got = _both(v, _units(synth, cfg, n, 8, seed, tail))
        assert np.isfinite(ref).all() and ref.std() > 0
        assert np.array_equal(ref.view(np.uint32), got.view(np.uint32)), (dtype, n)
    v.close()


def test_noalloc_full_size_two_calls_second_smaller(pkg, synth):
    """Full-size decoder, fp32: one 10 s utterance (2 units of 64 frames, 64 + 61 padded), then a single unit."""
    import torch
    torch.cuda.init()
    cfg = synth.FULL
    v = _load(pkg, synth, "FULL", 64)
    ref2, got2 = _both(v, _units(synth, cfg, 2, 64, 1000, tail=61))
    assert np.array_equal(ref2.view(np.uint32), got2.view(np.uint32))
    ref1, got1 = _both(v, _units(synth, cfg, 1, 64, 2000))
    assert np.array_equal(ref1.view(np.uint32), got1.view(np.uint32))
    assert np.abs(ref1).max() <= 1.0 and ref1.std() > 0.02
    # a pass that does not fit the 384 MiB array is refused before anything is launched, and the handle keeps serving
    import ctypes as C
    stride = (v.unit_samples + 63) // 64 * 64
    d_codes = torch.zeros((3, 64, 16), dtype=torch.int64, device="cuda")
    d_out = torch.full((3, stride), 3.0, dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    rc = v._L.hipvoc_decode_units_device_noalloc(v._h, d_codes.data_ptr(), 3, d_out.data_ptr(), stride)
    assert rc == -7, rc
    assert b"MiB" in v._L.hipvoc_last_error(v._h)
    v.sync()
    assert float(d_out.min()) == 3.0 and float(d_out.max()) == 3.0
    rc = v._L.hipvoc_decode_units_device_noalloc(v._h, d_codes.data_ptr(), 200, d_out.data_ptr(), stride)
    assert rc == -7
    ref1b, got1b = _both(v, _units(synth, cfg, 1, 64, 2000))
    assert np.array_equal(got1b.view(np.uint32), got1.view(np.uint32))
    v.close()


def test_noalloc_two_handles_on_one_gpu_back_to_back(pkg, synth):
    """Two handles on device 0 (the worker's default: --contexts 2) share the device's scratch array and enqueue on their
    own non-blocking streams.  Called one after the other with NO sync in between, the library has to order the two passes
    on the device; each result must still equal the allocating entry point's, bit for bit."""
    import torch
    torch.cuda.init()
    cfg = synth.SMALL
    va = _load(pkg, synth, "SMALL", 8)
    vb = _load(pkg, synth, "SMALL", 8)
    stride = (va.unit_samples + 63) // 64 * 64
    L = va.unit_samples
    for rep in range(4):
        ca = _units(synth, cfg, 9, 8, 100 + 10 * rep)
        cb = _units(synth, cfg, 7, 8, 300 + 10 * rep, tail=3)
        d_ca = torch.from_numpy(ca).cuda()
        d_cb = torch.from_numpy(cb).cuda()
        ref_a = torch.zeros((9, stride), dtype=torch.float32, device="cuda")
        ref_b = torch.zeros((7, stride), dtype=torch.float32, device="cuda")
        got_a = torch.full((9, stride), -7.0, dtype=torch.float32, device="cuda")
        got_b = torch.full((7, stride), -7.0, dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        va.decode_units_device(d_ca.data_ptr(), 9, ref_a.data_ptr(), stride)
        vb.decode_units_device(d_cb.data_ptr(), 7, ref_b.data_ptr(), stride)
        va.sync()
        vb.sync()
        # back to back, alternating handles, nothing waits on the host in between
        va.decode_units_device_noalloc(d_ca.data_ptr(), 9, got_a.data_ptr(), stride)
        vb.decode_units_device_noalloc(d_cb.data_ptr(), 7, got_b.data_ptr(), stride)
        va.decode_units_device_noalloc(d_ca.data_ptr(), 9, got_a.data_ptr(), stride)
        vb.decode_units_device_noalloc(d_cb.data_ptr(), 7, got_b.data_ptr(), stride)
        va.sync()
        vb.sync()
        for ref, got in ((ref_a, got_a), (ref_b, got_b)):
            r, g = ref.cpu().numpy()[:, :L], got.cpu().numpy()[:, :L]
            assert np.isfinite(r).all() and r.std() > 0
            assert np.array_equal(r.view(np.uint32), g.view(np.uint32)), rep
    va.close()
    vb.close()
