#!/usr/bin/env python3
"""Do two independent passes overlap usefully on one GPU?  One handle decoding N units (passes back to back on its stream) against
TWO handles (own stream, own workspace) decoding N/2 each at the same time.  `two_lanes.py [f32|bf16] [units] [K]`; with K > 2
also K handles taking the passes in turn.  (The handles are loaded with one lane each: HIPVOC_LANES=1.)"""
import os, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import _pkgload
pkg = _pkgload.load_pkg(); synth = pkg.synth
import torch
torch.cuda.init()
cfg = synth.FULL
path = os.path.join(tempfile.gettempdir(), "hipvoc_prof_full.safetensors")
if not os.path.exists(path):
    synth.save_safetensors(path, synth.make_weights(cfg, 0), cfg.metadata())
dt = {"f32": 0, "bf16": 1, "bf16x3": 2}[sys.argv[1] if len(sys.argv) > 1 else "f32"]
N = int(sys.argv[2]) if len(sys.argv) > 2 else 512
K = int(sys.argv[3]) if len(sys.argv) > 3 else 2
os.environ["HIPVOC_LANES"] = "1"
va = pkg.HipVocoder.load(path, unit_frames=64, dtype=dt)
vb = pkg.HipVocoder.load(path, unit_frames=64, dtype=dt)
codes = np.stack([synth.make_codes(64, 1000 + i, cfg) for i in range(N)])
d_codes = torch.from_numpy(codes).cuda()
stride = (va.unit_samples + 63) // 64 * 64
d_out = torch.empty((N, stride), dtype=torch.float32, device="cuda")
d_ref = torch.empty((N, stride), dtype=torch.float32, device="cuda")
torch.cuda.synchronize()
cb = 64 * cfg.num_quantizers * 8          # bytes of one unit's codes
def one():
    va.decode_units_device(d_codes.data_ptr(), N, d_ref.data_ptr(), stride); va.sync()
def two():
    h = N // 2
    va.decode_units_device(d_codes.data_ptr(), h, d_out.data_ptr(), stride)
    vb.decode_units_device(d_codes.data_ptr() + h * cb, N - h, d_out.data_ptr() + h * stride * 4, stride)
    va.sync(); vb.sync()
def two_interleaved():                      # pass by pass, alternating handles: the queues stay equally deep
    G = va.group_units
    for k, u0 in enumerate(range(0, N, G)):
        v = va if k % 2 == 0 else vb
        v.decode_units_device(d_codes.data_ptr() + u0 * cb, min(G, N - u0), d_out.data_ptr() + u0 * stride * 4, stride)
    va.sync(); vb.sync()
extra = [pkg.HipVocoder.load(path, unit_frames=64, dtype=dt) for _ in range(K - 2)]
def k_interleaved():
    hs = [va, vb] + extra
    G = va.group_units
    for k, u0 in enumerate(range(0, N, G)):
        hs[k % K].decode_units_device(d_codes.data_ptr() + u0 * cb, min(G, N - u0), d_out.data_ptr() + u0 * stride * 4, stride)
    for v in hs: v.sync()
cases = [("one handle", one), ("two handles, halves", two), ("two handles, alternating passes", two_interleaved)]
if K > 2: cases.append((f"{K} handles, alternating passes", k_interleaved))
for name, fn in cases + [("one handle", one)]:
    fn(); fn()
    t = time.perf_counter()
    for _ in range(3): fn()
    dtm = (time.perf_counter() - t) / 3
    same = bool(torch.equal(d_out, d_ref)) if fn is not one else True
    print(f"{name:34s} {dtm*1e3:8.2f} ms per {N} units  {5.12*N/dtm:8.0f} audio-s/s  bit-identical to one handle: {same}")
