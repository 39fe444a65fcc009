#!/usr/bin/env python3
"""Per-step device time of one bf16-mode pass (full model, HIPVOC_GROUP units).  `bf16_steps.py x3`: the bf16x3 mode."""
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
X3 = len(sys.argv) > 1 and sys.argv[1] == "x3"
v = pkg.HipVocoder.load(path, unit_frames=64, dtype=2 if X3 else 1)
G = v.group_units
codes = np.stack([synth.make_codes(64, 1000 + i, cfg) for i in range(G)])
d_codes = torch.from_numpy(codes).cuda()
stride = (v.unit_samples + 63) // 64 * 64
d_out = torch.empty((G, stride), dtype=torch.float32, device="cuda")
torch.cuda.synchronize()
for _ in range(2):
    v.decode_units_device(d_codes.data_ptr(), G, d_out.data_ptr(), stride); v.sync()
t = time.perf_counter()
for _ in range(3):
    v.decode_units_device(d_codes.data_ptr(), G, d_out.data_ptr(), stride)
v.sync()
dt = (time.perf_counter() - t) / 3
print(f"{'bf16x3' if X3 else 'bf16'} G={G}: {dt*1e3:.2f} ms/pass  {5.12*G/dt:.0f} xRT  {2*v.unit_macs*G/dt/1e12:.1f} TFLOP/s")
v.set_profile(True)
v.decode_units_device(d_codes.data_ptr(), G, d_out.data_ptr(), stride); v.sync()
prof = v.last_profile(); tot = sum(x[0] for x in prof.values())
for k, (ms, fl, n) in sorted(prof.items(), key=lambda kv: -kv[1][0]):
    print(f"  {k:46s} {ms:8.3f} ms {100*ms/tot:5.1f}%  {fl/max(ms,1e-9)/1e9:7.1f} TFLOP/s  x{n}")
