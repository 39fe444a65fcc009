#!/usr/bin/env python3
"""Per-step device time of one encoder pass (full model, 64 clips of 10 s)."""
import os, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import _pkgload
pkg = _pkgload.load_pkg(); synth = pkg.synth
import torch
torch.cuda.init()
cfg = synth.ENC_FULL
path = os.path.join(tempfile.gettempdir(), "hipenc_prof_full.safetensors")
if not os.path.exists(path):
    synth.save_safetensors(path, synth.make_encoder_weights(cfg, 0), cfg.metadata())
e = pkg.HipSpeechTokenizer.load(path)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
n = 240000
audio = np.stack([synth.make_audio(n, 2000 + i % 4) for i in range(B)])
d_audio = torch.from_numpy(audio).cuda()
d_codes = torch.empty((B, e.n_frames(n), 16), dtype=torch.int64, device="cuda")
for _ in range(2):
    e.encode_device(d_audio.data_ptr(), B, n, n, d_codes.data_ptr()); e.sync()
t = time.perf_counter()
for _ in range(3):
    e.encode_device(d_audio.data_ptr(), B, n, n, d_codes.data_ptr())
e.sync()
dt = (time.perf_counter() - t) / 3
fl = 2 * e.macs_per_sample * n * B
print(f"encoder B={B}: {dt*1e3:.2f} ms/pass  {10*B/dt:.0f} xRT  {fl/dt/1e12:.1f} TFLOP/s")
e.set_profile(True)
e.encode_device(d_audio.data_ptr(), B, n, n, d_codes.data_ptr()); e.sync()
prof = e.last_profile(); tot = sum(x[0] for x in prof.values())
for k, (ms, fl, nl) in sorted(prof.items(), key=lambda kv: -kv[1][0]):
    print(f"  {k:52s} {ms:8.3f} ms {100*ms/tot:5.1f}%  {fl/max(ms,1e-9)/1e9:7.1f} TFLOP/s  x{nl}")
