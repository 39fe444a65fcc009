#!/usr/bin/env python3
"""Single-utterance latency (host codes in, host waveform out) against the sum of its kernels' device time."""
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
for dtype in (0, 1, 2):          # fp32, bf16, bf16x3
    v = pkg.HipVocoder.load(path, unit_frames=64, dtype=dtype)
    codes = synth.make_codes(125, 1000, cfg)
    for _ in range(3):
        v.synthesize(codes)
    ts = []
    for _ in range(20):
        t = time.perf_counter(); v.synthesize(codes); ts.append(time.perf_counter() - t)
    v.set_profile(True)
    v.synthesize(codes)
    prof = v.last_profile()
    v.set_profile(False)
    dev = sum(x[0] for x in prof.values()); n = sum(x[2] for x in prof.values())
    for k, (ms, fl, ln) in sorted(prof.items(), key=lambda kv: -kv[1][0]):
        print(f"    {k:58s} {ms:7.3f} ms {100*ms/dev:5.1f}%  {fl/ms/1e9 if ms > 0 else 0:7.1f} TFLOP/s  x{ln}")
    print(f"dtype={dtype}: latency median {1e3*np.median(ts):.2f} ms, min {1e3*min(ts):.2f}; event-to-event device time {dev:.2f} ms over {n} launches")
