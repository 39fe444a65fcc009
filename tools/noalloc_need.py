#!/usr/bin/env python3
"""Workspace the allocation-free entry point needs per dtype and unit count (from its own error text / success)."""
import os, sys, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _pkgload
pkg = _pkgload.load_pkg(); synth = pkg.synth
import torch
path = os.path.join(tempfile.gettempdir(), "hipvoc_prof_full.safetensors")
if not os.path.exists(path):
    synth.save_safetensors(path, synth.make_weights(synth.FULL, 0), synth.FULL.metadata())
for dt in (0, 1, 2):
    v = pkg.HipVocoder.load(path, device=0, unit_frames=64, dtype=dt)
    for n in (1, 2, 3, 4):
        codes = np.stack([synth.make_codes(64, 10 + u, synth.FULL) for u in range(n)])
        d = torch.from_numpy(np.ascontiguousarray(codes)).cuda()
        out = torch.zeros((n, v.unit_samples), dtype=torch.float32, device="cuda")
        try:
            v.decode_units_device_noalloc(d.data_ptr(), n, out.data_ptr(), v.unit_samples); v.sync()
            print(f"dtype {dt} n={n}: ok")
        except Exception as e:
            print(f"dtype {dt} n={n}: {e}")
    v.close()
