#!/usr/bin/env python3
"""Where does a packed-plan stage differ from the fp32-exchange plan?  usage: packed_stage.py CFG T STAGE"""
import os, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import _pkgload
pkg = _pkgload.load_pkg(); synth = pkg.synth
import torch
torch.cuda.init()
name, T, st = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
cfg = getattr(synth, name)
path = os.path.join(tempfile.gettempdir(), f"hipvoc_pk_{name}.safetensors")
synth.save_safetensors(path, synth.make_weights(cfg, 0), cfg.metadata())
codes = synth.make_codes(T, 21 + T, cfg)
out = []
for pk in (0, 1):
    os.environ["HIPVOC_PACKED"] = str(pk)
    v = pkg.HipVocoder.load(path, unit_frames=T, dtype=1)
    out.append(v.stage(codes, st, 1 << 26))
a, b = out
print("shape", a.shape)
d = np.abs(a - b)
print("max", d.max(), "frac differing", (d > 0).mean())
pc = (d > 1e-6).mean(axis=1); pp = (d > 1e-6).mean(axis=0)
np.set_printoptions(linewidth=200, precision=2, suppress=True)
print("per channel (first 64):", pc[:64])
print("per position:", pp)
