#!/usr/bin/env python3
"""bf16 mode: packed activation exchange (HIPVOC_PACKED=1) against the fp32 exchange (=0), fused and unfused plans."""
import os, sys, tempfile
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
T = int(sys.argv[1]) if len(sys.argv) > 1 else 8
codes = synth.make_codes(T, 11, cfg)
def run(packed, fuse):
    os.environ["HIPVOC_PACKED"] = str(packed); os.environ["HIPVOC_FUSE"] = str(fuse)
    v = pkg.HipVocoder.load(path, unit_frames=T, dtype=1)
    return v.synthesize(codes)
for fuse in (0, 1):
    a = run(0, fuse); b = run(1, fuse)
    d = np.abs(a - b)
    bad = np.nonzero(d > 0)[0]
    print(f"fuse={fuse}: n={a.size} max diff {d.max():.3e} differing {bad.size} first {bad[:5]} last {bad[-5:] if bad.size else []} nan={np.isnan(b).sum()}")
