#!/usr/bin/env python3
"""bf16x3 mode against the fp32 mode of the same library and against the fp32 CPU oracle (full model, one 10 s utterance)."""
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
W = synth.make_weights(cfg, 0)
if not os.path.exists(path):
    synth.save_safetensors(path, W, cfg.metadata())
codes = synth.make_codes(125, 1000, cfg)
def snr(ref, got):
    return 10 * np.log10(np.sum(ref.astype(np.float64) ** 2) / max(np.sum((ref.astype(np.float64) - got) ** 2), 1e-300))
outs = {}
for name, dt in (("f32", 0), ("bf16", 1), ("bf16x3", 2)):
    v = pkg.HipVocoder.load(path, unit_frames=64, dtype=dt)
    outs[name] = v.synthesize(codes)
    del v
for name in ("bf16", "bf16x3"):
    print(f"{name} vs fp32 HIP: max-abs {np.abs(outs[name] - outs['f32']).max():.3e}  SNR {snr(outs['f32'], outs[name]):.1f} dB")
if len(sys.argv) > 1 and sys.argv[1] == "oracle":
    vo = _pkgload.load_oracle()
    o = vo.Oracle(W, {k: float(v) for k, v in cfg.metadata().items()})
    vo.set_threads(0)
    ref = o.synthesize(codes[:64], 64)
    for name in ("f32", "bf16x3"):
        got = outs[name][: ref.size]
        print(f"{name} vs CPU oracle (unit 0): max-abs {np.abs(got - ref).max():.3e}  SNR {snr(ref, got):.1f} dB")
