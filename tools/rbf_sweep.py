#!/usr/bin/env python3
"""HIPVOC_RES_BF16 per block: what each block's bf16 residual stream costs (dB against the fp32 oracle, three seeds' worth of codes on
seed-0 weights) and saves (ms per 64-unit pass).  `rbf_sweep.py [masks...]`"""
import os, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import _pkgload
pkg = _pkgload.load_pkg(); synth = pkg.synth
import torch
torch.cuda.init()
cfg = synth.FULL
path = os.path.join(tempfile.gettempdir(), "hipvoc_prof_full.safetensors")
W = synth.make_weights(cfg, 0)
if not os.path.exists(path):
    synth.save_safetensors(path, W, cfg.metadata())
vo = _pkgload.load_oracle()                 # (a checker tool, as tools/margin_stages.py: the oracle is the reference, never the product)
o32 = vo.Oracle(W, {k: float(v) for k, v in cfg.metadata().items()})
seqs = [synth.make_codes(64, 9000 + i, cfg) for i in range(3)]
refs = [o32.synthesize(c, 64) for c in seqs]
def snr(ref, got):
    return 10 * np.log10(np.sum(ref.astype(np.float64) ** 2) / np.sum((got.astype(np.float64) - ref) ** 2))
N = 256
codes = np.stack([synth.make_codes(64, 1000 + i, cfg) for i in range(N)])
d_codes = torch.from_numpy(codes).cuda()
os.environ["HIPVOC_LANES"] = "1"
for mask in (sys.argv[1:] or ["0", "1", "2", "4", "3", "7"]):
    os.environ["HIPVOC_RES_BF16"] = mask
    v = pkg.HipVocoder.load(path, unit_frames=64, dtype=1)
    stride = (v.unit_samples + 63) // 64 * 64
    d_out = torch.empty((N, stride), dtype=torch.float32, device="cuda")
    db = [snr(r, v.synthesize(c)) for c, r in zip(seqs, refs)]
    for _ in range(2):
        v.decode_units_device(d_codes.data_ptr(), N, d_out.data_ptr(), stride); v.sync()
    t = time.perf_counter()
    for _ in range(3):
        v.decode_units_device(d_codes.data_ptr(), N, d_out.data_ptr(), stride); v.sync()
    ms = (time.perf_counter() - t) / 3 / (N / 64) * 1e3
    print(f"HIPVOC_RES_BF16={mask}: {ms:6.2f} ms per pass (one lane)   " + " / ".join(f"{x:.2f}" for x in db) + " dB", flush=True)
    v.close()
