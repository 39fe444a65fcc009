#!/usr/bin/env python3
"""bf16 conv-GEMM mode vs the oracle (fp32 and bf16-emulating), per stage; then a timed full-size pass."""
import os, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import _pkgload
pkg = _pkgload.load_pkg(); synth = pkg.synth; vo = _pkgload.load_oracle()
import torch
torch.cuda.init()

def snr(ref, x):
    ref = ref.astype(np.float64); x = x.astype(np.float64)
    return 10 * np.log10((ref ** 2).sum() / max(((x - ref) ** 2).sum(), 1e-300))

for cfg_name, T in (("SMALL", 8), ("ODD", 9), ("FULL", 3)):
    cfg = getattr(synth, cfg_name)
    W = synth.make_weights(cfg, 0)
    path = os.path.join(tempfile.gettempdir(), f"hipvoc_bf16chk_{cfg_name}.safetensors")
    synth.save_safetensors(path, W, cfg.metadata())
    md = {k: float(v) for k, v in cfg.metadata().items()}
    o32 = vo.Oracle(W, md)
    ob = vo.Oracle(W, dict(md, bf16_gemm=1))
    v = pkg.HipVocoder.load(path, unit_frames=T, dtype=1)
    codes = synth.make_codes(T, 21 + T, cfg)
    print(f"== {cfg_name} T={T}  stages={v.n_stages}", flush=True)
    for s in range(v.n_stages):
        rb = ob.stage(codes, s); r32 = o32.stage(codes, s)
        got = v.stage(codes, s, rb.size + 8)
        sc = np.abs(r32).max()
        print(f"  stage {s:2d} shape {got.shape}  |ref|max {sc:9.3e}  vs bf16-oracle: maxabs {np.abs(got-rb).max():9.3e} snr {snr(rb,got):6.1f} dB"
              f"   vs fp32-oracle: maxabs {np.abs(got-r32).max():9.3e} snr {snr(r32,got):6.1f} dB   (oracle bf16 vs fp32 snr {snr(r32,rb):6.1f})", flush=True)
    v.close()

cfg = synth.FULL
path = os.path.join(tempfile.gettempdir(), "hipvoc_bf16chk_FULL.safetensors")
v = pkg.HipVocoder.load(path, unit_frames=64, dtype=1)
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
print(f"bf16 G={G}: {dt*1e3:.2f} ms/pass  {5.12*G/dt:.0f} xRT  {2*v.unit_macs*G/dt/1e12:.1f} TFLOP/s")
v.set_profile(True)
v.decode_units_device(d_codes.data_ptr(), G, d_out.data_ptr(), stride); v.sync()
prof = v.last_profile(); tot = sum(x[0] for x in prof.values())
for k, (ms, fl, n) in sorted(prof.items(), key=lambda kv: -kv[1][0])[:24]:
    print(f"  {k:46s} {ms:8.3f} ms {100*ms/tot:5.1f}%  {fl/max(ms,1e-9)/1e9:7.1f} TFLOP/s  x{n}")
