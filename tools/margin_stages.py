#!/usr/bin/env python3
"""Per-stage max-abs error of the GPU path against the fp32 CPU oracle at FULL size, for one weight family and seed
(tests/test_gpu_margin.py calls it with HIPVOC_WINO=1 and =0 when a family loses margin, to name the layer).

    python3 tools/margin_stages.py <gauss|hostile> <seed> [dtype 0|1|2] [frames]
"""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import _pkgload  # noqa: E402


def main():
    fam = sys.argv[1] if len(sys.argv) > 1 else "hostile"
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    dtype = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    T = int(sys.argv[4]) if len(sys.argv) > 4 else 16
    import torch
    torch.cuda.init()
    pkg = _pkgload.load_pkg()
    vo = _pkgload.load_oracle()
    synth = pkg.synth
    cfg = synth.FULL
    W = synth.make_weights(cfg, seed, fam)
    path = os.path.join(tempfile.gettempdir(), f"hipvoc_margin_{fam}_{seed}.safetensors")
    if not os.path.exists(path):
        synth.save_safetensors(path, W, cfg.metadata())
    v = pkg.HipVocoder.load(path, device=0, unit_frames=T, dtype=dtype)
    o = vo.Oracle(W, {k: float(x) for k, x in cfg.metadata().items()})
    codes = synth.make_codes(T, 1000 + seed, cfg)
    print(f"{fam} seed {seed} dtype {dtype} T={T} HIPVOC_WINO={os.environ.get('HIPVOC_WINO', '1')}")
    for s in range(v.n_stages):
        ref = o.stage(codes, s)
        got = v.stage(codes, s, ref.size + 8)
        e = float(np.abs(got - ref).max())
        print(f"stage {s:2d} {str(ref.shape):>16s} ref rms {ref.std():9.3e} max {np.abs(ref).max():9.3e}  err {e:9.3e}  err/rms {e / max(ref.std(), 1e-30):9.3e}")
    v.close()


if __name__ == "__main__":
    main()
