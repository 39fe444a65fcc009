#!/usr/bin/env python3
"""Per-stage GPU-vs-oracle comparison (debug aid; prints max-abs error of every stage)."""
import os, sys, time, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import _pkgload
pkg = _pkgload.load_pkg(); vo = _pkgload.load_oracle(); synth = pkg.synth

def main():
    cases = sys.argv[1:] or ["SMALL:5:8", "ODD:9:9", "SMALL:4:4", "FULL:3:3"]
    for case in cases:
        name, T, unit = case.split(":"); T = int(T); unit = int(unit)
        cfg = getattr(synth, name)
        W = synth.make_weights(cfg, 0)
        path = os.path.join(tempfile.gettempdir(), f"hipvoc_{name}_0.safetensors")
        synth.save_safetensors(path, W, cfg.metadata())
        t = time.time(); v = pkg.HipVocoder.load(path, unit_frames=unit); tl = time.time() - t
        o = vo.Oracle(W, {k: float(x) for k, x in cfg.metadata().items()})
        codes = synth.make_codes(T, 1000, cfg)
        print(f"== {case}: load {tl:.2f}s stages {v.n_stages} unit_macs {v.unit_macs/1e9:.3f} G", flush=True)
        worst = 0.0
        for s in [1000] + list(range(v.n_stages)):
            ref = o.stage(codes, s)
            got = v.stage(codes, s, ref.size + 16)
            if got.shape != ref.shape:
                print(f" stage {s}: SHAPE {got.shape} vs {ref.shape}"); continue
            err = float(np.abs(got - ref).max()); worst = max(worst, err)
            print(f" stage {s:4d} {str(ref.shape):>16} maxabs {err:.3e} refstd {ref.std():.3f}" + ("  <<<<" if not err < 1e-3 else ""), flush=True)
        print(f"   worst {worst:.3e}")
        v.close()

if __name__ == "__main__":
    main()
