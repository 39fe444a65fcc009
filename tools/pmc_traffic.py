#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into HBM-side bytes per launch, per kernel symbol AND per step
of the launch plan, beside each step's algorithmic bytes.

    tools/pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> [out.json units_per_pass "about" [steps.txt]]

steps.txt: the stderr of the profiled program run with HIPVOC_TRACE_STEPS=1 (lines "HIPVOC_STEP <i> <step> <kernel> <macs>
<bytes>", one per launch, in launch order).  rocprofv3's counter rows are in dispatch order, so the k-th library kernel of
the run is the k-th traced launch: that attributes every dispatch to its step (b1_conv7, b3_resunit ...), which the kernel
symbol alone cannot (conv_gemm_kernel<4,...> runs 21 different layers per pass).

Units and corrections per MI355X_MICROARCH.md (HBM section): both counters are in KiB; on gfx950 FETCH_SIZE reports half
of the bytes of wide coalesced reads, so it is doubled; WRITE_SIZE is exact.
"""
import collections
import csv
import json
import sys


def sym(name):
    return name.split("(")[0].replace("void ", "").replace("hipvoc::", "")


def rows(path, counter):
    """[(dispatch_id, kernel symbol, grid, value)] in dispatch order."""
    out = []
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        out.append((int(r["Dispatch_Id"]), sym(r["Kernel_Name"]), r.get("Grid_Size", ""), float(r["Counter_Value"])))
    # one row per dispatch (the counter may be reported per XCC / dimension: sum them)
    agg = collections.OrderedDict()
    for d, k, g, v in sorted(out):
        if d in agg:
            agg[d][2] += v
        else:
            agg[d] = [k, g, v]
    return [(d, k, g, v) for d, (k, g, v) in agg.items()]


def main():
    fr = rows(sys.argv[1], "FETCH_SIZE")
    wr = rows(sys.argv[2], "WRITE_SIZE")
    f, nf, w, nw = collections.defaultdict(float), collections.Counter(), collections.defaultdict(float), collections.Counter()
    for _, k, _, v in fr:
        f[k] += v
        nf[k] += 1
    for _, k, _, v in wr:
        w[k] += v
        nw[k] += 1
    print("kernel, launches, fetch_MB_per_launch(x2 corrected), write_MB_per_launch, total_MB_per_launch")
    for k in sorted(f, key=lambda k: -(2 * f[k] + w.get(k, 0))):
        fl = 2 * f[k] * 1024 / nf[k] / 1e6
        wl = w.get(k, 0) * 1024 / max(nw.get(k, 0), 1) / 1e6
        print(f"{k}, {nf[k]}, {fl:.2f}, {wl:.2f}, {fl + wl:.2f}")
    out = None
    if len(sys.argv) > 3:
        out = {"_about": sys.argv[5] if len(sys.argv) > 5 else "", "units_per_pass": int(sys.argv[4]), "kernels": {}}
        for k in f:
            out["kernels"][k.replace(" ", "")] = {
                "launches": nf[k], "fetch_bytes_per_launch": int(2 * f[k] * 1024 / nf[k]),
                "write_bytes_per_launch": int(w.get(k, 0) * 1024 / max(nw.get(k, 0), 1))}
    if len(sys.argv) > 6:
        steps = []
        for ln in open(sys.argv[6], errors="ignore"):
            if ln.startswith("HIPVOC_STEP "):
                _, i, name, kern, macs, byt = ln.split()
                steps.append((int(i), name, kern, float(macs), float(byt)))
        lib = lambda rs: [r for r in rs if not r[1].startswith("__amd_rocclr")]
        per = collections.OrderedDict()
        for which, rs in (("fetch", lib(fr)), ("write", lib(wr))):
            if len(rs) != len(steps):
                print(f"# {which}: {len(rs)} library dispatches but {len(steps)} traced launches: per-step attribution skipped")
                per = None
                break
            for (d, k, g, v), (i, name, kern, macs, byt) in zip(rs, steps):
                if k.replace(" ", "") != kern.replace(" ", ""):
                    print(f"# {which}: dispatch {d} is {k}, trace says {kern}: per-step attribution skipped")
                    per = None
                    break
                e = per.setdefault((name, kern), {"n": 0, "fetch": 0.0, "write": 0.0, "alg": 0.0})
                if which == "fetch":
                    e["n"] += 1
                    e["alg"] += byt
                e[which] += v * 1024 * (2 if which == "fetch" else 1)
            if per is None:
                break
        if per:
            print("\nstep | kernel, launches, fetch_MB, write_MB, measured_MB, algorithmic_MB (activations in + out + residual), measured/algorithmic   [per launch]")
            if out is not None:
                out["steps"] = {}
            for (name, kern), e in sorted(per.items(), key=lambda kv: -(kv[1]["fetch"] + kv[1]["write"])):
                n = e["n"]
                fm, wm, am = e["fetch"] / n / 1e6, e["write"] / n / 1e6, e["alg"] / n / 1e6
                ratio = (fm + wm) / am if am else float("nan")
                print(f"{name}|{kern}, {n}, {fm:.1f}, {wm:.1f}, {fm + wm:.1f}, {am:.1f}, {ratio:.2f}")
                if out is not None:
                    out["steps"][f"{name}|{kern}".replace(" ", "")] = {
                        "launches": n, "fetch_bytes_per_launch": int(e["fetch"] / n), "write_bytes_per_launch": int(e["write"] / n),
                        "algorithmic_bytes_per_launch": int(e["alg"] / n)}
    if out is not None:
        json.dump(out, open(sys.argv[3], "w"), indent=1)


if __name__ == "__main__":
    main()
