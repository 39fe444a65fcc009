#!/usr/bin/env python3
"""PROJECTION of the multi-GPU curves from ONE GPU (no multi-GPU node was available to measure on): device-resident
passes of the per-rank shares of BASELINE configs[3] (one 7 500-frame stream = 118 units: 118 / 59 / 30 / 15 units per rank
at 1 / 2 / 4 / 8 GPUs) in fp32, and configs[4]'s per-GPU share (512 utterances, bf16; weak scaling: the share does not
change with N).  Units are independent and no rank waits for another before the final gather, so the strong-scaling time
at N GPUs is the time of the LARGEST share plus the gather; the gather (<= 15 units x 0.49 MB per rank over xGMI) is not
measurable here and is left out -- the table says so.

    python3 tools/scaling_projection.py > profiles/r03_scaling_projection.txt      (on the GPU box)
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def bench(*args):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu", "--no-host", "--no-latency", *args],
                       capture_output=True, text=True, timeout=600)
    if r.returncode:
        sys.stderr.write(r.stderr[-2000:])
        raise SystemExit(r.returncode)
    return json.loads(r.stdout.strip().splitlines()[-1])


def main():
    print("PROJECTION, NOT A MEASUREMENT: one MI355X timing the per-rank shares; the inter-GPU gather is not included.")
    print()
    print("configs[3] long-form stream, 7 500 frames = 118 reference units, fp32, strong scaling")
    print(f"{'GPUs':>4s} {'largest share (units)':>22s} {'ms per step':>12s} {'ms per unit':>12s} {'audio-s/s (whole job)':>22s} {'efficiency vs 1 GPU':>20s}")
    base = None
    rows = []
    for n, units in ((1, 118), (2, 59), (4, 30), (8, 15)):
        frames = 7500 if units == 118 else units * 64
        d = bench("--workload", "longform", "--frames", str(frames), "--steps", "10", "--warmup", "2")
        ms = d["ms_per_step"]
        whole = 600.0 / (ms * 1e-3)
        if base is None:
            base = whole
        rows.append((n, units, ms))
        print(f"{n:4d} {units:22d} {ms:12.2f} {ms / units:12.3f} {whole:22.0f} {whole / (n * base):20.3f}")
    print()
    print("configs[4] batch of 4 096 utterances = 512 per GPU, bf16, weak scaling: every rank runs the same share;")
    d = bench("--workload", "batch4096", "--steps", "5", "--warmup", "1")
    print(f"one GPU's share: {d['ms_per_step']:.1f} ms per step, {d['value']:.0f} audio-s/s; N GPUs (no data-path collective, "
          f"barrier only) project to N x that: 8 GPUs {8 * d['value']:.0f} audio-s/s = {8 * d['value']:.0f}x real time")
    print()
    print("configs[2] batch of 256 utterances per GPU, fp32, weak scaling (the headline workload):")
    d = bench("--steps", "10", "--warmup", "2")
    print(f"one GPU: {d['ms_per_step']:.1f} ms per step, {d['value']:.0f} audio-s/s; 8 GPUs project to {8 * d['value']:.0f} audio-s/s")


if __name__ == "__main__":
    main()
