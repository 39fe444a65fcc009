#!/usr/bin/env python3
"""bench_encoder.py -- throughput of the speech-tokenizer ENCODER path on MI355X (the headline
vocoder line is bench.py's).  Same contract and JSON shape as bench.py:

    python bench_encoder.py --gpus N --steps K --warmup W

A "step" is one pass over the batch: per GPU, 64 synthetic 10 s clips (24 kHz mono f32) -> [125, 16] codes each,
audio and codes resident in HBM when the timed region starts.  Weak scaling, no data-path collective (clips are
independent).  roofline: the dominant kernel by device time (hipEvents on the library's stream), algorithmic FLOPs
over its duration against the fp32 MFMA peak.  cpu_baseline: the CPU oracle (oracle/enc_oracle.c, kind "port") on one
3 s clip on this box's host cores, whose codes are also compared with the GPU's.
"""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import _pkgload  # noqa: E402

CLIPS_PER_GPU = 64
CLIP_SECONDS = 10
PEAK_F32_MFMA_TFLOPS = 157.3    # MI355X_MICROARCH.md, "Peak FP32 (matrix)"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--clips", type=int, default=CLIPS_PER_GPU)
    ap.add_argument("--no-cpu", action="store_true")
    args = ap.parse_args()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    dist = None
    torch.cuda.set_device(local_rank)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    dev = torch.device("cuda", local_rank)

    pkg = _pkgload.load_pkg()
    synth = pkg.synth
    cfg = synth.ENC_FULL
    W = synth.make_encoder_weights(cfg, 0)
    wpath = os.path.join(tempfile.gettempdir(), f"hipenc_bench_full_r{rank}.safetensors")
    synth.save_safetensors(wpath, W, cfg.metadata())
    enc = pkg.HipSpeechTokenizer.load(wpath, device=local_rank)
    os.remove(wpath)

    B, n = args.clips, 24000 * CLIP_SECONDS
    T = enc.n_frames(n)
    # a handful of distinct clips tiled over the batch (generation is host-side numpy; content does not change the work)
    base = [synth.make_audio(n, 2000 + rank * 8 + i) for i in range(8)]
    audio = np.stack([base[i % 8] for i in range(B)])
    d_audio = torch.from_numpy(audio).to(dev)
    d_codes = torch.empty((B, T, enc.n_codebooks), dtype=torch.int64, device=dev)
    torch.cuda.synchronize()

    def step():
        enc.encode_device(d_audio.data_ptr(), B, n, n, d_codes.data_ptr())

    def fence():
        enc.sync()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    value = world * B * CLIP_SECONDS * args.steps / dt

    roofline, kernels, lat_ms, cpu, parity = None, {}, None, None, None
    if rank == 0:
        clip3 = base[0][: 24000 * 3]
        enc.encode(clip3)
        ts = []
        for _ in range(5):
            t = time.perf_counter()
            got3 = enc.encode(clip3)
            ts.append(time.perf_counter() - t)
        lat_ms = 1e3 * float(np.median(ts))
        enc.set_profile(True)
        step()
        enc.sync()
        prof = enc.last_profile()
        enc.set_profile(False)
        for key, (ms, flops, launches) in prof.items():
            k = kernels.setdefault(key.split("|")[1], {"ms": 0.0, "flops": 0.0, "launches": 0})
            k["ms"] += ms; k["flops"] += flops; k["launches"] += launches
        dom = max(kernels, key=lambda nme: kernels[nme]["ms"])
        kd = kernels[dom]
        ach = kd["flops"] / (kd["ms"] * 1e-3) / 1e12
        # HBM bytes per launch from the committed PMC passes of the same pass size (see bench.py: pmc_traffic)
        traffic, traffic_src = None, None
        try:
            pj = json.load(open(os.path.join(ROOT, "profiles", "r01_encoder_pmc.json")))
            kk = pj.get("kernels", {}).get(dom.replace(" ", ""))
            if kk and pj.get("units_per_pass") == B:
                traffic, traffic_src = kk["fetch_bytes_per_launch"] + kk["write_bytes_per_launch"], "profiles/r01_encoder_pmc.json"
        except (OSError, ValueError):
            pass
        roofline = {"bound": "mfma", "kernel": dom, "achieved": round(ach, 3), "peak": PEAK_F32_MFMA_TFLOPS,
                    "unit": "TFLOP/s", "frac": round(ach / PEAK_F32_MFMA_TFLOPS, 4), "traffic": traffic,
                    "traffic_source": traffic_src,
                    "avg_launch_ms": round(kd["ms"] / kd["launches"], 4), "launches_per_pass": kd["launches"],
                    "clips_per_pass": B, "share_of_pass_time": round(kd["ms"] / sum(k["ms"] for k in kernels.values()), 3)}
        if world == 1 and not args.no_cpu:
            eo = _pkgload.load_enc_oracle()
            oracle = eo.EncOracle(W, {k: float(v) for k, v in cfg.metadata().items()})
            cores = eo.set_threads(0)
            t = time.perf_counter()
            ref3 = oracle.encode(clip3)
            cdt = time.perf_counter() - t
            cpu = {"value": round(3.0 / cdt, 3), "unit": "audio-sec/sec", "cores": cores, "kind": "port",
                   "sample": f"one 3 s clip of the same batch, oracle/enc_oracle.c with {cores} OpenMP threads, {cdt:.1f} s"}
            nb = len(cfg.ratios)
            lat = enc.stage(clip3, 2 + 2 * nb + cfg.num_layers)
            ref_lat = oracle.stage(clip3, 2 + 2 * nb + cfg.num_layers)
            num = float(np.sum(ref_lat.astype(np.float64) ** 2)); den = float(np.sum((lat.astype(np.float64) - ref_lat) ** 2))
            parity = {"frames_identical": float((got3 == ref3).all(axis=1).mean()), "codes_identical": float((got3 == ref3).mean()),
                      "quantiser_bit_exact_on_gpu_latent": bool(np.array_equal(oracle.quantize(lat), got3)),
                      "latent_max_abs_err": float(np.abs(lat - ref_lat).max()),
                      "latent_snr_db": round(10 * np.log10(num / max(den, 1e-300)), 1), "frames_checked": int(ref3.shape[0])}
        flops_per_s_audio = 2.0 * enc.macs_per_sample * 24000
        line = {
            "metric": "encoder audio-sec/sec (xRT)", "value": round(value, 1), "unit": "audio-sec/sec", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"speech-tokenizer encoder (SURVEY 8f-4): {B} synthetic {CLIP_SECONDS} s clips per GPU "
                                   f"(24 kHz mono f32 -> [{T}, {enc.n_codebooks}] codes each) in one pass, fp32, seeded "
                                   f"random-init weights of the Mimi-style 12.5 Hz encoder (56 M params)",
                       "clips_per_gpu": B, "clip_seconds": CLIP_SECONDS},
            "gflop_per_audio_second": round(flops_per_s_audio / 1e9, 3),
            "achieved_tflops_whole_path": round(value * flops_per_s_audio / 1e12, 2),
            "single_clip_3s_latency_ms": round(lat_ms, 2),
            "roofline": roofline, "cpu_baseline": cpu, "parity_vs_oracle": parity,
            "kernels": {k: {"ms_per_pass": round(v["ms"], 3), "tflops": round(v["flops"] / max(v["ms"], 1e-9) / 1e9, 2),
                            "launches": v["launches"]} for k, v in sorted(kernels.items(), key=lambda kv: -kv[1]["ms"])},
        }
        print(json.dumps(line), flush=True)
    enc.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
