#!/usr/bin/env python3
"""bench.py -- vocoder throughput on MI355X (BASELINE.json metric: audio-seconds decoded per second).

    python bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over the batch: per GPU, 256 synthetic 10 s utterances
(125 frames x 16 codebooks each = 2 reference units of 64 frames: 64 + 61 padded with code 0,
src/worker/vocoder.rs:60-85), i.e. BASELINE.json configs[2].  Codes and the output buffer are
resident in HBM when the timed region starts.  Weak scaling: every rank decodes its own 256
utterances; there is no data-path collective (units are independent).

Besides the contract fields the JSON line carries
  roofline     : the dominant kernel (by device time), algorithmic FLOPs / live-measured duration
                 (hipEvents on the library's stream) against the fp32 MFMA peak
  cpu_baseline : the CPU oracle (oracle/voc_oracle.c, kind "port") timed on this box's host cores on
                 a bounded sample of the same workload (rank 0, N=1 only)
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

UTTERANCES_PER_GPU = 256
FRAMES_PER_UTT = 125            # 10.0 s at 12.5 Hz
AUDIO_S_PER_UTT = FRAMES_PER_UTT * 1920 / 24000.0
PEAK_F32_MFMA_TFLOPS = 157.3    # MI355X_MICROARCH.md, "Peak FP32 (matrix)"
PEAK_BF16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md, dense bf16
MODE_TEXT = {
    "f32": "fp32",
    "bf16": "bf16 conv-GEMM mode (fp32 accumulate; parity_vs_oracle is against the fp32 oracle)",
    "bf16x3": "bf16x3 split-operand mode (operands as hi + mid bf16 planes, three bf16 MFMAs per product, fp32 accumulate; "
              "held to the fp32 gates: parity_vs_oracle is against the fp32 oracle)",
}


def pmc_traffic(kernel, units_per_pass):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (profiles/*_pmc_traffic_*.json:
    FETCH_SIZE and WRITE_SIZE collected in separate runs, corrected as MI355X_MICROARCH.md prescribes).  PMC
    counters cannot be read from inside this process, so the figure is the last profiled one, valid only for
    the same kernel symbol and pass size; otherwise null."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic_*.json")), reverse=True):
        try:
            d = json.load(open(path))
        except (OSError, ValueError):
            continue
        k = d.get("kernels", {}).get(kernel.replace(" ", ""))
        if k and d.get("units_per_pass") == units_per_pass:
            return k["fetch_bytes_per_launch"] + k["write_bytes_per_launch"], os.path.relpath(path, ROOT)
    return None, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--utterances", type=int, default=UTTERANCES_PER_GPU)
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--dtype", choices=["f32", "bf16", "bf16x3"], default="f32",
                    help="f32 (default; the reference's precision and the headline number), the bf16 conv-GEMM mode "
                         "(BASELINE configs[4]; separately stated tolerance) or the split-operand bf16x3 mode (fp32 gates, "
                         "three bf16 MFMAs per product); never the default")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    pkg = _pkgload.load_pkg()
    synth = pkg.synth
    cfg = synth.FULL
    W = synth.make_weights(cfg, 0)
    wpath = os.path.join(tempfile.gettempdir(), f"hipvoc_bench_full_r{rank}.safetensors")
    synth.save_safetensors(wpath, W, cfg.metadata())
    voc = pkg.HipVocoder.load(wpath, device=local_rank, unit_frames=64, dtype={"f32": 0, "bf16": 1, "bf16x3": 2}[args.dtype])
    os.remove(wpath)

    # --- synthetic workload: utterance u of rank r uses seed 1000 + r*U + u
    U = args.utterances
    T = voc.max_tokens
    units_per_utt = (FRAMES_PER_UTT + T - 1) // T
    n_units = U * units_per_utt
    codes = np.zeros((n_units, T, 16), dtype=np.int64)          # pad = code 0
    for u in range(U):
        c = synth.make_codes(FRAMES_PER_UTT, 1000 + rank * U + u, cfg)
        for k in range(units_per_utt):
            seg = c[k * T:(k + 1) * T]
            codes[u * units_per_utt + k, : seg.shape[0]] = seg
    stride = (voc.unit_samples + 63) // 64 * 64
    d_codes = torch.from_numpy(codes).to(dev)
    d_out = torch.empty((n_units, stride), dtype=torch.float32, device=dev)
    torch.cuda.synchronize()

    def step():
        voc.decode_units_device(d_codes.data_ptr(), n_units, d_out.data_ptr(), stride)

    def fence():
        voc.sync()
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
    audio_s = world * U * AUDIO_S_PER_UTT * args.steps
    value = audio_s / dt

    # --- single-utterance latency (configs[1]): host codes in, host waveform out, PCIe included
    lat_ms = None
    if rank == 0:
        c1 = synth.make_codes(FRAMES_PER_UTT, 1000, cfg)
        voc.synthesize(c1)
        ts = []
        for _ in range(5):
            t = time.perf_counter()
            voc.synthesize(c1)
            ts.append(time.perf_counter() - t)
        lat_ms = 1e3 * float(np.median(ts))

    # --- roofline of the dominant kernel: one profiled pass (hipEvents around every launch)
    roofline = None
    kernels = {}
    if rank == 0:
        voc.set_profile(True)
        G = voc.group_units
        voc.decode_units_device(d_codes.data_ptr(), G, d_out.data_ptr(), stride)
        voc.sync()
        prof = voc.last_profile()
        voc.set_profile(False)
        for key, (ms, flops, launches) in prof.items():
            kname = key.split("|")[1]
            k = kernels.setdefault(kname, {"ms": 0.0, "flops": 0.0, "launches": 0})
            k["ms"] += ms
            k["flops"] += flops
            k["launches"] += launches
        dom = max(kernels, key=lambda n: kernels[n]["ms"])
        kd = kernels[dom]
        achieved = kd["flops"] / (kd["ms"] * 1e-3) / 1e12
        traffic, traffic_src = pmc_traffic(dom, G)
        # bf16x3: three bf16 MFMAs per algorithmic product, so the roof of the ALGORITHMIC rate is a third of the bf16 peak
        peak = {"f32": PEAK_F32_MFMA_TFLOPS, "bf16": PEAK_BF16_MFMA_TFLOPS, "bf16x3": round(PEAK_BF16_MFMA_TFLOPS / 3, 1)}[args.dtype]
        roofline = {"bound": "mfma", "kernel": dom, "achieved": round(achieved, 3), "peak": peak,
                    "unit": "TFLOP/s", "frac": round(achieved / peak, 4), "traffic": traffic,
                    "traffic_source": traffic_src,
                    "avg_launch_ms": round(kd["ms"] / kd["launches"], 4), "launches_per_pass": kd["launches"],
                    "units_per_pass": G, "share_of_pass_time": round(kd["ms"] / sum(k["ms"] for k in kernels.values()), 3)}

    # --- HBM-bound kernels: algorithmic bytes (from the tensor shapes) / live-measured time, against 8 TB/s
    hbm = {}
    if rank == 0 and kernels:
        G = voc.group_units
        L = T
        dw_bytes = 0
        for r in cfg.upsampling_ratios:
            L *= r
            dw_bytes += 2 * cfg.latent_dim * L * 4 * G                 # x in, normalised y out
        Lw = voc.unit_samples
        alg = {
            "final_conv_kernel<7>": G * (cfg.decoder_dim // 16 * Lw + Lw) * 4,   # C=96 activations in, 1 channel out
            next((k for k in kernels if k.startswith("dwconv_ln")), "dwconv_ln_kernel"): dw_bytes,
            next((k for k in kernels if k.startswith("rmsnorm")), "rmsnorm_kernel"): (2 * cfg.num_hidden_layers + 1) * 2 * cfg.hidden_size * G * T * 4,
            "swiglu_kernel": cfg.num_hidden_layers * 3 * cfg.intermediate_size * G * T * 4,
        }
        for k, b in alg.items():
            if k in kernels and kernels[k]["ms"] > 0:
                gbs = b / (kernels[k]["ms"] * 1e-3) / 1e9
                hbm[k] = {"algorithmic_MB_per_pass": round(b / 1e6, 1), "achieved_GBps": round(gbs, 1),
                          "frac_of_8TBps": round(gbs / 8000.0, 3)}

    # --- CPU baseline + parity spot check (rank 0, N=1 only)
    cpu = None
    parity = None
    if rank == 0 and world == 1 and not args.no_cpu:
        vo = _pkgload.load_oracle()
        oracle = vo.Oracle(W, {k: float(v) for k, v in cfg.metadata().items()})
        cores = vo.set_threads(0)
        n_sample = 3
        t = time.perf_counter()
        refs = [oracle.run_unit(codes[i]) for i in range(n_sample)]
        cdt = time.perf_counter() - t
        # kept audio of these units: unit 0 and 2 are full (64 frames), unit 1 is a 61-frame tail
        kept = sum(min((FRAMES_PER_UTT - (i % units_per_utt) * T), T) * 1920 for i in range(n_sample)) / 24000.0
        cpu = {"value": round(kept / cdt, 3), "unit": "audio-sec/sec", "cores": cores, "kind": "port",
               "sample": f"{n_sample} reference units (64-frame, incl. one padded 61-frame tail) of utterances 0-1 "
                         f"of the same batch, oracle/voc_oracle.c with {cores} OpenMP threads, {cdt:.1f} s"}
        # the reference's own session uses intra_threads = clamp(nproc, 1, 4) (src/worker/vocoder.rs:138-141)
        vo.set_threads(4)
        t = time.perf_counter()
        oracle.run_unit(codes[0])
        cdt4 = time.perf_counter() - t
        vo.set_threads(0)
        cpu["value_4_threads"] = round(T * 1920 / 24000.0 / cdt4, 3)
        cpu["sample_4_threads"] = f"1 full reference unit, 4 OpenMP threads, {cdt4:.1f} s"
        got = d_out[:n_sample].cpu().numpy()[:, : voc.unit_samples]
        err = max(float(np.abs(got[i] - refs[i]).max()) for i in range(n_sample))
        num = sum(float(np.sum(refs[i].astype(np.float64) ** 2)) for i in range(n_sample))
        den = sum(float(np.sum((got[i].astype(np.float64) - refs[i]) ** 2)) for i in range(n_sample))
        parity = {"max_abs_err": err, "snr_db": round(10 * np.log10(num / max(den, 1e-300)), 1), "units_checked": n_sample}

    if rank == 0:
        unit_flops = 2.0 * voc.unit_macs
        line = {
            "metric": "vocoder audio-sec/sec (xRT)", "value": round(value, 2), "unit": "audio-sec/sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * dt / args.steps, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"configs[2]: batched decode of {U} synthetic 10 s code sequences per GPU "
                                   f"(125 frames x 16 codebooks -> 2 reference units of 64 frames each), "
                                   f"{MODE_TEXT[args.dtype]}, "
                                   f"seeded random-init weights of the Qwen3-TTS 12 Hz decoder (114 M params)",
                       "utterances_per_gpu": U, "units_per_gpu": n_units, "unit_frames": T,
                       "units_per_pass": voc.group_units},
            "computed_frames_per_s": round(world * n_units * T * args.steps / dt, 1),
            "gflop_per_unit": round(unit_flops / 1e9, 2),
            "achieved_tflops_whole_path": round(world * n_units * args.steps * unit_flops / dt / 1e12, 2),
            "single_utterance_latency_ms": None if lat_ms is None else round(lat_ms, 2),
            "roofline": roofline, "cpu_baseline": cpu, "parity_vs_oracle": parity, "hbm_bound_kernels": hbm,
            "kernels": {k: {"ms_per_pass": round(v["ms"], 3), "tflops": round(v["flops"] / max(v["ms"], 1e-9) / 1e9, 2),
                            "launches": v["launches"]} for k, v in sorted(kernels.items(), key=lambda kv: -kv[1]["ms"])},
        }
        print(json.dumps(line), flush=True)
    voc.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
