#!/usr/bin/env python3
"""bench.py -- vocoder throughput on MI355X (BASELINE.json metric: audio-seconds decoded per second).

    python bench.py --gpus N --steps K --warmup W [--workload batch256|longform|batch4096] [--dtype f32|bf16|bf16x3]

A "step" is one pass of the hot path over the workload.  Codes and the output buffer are resident in HBM when the timed
region starts.  Workloads (BASELINE.json `configs`):

  batch256  (default; configs[2])  per GPU 256 synthetic 10 s utterances (125 frames x 16 codebooks = 2 reference units of
            64 frames: 64 + 61 padded with code 0, src/worker/vocoder.rs:60-85).  Weak scaling.
  longform  (configs[3])  ONE 10-minute stream (7 500 frames = 118 reference units: 117 full + a 12-frame tail) split into
            contiguous unit ranges, one per GPU (qwen3-tts_amd/sharding.py); each step decodes the shards and gathers them
            on rank 0 (torch.distributed gather of device buffers: RCCL with the nccl backend).  After the timed region
            rank 0 stitches the kept samples, decodes the whole stream alone and compares the two bit for bit
            (`stitching`).  Strong scaling.
  batch4096 (configs[4])  the 8-GPU node batch of 4 096 utterances = 512 per GPU, bf16 conv-GEMM mode by default.  Weak.

Units are independent in the reference's semantics (vocoder.rs:65-82), so no workload has a data-path collective; the
long-form gather only assembles finished samples.

N > 1: started either by the driver (`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`, one rank
per GPU; RANK / LOCAL_RANK / WORLD_SIZE from the environment) or directly (`python bench.py --gpus N`): then this process
touches no GPU, checks that N devices exist, starts that same launcher as a CHILD process and exits with its code.

Besides the contract fields the JSON line carries
  roofline            : the dominant kernel (by device time), algorithmic FLOPs / live-measured duration (hipEvents on the
                        library's stream) against the MFMA peak of the mode
  cpu_baseline        : the CPU oracle (oracle/voc_oracle.c, kind "port") timed on this box's host cores on a bounded
                        sample of the same workload (rank 0, N=1 only)
  value_host_to_host  : the same utterances through hipvoc_decode_batch -- pageable host codes in, pageable host samples
                        out, PCIe both ways; never `value`
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import _pkgload  # noqa: E402

FRAMES_PER_UTT = 125            # 10.0 s at 12.5 Hz
AUDIO_S_PER_UTT = FRAMES_PER_UTT * 1920 / 24000.0
LONGFORM_FRAMES = 7500          # 10 minutes
LONGFORM_SEED = 4242
PEAK_F32_MFMA_TFLOPS = 157.3    # MI355X_MICROARCH.md, "Peak FP32 (matrix)"
PEAK_BF16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md, dense bf16
MODE_TEXT = {
    "f32": "fp32",
    "bf16": "bf16 conv-GEMM mode (fp32 accumulate; parity_vs_oracle is against the fp32 oracle; stated tolerance of the mode: "
            "SNR >= 30 dB, max-abs <= 3e-2 -- set after measurement, looser than SURVEY 8d's proposed 35 dB / 2e-2; the fp32 "
            "gates 1e-3 / 60 dB do not apply)",
    "bf16x3": "bf16x3 split-operand mode (operands as hi + mid bf16 planes, three bf16 MFMAs per product, fp32 accumulate; "
              "held to the fp32 gates: parity_vs_oracle is against the fp32 oracle)",
}
WORKLOADS = {"batch256": (256, "f32"), "batch4096": (512, "bf16"), "longform": (None, "f32")}


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


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="batch256")
    ap.add_argument("--utterances", type=int, default=None, help="utterances per GPU of the batch workloads (default: the workload's)")
    ap.add_argument("--frames", type=int, default=LONGFORM_FRAMES, help="frames of the long-form stream")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-host", action="store_true", help="skip the host-to-host leg")
    ap.add_argument("--no-latency", action="store_true", help="skip the single-utterance latency leg (profiling runs: keeps small launches out of the per-kernel averages)")
    ap.add_argument("--dtype", choices=["f32", "bf16", "bf16x3"], default=None,
                    help="f32 (the reference's precision and the headline number; default of batch256 / longform), the bf16 "
                         "conv-GEMM mode (default of batch4096, BASELINE configs[4]; separately stated tolerance) or the "
                         "split-operand bf16x3 mode (fp32 gates, three bf16 MFMAs per product)")
    ap.add_argument("--selftest-stub-decode", action="store_true",
                    help="TEST ONLY (tests/test_bench_cpu.py): no GPU, gloo, a fake decode -- exercises the launcher, sharding, "
                         "gather and stitching logic; the printed line says SELFTEST and is no measurement")
    args = ap.parse_args(argv)
    if args.dtype is None:
        args.dtype = WORKLOADS[args.workload][1]
    if args.utterances is None:
        args.utterances = WORKLOADS[args.workload][0]
    if args.gpus < 1 or args.steps < 1 or args.warmup < 0:
        ap.error("--gpus and --steps must be positive")
    return args


def launch_ranks(args, argv):
    """`python bench.py --gpus N` with no WORLD_SIZE: start N ranks as a CHILD process tree (never re-exec: nothing here
    has touched a GPU, and nothing will) and leave with its exit code."""
    if not args.selftest_stub_decode:
        import torch
        have = torch.cuda.device_count()          # counts devices without initialising the runtime
        if have < args.gpus:
            sys.stderr.write(f"bench.py: --gpus {args.gpus} asked for, {have} HIP device(s) visible: refusing to print a line "
                             f"that says n_gpus={args.gpus}\n")
            return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    return subprocess.call(cmd)


class StubVocoder:
    """TEST ONLY stand-in for HipVocoder (--selftest-stub-decode): a cheap deterministic function of the codes with the real
    shapes, so that the multi-rank plumbing can run on CPU under gloo.  Never used in a measurement."""
    max_tokens = 8
    unit_samples = 8 * 1920 - 555
    group_units = 4
    unit_macs = 1

    def decode_units(self, codes):          # [n, T, 16] int64 -> [n, stride] float32
        import torch
        n = codes.shape[0]
        stride = (self.unit_samples + 63) // 64 * 64
        t = torch.arange(stride, dtype=torch.float32)[None, :]
        key = (codes.to(torch.float64) * torch.arange(1, 1 + codes.shape[1] * 16, dtype=torch.float64).reshape(1, -1, 16)).sum((1, 2))
        return torch.sin(t * 1e-3 + (key % 1000.0).to(torch.float32)[:, None]).contiguous() if n else torch.zeros((0, stride))


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args, argv))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.stderr.write(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: start one rank per GPU "
                         f"(--nproc-per-node {args.gpus}) or run `python bench.py --gpus {args.gpus}`\n")
        sys.exit(2)
    stub = args.selftest_stub_decode
    import torch
    dist = None
    if stub:
        dev = torch.device("cpu")
        if world > 1:
            import torch.distributed as dist
            dist.init_process_group("gloo")
    else:
        if torch.cuda.device_count() <= local_rank:
            sys.stderr.write(f"bench.py: rank {rank} wants device {local_rank}, {torch.cuda.device_count()} visible\n")
            sys.exit(2)
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
        if world > 1:
            import torch.distributed as dist
            dist.init_process_group("nccl", device_id=dev)

    pkg = _pkgload.load_pkg()
    synth = pkg.synth
    sh = pkg.sharding
    cfg = synth.FULL
    W = None
    if stub:
        voc = StubVocoder()
    else:
        W = synth.make_weights(cfg, 0)
        wpath = os.path.join(tempfile.gettempdir(), f"hipvoc_bench_full_r{rank}.safetensors")
        synth.save_safetensors(wpath, W, cfg.metadata())
        voc = pkg.HipVocoder.load(wpath, device=local_rank, unit_frames=64, dtype={"f32": 0, "bf16": 1, "bf16x3": 2}[args.dtype])
        os.remove(wpath)
    T = voc.max_tokens
    Lu = voc.unit_samples
    stride = (Lu + 63) // 64 * 64

    # ---------------------------------------------------------------- workload -> this rank's units
    longform = args.workload == "longform"
    utt_codes = []                                              # batch workloads: this rank's utterances (host-to-host leg)
    if longform:
        stream = synth.make_codes(args.frames, LONGFORM_SEED, cfg)
        spans = sh.unit_spans(args.frames, T)
        lo, hi = sh.shard_range(len(spans), rank, world)        # contiguous unit range of this rank
        my_spans = spans[lo:hi]
        n_units = hi - lo
        codes = np.zeros((max(n_units, 1), T, 16), dtype=np.int64)          # pad = code 0 (vocoder.rs:62,70)
        for i, (a, b) in enumerate(my_spans):
            codes[i, : b - a] = stream[a:b]
        keep = [min((b - a) * 1920, Lu) for a, b in spans]     # vocoder.rs:79-80, all units (rank 0 stitches)
        audio_s_per_step = args.frames * 1920 / 24000.0          # whole job, strong scaling
    else:
        U = args.utterances
        units_per_utt = (FRAMES_PER_UTT + T - 1) // T
        n_units = U * units_per_utt
        codes = np.zeros((n_units, T, 16), dtype=np.int64)
        for u in range(U):                                      # utterance u of rank r uses seed 1000 + r*U + u
            c = synth.make_codes(FRAMES_PER_UTT, 1000 + rank * U + u, cfg)
            utt_codes.append(c)
            for k in range(units_per_utt):
                seg = c[k * T:(k + 1) * T]
                codes[u * units_per_utt + k, : seg.shape[0]] = seg
        audio_s_per_step = world * U * AUDIO_S_PER_UTT
    d_codes = torch.from_numpy(codes).to(dev)
    d_out = torch.empty((max(n_units, 1), stride), dtype=torch.float32, device=dev)
    gatherer = sh.UnitGather(len(spans), stride, rank, world, dist, dev) if longform and world > 1 else None

    def sync_dev():
        if not stub:
            voc.sync()
            torch.cuda.synchronize()

    def step():
        if stub:
            d_out[:n_units] = voc.decode_units(d_codes[:n_units])
        elif n_units:
            voc.decode_units_device(d_codes.data_ptr(), n_units, d_out.data_ptr(), stride)
        if longform and world > 1:
            # the library decodes on its own stream: finish it, then gather the shards on rank 0
            sync_dev()
            gatherer.gather(d_out, n_units)

    def fence():
        sync_dev()
        if dist is not None:
            dist.barrier()

    sync_dev()
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
    value = audio_s_per_step * args.steps / dt

    # ---------------------------------------------------------------- long-form: stitch, and compare with a 1-rank decode
    stitching = None
    if longform and rank == 0:
        if world > 1:
            stitched = gatherer.stitch(keep)
        else:
            g = d_out.cpu().numpy()
            stitched = np.concatenate([g[i, : keep[i]] for i in range(n_units)])
        if stub:
            all_codes = np.zeros((len(spans), T, 16), dtype=np.int64)
            for i, (a, b) in enumerate(spans):
                all_codes[i, : b - a] = stream[a:b]
            w = voc.decode_units(torch.from_numpy(all_codes)).numpy()
            whole = np.concatenate([w[i, : keep[i]] for i in range(len(spans))])
        else:
            whole = voc.synthesize(stream)                      # this GPU alone, host API, ordinary chunk loop
        stitching = {"units": len(spans), "samples": int(stitched.size),
                     "sha256_stitched": hashlib.sha256(stitched.tobytes()).hexdigest(),
                     "sha256_single_gpu": hashlib.sha256(whole.tobytes()).hexdigest(),
                     "bit_identical_to_single_gpu_decode": bool(stitched.size == whole.size and np.array_equal(stitched.view(np.uint32), whole.view(np.uint32))),
                     "units_per_rank": [sh.shard_range(len(spans), r, world)[1] - sh.shard_range(len(spans), r, world)[0] for r in range(world)],
                     "gather": None if world == 1 else "torch.distributed.gather of device buffers, inside the timed step"}

    # ---------------------------------------------------------------- host to host (PCIe both ways): never `value`
    h2h = None
    if not stub and not longform and not args.no_host:
        # the C ABI as a host that reuses its buffers calls it: pageable numpy arrays, touched once beforehand (a fresh
        # 245 MB allocation per call would time the kernel's page faults, not the library)
        import ctypes as C
        Uh = len(utt_codes)
        h_codes = np.ascontiguousarray(np.concatenate(utt_codes, axis=0))
        h_nf = np.full(Uh, FRAMES_PER_UTT, dtype=np.int32)
        h_off = (np.arange(Uh, dtype=np.uint64) * np.uint64(FRAMES_PER_UTT * 1920))
        h_out = np.zeros(Uh * FRAMES_PER_UTT * 1920, dtype=np.float32)
        h_n = np.zeros(Uh, dtype=np.uint64)

        def host_step():
            rc = voc._L.hipvoc_decode_batch(voc._h, h_codes.ctypes.data, h_nf.ctypes.data, Uh, h_out.ctypes.data,
                                            h_off.ctypes.data, h_n.ctypes.data)
            assert rc == 0, voc._L.hipvoc_last_error(voc._h)

        host_step()
        fence()
        k_h = min(args.steps, 3)
        t0 = time.perf_counter()
        for _ in range(k_h):
            host_step()
        dth = time.perf_counter() - t0           # (the call returns with the samples in host memory: no fence needed)
        if dist is not None:
            tt = torch.tensor([dth], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dth = float(tt.item())
        h2h = audio_s_per_step * k_h / dth
        assert int(h_n.sum()) == Uh * (Lu + (FRAMES_PER_UTT - T) * 1920)
        # and it is the same waveform, bit for bit, as the device-resident entry point produced
        first = d_out[0].cpu().numpy()[:Lu]
        assert np.array_equal(first.view(np.uint32), h_out[:Lu].view(np.uint32))

    # ---------------------------------------------------------------- single-utterance latency (configs[1]), PCIe included
    lat_ms = None
    if rank == 0 and not stub and not args.no_latency:
        c1 = synth.make_codes(FRAMES_PER_UTT, 1000, cfg)
        voc.synthesize(c1)
        ts = []
        for _ in range(5):
            t = time.perf_counter()
            voc.synthesize(c1)
            ts.append(time.perf_counter() - t)
        lat_ms = 1e3 * float(np.median(ts))

    # ---------------------------------------------------------------- roofline of the dominant kernel: one profiled pass
    roofline = None
    by_kernel = []
    kernels = {}
    exec_share_path = None
    peak = None
    if rank == 0 and not stub:
        voc.set_profile(True)
        G = min(voc.group_units, max(n_units, 1))
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
        # bf16x3: three bf16 MFMAs per algorithmic product, so the roof of the ALGORITHMIC rate is a third of the bf16 peak
        peak = {"f32": PEAK_F32_MFMA_TFLOPS, "bf16": PEAK_BF16_MFMA_TFLOPS, "bf16x3": round(PEAK_BF16_MFMA_TFLOPS / 3, 1)}[args.dtype]
        total_ms = sum(k["ms"] for k in kernels.values())

        def executed_share(name):
            """Fraction of a symbol's algorithmic (direct-form) multiplications that its matrix cores execute."""
            if name.startswith("wino_ps_kernel") and name.endswith("true>"):
                # a whole residual unit (conv7 by F(2,7) + the 1x1 conv): of the direct form's 7 + 1 tap-products, 4 + 1
                return 5 / 8, "a residual unit whose k=7 conv runs as Winograd F(2,7): 5/8 of the direct-form multiplications"
            if name.startswith("wino_unit_kernel"):
                return 5 / 8, "a residual unit whose k=7 conv runs as Winograd F(2,7): 5/8 of the direct-form multiplications"
            if name.startswith("wino_conv7_kernel") or name.startswith("wino_ps_kernel"):
                return 4 / 7, "a Winograd F(2,7) convolution: 8 multiplications where the direct form has 14"
            if name.startswith("wino_tconv_kernel"):
                return 4 / 6, "a Winograd F(3,2) transposed convolution (k = 2*stride): 4 multiplications where the direct form has 6"
            return 1.0, None

        def roof(name):
            # `frac` is the EXECUTED matrix-core rate over the peak (<= 1 by construction); for the Winograd symbols the
            # direct-form (algorithmic) rate, which can exceed the peak, is reported beside it as algorithmic_*
            kd = kernels[name]
            algorithmic = kd["flops"] / (kd["ms"] * 1e-3) / 1e12
            share, why = executed_share(name)
            achieved = algorithmic * share
            traffic, traffic_src = pmc_traffic(name, G)
            r = {"bound": "mfma", "kernel": name, "achieved": round(achieved, 3), "peak": peak, "unit": "TFLOP/s",
                 "frac": round(achieved / peak, 4), "traffic": traffic, "traffic_source": traffic_src,
                 "avg_launch_ms": round(kd["ms"] / kd["launches"], 4), "launches_per_pass": kd["launches"],
                 "units_per_pass": G, "share_of_pass_time": round(kd["ms"] / total_ms, 3),
                 "algorithmic_tflops": round(algorithmic, 3), "algorithmic_over_peak": round(algorithmic / peak, 4)}
            if why:
                r["note"] = (f"achieved / frac = multiplications the matrix cores EXECUTE / time; the kernel is {why}; "
                             "algorithmic_* counts the direct form's 2*MAC (SURVEY 8d) and may exceed the peak")
            return r

        dom = max(kernels, key=lambda n: kernels[n]["ms"])
        roofline = roof(dom)
        # share of the whole path's algorithmic FLOPs that the matrix cores execute (1.0 where no Winograd form runs)
        tot_flops = sum(k["flops"] for k in kernels.values())
        exec_share_path = sum(k["flops"] * executed_share(n)[0] for n, k in kernels.items()) / tot_flops if tot_flops else None
        by_kernel = [roof(n) for n in sorted(kernels, key=lambda n: -kernels[n]["ms"]) if kernels[n]["ms"] / total_ms >= 0.03 and kernels[n]["flops"] > 0]

    # ---------------------------------------------------------------- HBM-bound kernels: algorithmic bytes / live time
    hbm = {}
    if rank == 0 and kernels:
        G = roofline["units_per_pass"]
        L = T
        dw_bytes = 0
        for r in cfg.upsampling_ratios:
            L *= r
            dw_bytes += 2 * cfg.latent_dim * L * 4 * G                 # x in, normalised y out
        alg = {
            # C = 96 activations in (fp32, or the bf16 mode's packed bf16 tensor), 1 channel out
            next((k for k in kernels if k.startswith("final_conv")), "final_conv_kernel<7,false>"):
                G * (cfg.decoder_dim // 16 * Lu * (2 if any(k.startswith("final_conv_packed") for k in kernels) else 4) + Lu * 4),
            next((k for k in kernels if k.startswith("dwconv_ln")), "dwconv_ln_kernel"): dw_bytes,
            next((k for k in kernels if k.startswith("rmsnorm")), "rmsnorm_kernel"): (2 * cfg.num_hidden_layers + 1) * 2 * cfg.hidden_size * G * T * 4,
            "swiglu_kernel": cfg.num_hidden_layers * 3 * cfg.intermediate_size * G * T * 4,
        }
        for k, b in alg.items():
            if k in kernels and kernels[k]["ms"] > 0:
                gbs = b / (kernels[k]["ms"] * 1e-3) / 1e9
                hbm[k] = {"algorithmic_MB_per_pass": round(b / 1e6, 1), "achieved_GBps": round(gbs, 1),
                          "frac_of_8TBps": round(gbs / 8000.0, 3)}

    # ---------------------------------------------------------------- CPU baseline + parity spot check (rank 0, N=1 only)
    cpu = None
    parity = None
    if rank == 0 and world == 1 and not args.no_cpu and not stub:
        vo = _pkgload.load_oracle()
        oracle = vo.Oracle(W, {k: float(v) for k, v in cfg.metadata().items()})
        cores = vo.set_threads(0)
        n_sample = min(3, n_units)
        step()                                               # (the host-to-host leg does not touch d_out; be explicit anyway)
        sync_dev()
        t = time.perf_counter()
        refs = [oracle.run_unit(codes[i]) for i in range(n_sample)]
        cdt = time.perf_counter() - t
        if longform:
            kept = sum(keep[i] for i in range(n_sample)) / 24000.0
            what = f"the first {n_sample} reference units (64 frames) of the same stream"
        else:
            # kept audio of these units: unit 0 and 2 are full (64 frames), unit 1 is a 61-frame tail
            kept = sum(min((FRAMES_PER_UTT - (i % units_per_utt) * T), T) * 1920 for i in range(n_sample)) / 24000.0
            what = f"{n_sample} reference units (64-frame, incl. one padded 61-frame tail) of utterances 0-1 of the same batch"
        cpu = {"value": round(kept / cdt, 3), "unit": "audio-sec/sec", "cores": cores, "kind": "port",
               "sample": f"{what}, oracle/voc_oracle.c with {cores} OpenMP threads, {cdt:.1f} s"}
        # the reference's own session uses intra_threads = clamp(nproc, 1, 4) (src/worker/vocoder.rs:138-141)
        vo.set_threads(4)
        t = time.perf_counter()
        oracle.run_unit(codes[0])
        cdt4 = time.perf_counter() - t
        vo.set_threads(0)
        cpu["value_4_threads"] = round(T * 1920 / 24000.0 / cdt4, 3)
        cpu["sample_4_threads"] = f"1 full reference unit, 4 OpenMP threads, {cdt4:.1f} s"
        got = d_out[:n_sample].cpu().numpy()[:, :Lu]
        err = max(float(np.abs(got[i] - refs[i]).max()) for i in range(n_sample))
        num = sum(float(np.sum(refs[i].astype(np.float64) ** 2)) for i in range(n_sample))
        den = sum(float(np.sum((got[i].astype(np.float64) - refs[i]) ** 2)) for i in range(n_sample))
        parity = {"max_abs_err": err, "snr_db": round(10 * np.log10(num / max(den, 1e-300)), 1), "units_checked": n_sample}

    if rank == 0:
        unit_flops = 2.0 * voc.unit_macs
        if longform:
            total_units = len(spans)
            workload = (f"configs[3]: one {args.frames}-frame stream ({args.frames * 0.08:.0f} s) = {total_units} reference units of 64 frames "
                        f"(the last holds {args.frames - (total_units - 1) * T} frames), contiguous unit ranges over {world} GPU(s), "
                        f"gathered and stitched on rank 0")
            conf = {"workload": workload, "stream_frames": args.frames, "units": total_units}
        else:
            total_units = world * n_units
            name = "configs[4]: the 8-GPU node batch of 4096 utterances = 512 per GPU" if args.workload == "batch4096" else "configs[2]"
            workload = (f"{name}: batched decode of {args.utterances} synthetic 10 s code sequences per GPU "
                        f"(125 frames x 16 codebooks -> 2 reference units of 64 frames each)")
            conf = {"workload": workload, "utterances_per_gpu": args.utterances, "units_per_gpu": n_units}
        conf["workload"] += f", {MODE_TEXT[args.dtype]}, seeded random-init weights of the Qwen3-TTS 12 Hz decoder (114 M params)"
        conf.update({"unit_frames": T, "units_per_pass": voc.group_units, "lanes": getattr(voc, "lanes", 1)})
        if getattr(voc, "lanes", 1) > 1 and roofline:
            roofline["note_lanes"] = ("the timed steps run with two passes in flight (HIPVOC_LANES=2: alternate passes on two streams, a workspace each); "
                                      "the per-kernel durations here are one lane's, from one profiled pass -- compare with a rocprofv3 summary taken at HIPVOC_LANES=1")
        line = {
            "metric": "vocoder audio-sec/sec (xRT)", "value": round(value, 2), "unit": "audio-sec/sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * dt / args.steps, 3), "higher_is_better": True,
            "scaling": "strong" if longform else "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": conf,
            "computed_frames_per_s": round(total_units * T * args.steps / dt, 1),
            "gflop_per_unit": round(unit_flops / 1e9, 2),
            "achieved_tflops_whole_path": round(total_units * args.steps * unit_flops / dt / 1e12, 2),
            "achieved_executed_tflops_whole_path": None if exec_share_path is None else round(total_units * args.steps * unit_flops * exec_share_path / dt / 1e12, 2),
            "executed_frac_whole_path": None if exec_share_path is None else round(total_units * args.steps * unit_flops * exec_share_path / dt / 1e12 / peak, 4),
            "value_host_to_host": None if h2h is None else round(h2h, 2),
            "host_to_host_over_value": None if h2h is None else round(h2h / value, 4),
            "single_utterance_latency_ms": None if lat_ms is None else round(lat_ms, 2),
            "stitching": stitching,
            "roofline": roofline, "roofline_by_kernel": by_kernel, "cpu_baseline": cpu, "parity_vs_oracle": parity, "hbm_bound_kernels": hbm,
            "kernels": {k: {"ms_per_pass": round(v["ms"], 3), "tflops": round(v["flops"] / max(v["ms"], 1e-9) / 1e9, 2),
                            "launches": v["launches"]} for k, v in sorted(kernels.items(), key=lambda kv: -kv[1]["ms"])},
        }
        if stub:
            line["metric"] = "SELFTEST (stub decode on CPU: no measurement)"
            line["data"] = "SELFTEST"
        print(json.dumps(line), flush=True)
    if not stub:
        voc.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
