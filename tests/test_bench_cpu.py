"""bench.py's multi-rank plumbing on CPU: `--gpus N` must really start N ranks (the flag used to be ignored
and the line said n_gpus 1), a long-form stream must be sharded, gathered and stitched bit-identically, and asking for
more GPUs than exist must fail loudly instead of printing a line.  The decode itself is a stub here
(--selftest-stub-decode, gloo): what is under test is the launcher, sharding.py, the gather and the stitching."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def run(args, env=None, timeout=300):
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    e["OMP_NUM_THREADS"] = "2"
    e.update(env or {})
    return subprocess.run([sys.executable, BENCH] + args, capture_output=True, text=True, timeout=timeout, env=e)


def json_lines(out):
    return [json.loads(ln) for ln in out.splitlines() if ln.startswith("{")]


def test_gpus_flag_starts_that_many_ranks_and_longform_stitches_bit_identically():
    r = run(["--gpus", "2", "--selftest-stub-decode", "--workload", "longform", "--frames", "100", "--steps", "2", "--warmup", "1"])
    assert r.returncode == 0, r.stdout + r.stderr
    lines = json_lines(r.stdout)
    assert len(lines) == 1, r.stdout                       # rank 0 prints ONE line
    ln = lines[0]
    assert ln["n_gpus"] == 2 and ln["scaling"] == "strong" and ln["steps"] == 2 and ln["warmup"] == 1
    assert ln["metric"].startswith("SELFTEST") and ln["data"] == "SELFTEST"
    st = ln["stitching"]
    assert st["units"] == 13 and st["units_per_rank"] == [7, 6]          # 100 frames / 8-frame stub units: 12 full + a 4-frame tail
    assert st["samples"] == 12 * (8 * 1920 - 555) + 4 * 1920
    assert st["bit_identical_to_single_gpu_decode"] and st["sha256_stitched"] == st["sha256_single_gpu"]
    assert "configs[3]" in ln["config"]["workload"]


def test_single_rank_line_and_weak_batch_workloads():
    r = run(["--selftest-stub-decode", "--utterances", "3", "--steps", "1", "--warmup", "0"])
    assert r.returncode == 0, r.stdout + r.stderr
    ln = json_lines(r.stdout)[0]
    assert ln["n_gpus"] == 1 and ln["scaling"] == "weak" and "configs[2]" in ln["config"]["workload"]
    assert ln["config"]["units_per_gpu"] == 3 * 16                        # 125 frames / 8-frame stub units
    r = run(["--gpus", "2", "--selftest-stub-decode", "--workload", "batch4096", "--utterances", "2", "--steps", "1", "--warmup", "0"])
    assert r.returncode == 0, r.stdout + r.stderr
    ln = json_lines(r.stdout)[0]
    assert ln["n_gpus"] == 2 and ln["dtype"] == "bf16" and "configs[4]" in ln["config"]["workload"]
    # whole-job aggregate: both ranks' audio over the slowest rank's time
    assert abs(ln["value"] * ln["ms_per_step"] * 1e-3 - 2 * 2 * 10.0) < 0.01 * 40


def test_more_gpus_than_devices_fails_loudly():
    import torch
    if torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("two or more GPUs here")
    r = run(["--gpus", "2", "--steps", "1", "--warmup", "0"])
    assert r.returncode != 0 and not json_lines(r.stdout)
    assert "--gpus 2" in r.stderr and "device" in r.stderr


def test_world_size_that_disagrees_with_gpus_is_refused():
    r = run(["--gpus", "2", "--selftest-stub-decode"], env={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and not json_lines(r.stdout) and "WORLD_SIZE" in r.stderr
