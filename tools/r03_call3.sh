#!/bin/bash
# margin tests, bf16 SQ counters, scaling projection (run on the GPU box from the repo root)
mkdir -p gpurun_out/r03
tools/gpu_tests.sh -k "margin_over or winograd_costs" && cp gpurun_out/r03/pytest.txt gpurun_out/r03/pytest_margin.txt && \
tools/prof_sq_one.sh bf16 bf16_r03a && \
timeout -k 10 900 python3 tools/scaling_projection.py > gpurun_out/r03/scaling_projection.txt 2> gpurun_out/r03/scaling_projection.err
