#!/bin/bash
# the fused block kernel: bitwise tests, then per-step times (run on the GPU box from the repo root)
mkdir -p gpurun_out/r03
tools/gpu_tests.sh -k "block3" && cp gpurun_out/r03/pytest.txt gpurun_out/r03/pytest_blk3.txt && \
timeout -k 10 200 python3 tools/bf16_steps.py > gpurun_out/r03/steps_bf16_blk3.txt 2> gpurun_out/r03/steps_bf16_blk3.err && \
head -n 12 gpurun_out/r03/steps_bf16_blk3.txt
