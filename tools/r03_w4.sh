#!/bin/bash
mkdir -p gpurun_out/r03
tools/gpu_tests.sh -k "4_wave or block3" && \
HIPVOC_BF16_W4=0 timeout -k 10 200 python3 tools/bf16_steps.py > gpurun_out/r03/steps_bf16_w8.txt 2>/dev/null && \
timeout -k 10 200 python3 tools/bf16_steps.py > gpurun_out/r03/steps_bf16_w4.txt 2>/dev/null && \
HIPVOC_BLK3=0 timeout -k 10 200 python3 tools/bf16_steps.py > gpurun_out/r03/steps_bf16_w4_noblk3.txt 2>/dev/null && \
for f in w8 w4 w4_noblk3; do echo "== $f"; head -n 4 gpurun_out/r03/steps_bf16_$f.txt; done
