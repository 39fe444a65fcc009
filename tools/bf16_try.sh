#!/bin/bash
# Quick check of a bf16-mode kernel change: the bf16 / bf16x3 GPU tests, then per-step times of both modes (plus any env settings given).
set -o pipefail
D=gpurun_out/bf; mkdir -p $D
timeout -k 10 600 python3 -u -m pytest tests -x -q -m gpu --timeout 300 --timeout-method thread -k "bf16 or x3 or packed" > $D/pytest.txt 2>&1 || { tail -n 30 $D/pytest.txt; exit 1; }
tail -n 2 $D/pytest.txt
timeout -k 10 200 python3 tools/bf16_steps.py > $D/steps_on.txt 2>$D/steps_on.err || { tail -n 20 $D/steps_on.err; exit 1; }
grep -E "G=64|transconv|final" $D/steps_on.txt
timeout -k 10 200 python3 tools/bf16_steps.py x3 > $D/steps_x3_on.txt 2>$D/steps_x3_on.err || { tail -n 20 $D/steps_x3_on.err; exit 1; }
grep -E "G=64|transconv|final" $D/steps_x3_on.txt
for v in "$@"; do
  n=$(echo "$v" | tr -c 'A-Za-z0-9=\n' '_')
  env $v timeout -k 10 200 python3 tools/bf16_steps.py > $D/steps_$n.txt 2>$D/steps_$n.err || { tail -n 20 $D/steps_$n.err; exit 1; }
  echo "$v"; grep -E "G=64|transconv|final" $D/steps_$n.txt
done
