#!/bin/bash
# Quick check of an fp32 kernel change: the fp32 parity tests, then per-step times at 64 units (plus any env settings given).
set -o pipefail
D=gpurun_out/f32; mkdir -p $D
timeout -k 10 600 python3 -u -m pytest tests/test_gpu_parity.py -x -q -m gpu --timeout 300 --timeout-method thread -k "not bf16 and not x3 and not packed" > $D/pytest.txt 2>&1 || { tail -n 30 $D/pytest.txt; exit 1; }
tail -n 2 $D/pytest.txt
timeout -k 10 200 python3 tools/profile_steps.py > $D/steps_on.txt 2>$D/steps_on.err || { tail -n 20 $D/steps_on.err; exit 1; }
grep -E "G=64|${PAT:-transconv}" $D/steps_on.txt
for v in "$@"; do
  n=$(echo "$v" | tr -c 'A-Za-z0-9=\n' '_')
  env $v timeout -k 10 200 python3 tools/profile_steps.py > $D/steps_$n.txt 2>$D/steps_$n.err || { tail -n 20 $D/steps_$n.err; exit 1; }
  echo "$v"; grep -E "G=64|${PAT:-transconv}" $D/steps_$n.txt
done
