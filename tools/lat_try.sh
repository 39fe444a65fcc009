#!/bin/bash
# Quick check of a small-launch kernel change: the bitwise tests that cover it, then the single-utterance latency
# breakdown with the change on and off.  Output: gpurun_out/lat/.
set -o pipefail
D=gpurun_out/lat; mkdir -p $D
timeout -k 10 500 python3 -u -m pytest tests/test_gpu_parity.py -x -q -m gpu --timeout 200 --timeout-method thread -k "gemm_dma_bitwise or gemm1_bitwise or winograd_conv7_against or fp32_gates or post_snake or parity" > $D/pytest.txt 2>&1 || { tail -n 30 $D/pytest.txt; exit 1; }
tail -n 3 $D/pytest.txt
timeout -k 10 200 python3 tools/latency_breakdown.py > $D/lat_on.txt 2>$D/lat_on.err || { tail -n 20 $D/lat_on.err; exit 1; }
grep "^dtype" $D/lat_on.txt
for v in "$@"; do
  n=$(echo "$v" | tr -c 'A-Za-z0-9=\n' '_')
  env $v timeout -k 10 200 python3 tools/latency_breakdown.py > $D/lat_$n.txt 2>$D/lat_$n.err || { tail -n 20 $D/lat_$n.err; exit 1; }
  echo "$v"; grep "^dtype=0" $D/lat_$n.txt
done
