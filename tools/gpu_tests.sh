#!/bin/bash
# The GPU suite in one process, each test under a time limit of its own (run on the GPU box from the repo root):
#   tools/gpu_tests.sh [pytest selection args]      -> gpurun_out/r03/pytest.txt
set -o pipefail
D=gpurun_out/r03; mkdir -p $D
timeout -k 10 1100 python3 -u -m pytest tests -x -v -s -m gpu --timeout 400 --timeout-method thread "$@" > $D/pytest.txt 2>&1
rc=$?
grep -E "passed|failed|error" $D/pytest.txt | tail -n 3
[ $rc -eq 0 ] || { grep -n -E "FAILED|Error|error:|assert" $D/pytest.txt | head -n 30; tail -n 40 $D/pytest.txt; }
exit $rc
