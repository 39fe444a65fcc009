#!/bin/bash
for d in 0 7 15; do
  HIPVOC_DBG=$d timeout -k 10 200 python3 tools/profile_steps.py > gpurun_out/steps_wino_dbg$d.txt 2>/dev/null || exit 1
  echo "DBG=$d: $(grep -E 'b1_conv7\|wino_conv7_kernel<1' gpurun_out/steps_wino_dbg$d.txt) $(grep -E 'b1_conv7\|wino_conv7_kernel<9' gpurun_out/steps_wino_dbg$d.txt)"
done
