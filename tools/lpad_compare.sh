#!/bin/bash
# row padding 4 vs 32 floats, three modes (run on the GPU box from the repo root)
O=gpurun_out
for lp in 4 32; do
  HIPVOC_LPAD=$lp timeout -k 10 200 python3 tools/profile_steps.py > $O/steps_f32_lp$lp.txt 2>/dev/null || exit 1
  HIPVOC_LPAD=$lp timeout -k 10 200 python3 tools/bf16_steps.py > $O/steps_bf16_lp$lp.txt 2>/dev/null || exit 1
  HIPVOC_LPAD=$lp timeout -k 10 200 python3 tools/bf16_steps.py x3 > $O/steps_x3_lp$lp.txt 2>/dev/null || exit 1
  for m in f32 bf16 x3; do echo "LPAD=$lp $m: $(sed -n 1p $O/steps_${m}_lp$lp.txt)"; done
done
