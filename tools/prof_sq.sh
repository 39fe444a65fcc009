#!/bin/bash
# one SQ counter pass per mode (run on the GPU box from the repo root)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
C="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT"
for leg in f32:profile_steps.py bf16:bf16_steps.py enc:enc_steps.py; do
  name=${leg%%:*}; script=${leg##*:}
  rm -rf $O/prof_sq_$name
  timeout -s KILL 400 rocprofv3 --pmc $C --output-format csv -d $O/prof_sq_$name -o sq -- python3 $R/tools/$script > $O/sq_$name.out 2>&1 || exit 1
  python3 $R/tools/pmc_sq.py $O/prof_sq_$name/sq_counter_collection.csv > $O/sq_$name.txt
  rm -rf $O/prof_sq_$name
done
