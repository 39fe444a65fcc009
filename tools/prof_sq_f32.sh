#!/bin/bash
# one SQ counter pass of the fp32 per-step profile (run on the GPU box from the repo root)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
C="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT"
rm -rf $O/prof_sq_f32
timeout -s KILL 400 rocprofv3 --pmc $C --output-format csv -d $O/prof_sq_f32 -o sq -- python3 $R/tools/profile_steps.py > $O/sq_f32.out 2>&1 || exit 1
python3 $R/tools/pmc_sq.py $O/prof_sq_f32/sq_counter_collection.csv > $O/sq_f32.txt
rm -rf $O/prof_sq_f32
