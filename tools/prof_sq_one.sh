#!/bin/bash
# One SQ counter pass for one mode (run on the GPU box from the repo root):  tools/prof_sq_one.sh <f32|bf16|enc> [out-name]
# -> gpurun_out/sq_<name>.txt  (per symbol: wave cycles, wait / issue-stall / active shares, MFMA busy, LDS stalls)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
name=$1; out=${2:-$1}
case $name in f32) script=profile_steps.py;; bf16) script=bf16_steps.py;; enc) script=enc_steps.py;; *) exit 64;; esac
C="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT"
rm -rf $O/prof_sq_$out
timeout -s KILL 400 rocprofv3 --pmc $C --output-format csv -d $O/prof_sq_$out -o sq -- python3 $R/tools/$script > $O/sq_$out.out 2>&1 || exit 1
python3 $R/tools/pmc_sq.py $O/prof_sq_$out/sq_counter_collection.csv > $O/sq_$out.txt
rm -rf $O/prof_sq_$out
