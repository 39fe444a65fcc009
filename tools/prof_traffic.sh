#!/bin/bash
# HBM traffic per kernel and per step (run on the GPU box from the repo root):  tools/prof_traffic.sh <tag> <steps script> [env...]
# FETCH_SIZE and WRITE_SIZE in separate passes (MI355X_MICROARCH.md), each its own run; the launch trace of the same program
# (HIPVOC_TRACE_STEPS=1) attributes the dispatches to steps.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
tag=$1; script=$2
rm -rf $O/prof_pmc_${tag}_fetch $O/prof_pmc_${tag}_write
export HIPVOC_TRACE_STEPS=1
timeout -s KILL 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/prof_pmc_${tag}_fetch -o fetch -- python3 $R/tools/$script > $O/pmc_${tag}_fetch.out 2> $O/pmc_${tag}_steps.txt || exit 1
timeout -s KILL 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/prof_pmc_${tag}_write -o write -- python3 $R/tools/$script > $O/pmc_${tag}_write.out 2> $O/pmc_${tag}_steps_w.txt || exit 1
python3 $R/tools/pmc_traffic.py $O/prof_pmc_${tag}_fetch/fetch_counter_collection.csv $O/prof_pmc_${tag}_write/write_counter_collection.csv \
    $O/pmc_traffic_${tag}.json ${UNITS:-64} "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE of tools/$script (separate runs), per kernel symbol and per step" $O/pmc_${tag}_steps.txt > $O/pmc_traffic_${tag}.txt
rm -rf $O/prof_pmc_${tag}_fetch $O/prof_pmc_${tag}_write
tail -n 45 $O/pmc_traffic_${tag}.txt
