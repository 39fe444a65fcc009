#!/bin/bash
# bf16x3 mode: rocprofv3 kernel-trace summary of the bench line, the un-profiled line, accuracy check and per-step times
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
D=$O/x3
rm -rf $D $O/prof_x3 && mkdir -p $D
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_x3 -o x3 -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu --dtype bf16x3 > $D/r01_bench_under_rocprof_bf16x3_g64.json 2> $D/prof.err
cp $O/prof_x3/x3_kernel_stats.csv $D/r01_kernel_stats_bench_bf16x3_g64.csv
rm -rf $O/prof_x3
cd $R
timeout -k 10 300 python3 bench.py --dtype bf16x3 > $D/r01_bench_bf16x3_g64.json 2> $D/bench.err
timeout -k 10 300 python3 tools/x3_check.py oracle > $D/r01_bf16x3_check_and_steps.txt 2>&1
timeout -k 10 200 python3 tools/bf16_steps.py x3 >> $D/r01_bf16x3_check_and_steps.txt 2>&1
echo x3 done
