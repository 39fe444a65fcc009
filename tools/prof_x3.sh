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
# HBM traffic of the mode's kernels: FETCH_SIZE and WRITE_SIZE in separate passes (MI355X_MICROARCH.md)
cd /tmp
timeout -s KILL 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/prof_pmc_x3_fetch -o fetch -- python3 $R/tools/bf16_steps.py x3 > $D/pmc_fetch.out 2>&1
timeout -s KILL 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/prof_pmc_x3_write -o write -- python3 $R/tools/bf16_steps.py x3 > $D/pmc_write.out 2>&1
cd $R
python3 tools/pmc_traffic.py $O/prof_pmc_x3_fetch/fetch_counter_collection.csv $O/prof_pmc_x3_write/write_counter_collection.csv $D/r01_pmc_traffic_bf16x3_g64.json 64 "As r01_pmc_traffic_g64.json, for tools/bf16_steps.py x3 (bf16x3 split-operand mode, fp32 exchange, 6 passes of 64 units)." > $D/r01_pmc_traffic_bf16x3_g64.txt
rm -rf $O/prof_pmc_x3_fetch $O/prof_pmc_x3_write
cp $D/r01_pmc_traffic_bf16x3_g64.json $R/profiles/
timeout -k 10 300 python3 bench.py --dtype bf16x3 > $D/r01_bench_bf16x3_g64.json 2> $D/bench.err
echo x3 done
