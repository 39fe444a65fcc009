#!/bin/bash
# rocprofv3 kernel-trace summaries of the three bench lines, without the PMC passes (run on the GPU box from the repo root)
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
rm -rf $O/prof_f32 $O/prof_bf16 $O/prof_enc
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_f32 -o f32 -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu > $O/bench_under_rocprof_f32.json 2> $O/prof_f32.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bf16 -o bf16 -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu --dtype bf16 > $O/bench_under_rocprof_bf16.json 2> $O/prof_bf16.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_enc -o enc -- python3 $R/bench_encoder.py --steps 1 --warmup 1 --no-cpu > $O/bench_under_rocprof_enc.json 2> $O/prof_enc.err
rm -f $O/prof_*/*_kernel_trace.csv
echo stats done
