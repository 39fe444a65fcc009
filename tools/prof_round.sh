#!/bin/bash
# round-end rocprofv3 kernel-trace summaries of the three bench lines (run on the GPU box from the repo root)
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
rm -rf $O/prof_*; 
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_f32 -o f32 -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu > $O/bench_under_rocprof_f32.json 2> $O/prof_f32.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bf16 -o bf16 -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu --dtype bf16 > $O/bench_under_rocprof_bf16.json 2> $O/prof_bf16.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_enc -o enc -- python3 $R/bench_encoder.py --steps 1 --warmup 1 --no-cpu > $O/bench_under_rocprof_enc.json 2> $O/prof_enc.err
echo stats done
# HBM traffic: FETCH_SIZE and WRITE_SIZE in separate passes (MI355X_MICROARCH.md), each its own run
for leg in f32:profile_steps.py bf16:bf16_steps.py enc:enc_steps.py; do
  name=${leg%%:*}; script=${leg##*:}
  timeout -s KILL 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/prof_pmc_${name}_fetch -o fetch -- python3 $R/tools/$script > $O/pmc_${name}_fetch.out 2>&1
  timeout -s KILL 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/prof_pmc_${name}_write -o write -- python3 $R/tools/$script > $O/pmc_${name}_write.out 2>&1
  echo pmc $name done
done
