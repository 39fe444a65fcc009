#!/bin/bash
# Everything profiles/ holds for round 2, in one GPU call (run on the GPU box from the repo root): rocprofv3 kernel-trace
# stats of the bench lines, per-step HBM traffic (FETCH / WRITE PMC passes + launch trace), SQ counters, the un-profiled
# bench lines of every workload, per-step times, the single-utterance latency breakdown.  Outputs: gpurun_out/r02/.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
D=$O/r02
rm -rf $D && mkdir -p $D
cd /tmp && export TMPDIR=/tmp
for leg in f32: bf16:--dtype=bf16 bf16x3:--dtype=bf16x3; do
  name=${leg%%:*}; flag=${leg##*:}
  rm -rf $O/prof_$name
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$name -o $name -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu --no-host --no-latency $flag > $D/r02_bench_under_rocprof_${name}_g64.json 2> $D/prof_$name.err || exit 1
  cp $O/prof_$name/${name}_kernel_stats.csv $D/r02_kernel_stats_bench_${name}_g64.csv
  rm -rf $O/prof_$name
done
echo stats done
cd $R
tools/prof_traffic.sh f32 profile_steps.py > $D/traffic_f32.log 2>&1 || exit 1
tools/prof_traffic.sh bf16 bf16_steps.py > $D/traffic_bf16.log 2>&1 || exit 1
cp $O/pmc_traffic_f32.txt $D/r02_pmc_traffic_per_step_f32_g64.txt; cp $O/pmc_traffic_f32.json $D/r02_pmc_traffic_f32_g64.json
cp $O/pmc_traffic_bf16.txt $D/r02_pmc_traffic_per_step_bf16_g64.txt; cp $O/pmc_traffic_bf16.json $D/r02_pmc_traffic_bf16_g64.json
echo traffic done
tools/prof_sq_f32.sh || exit 1
cp $O/sq_f32.txt $D/r02_pmc_sq_f32.txt
echo sq done
cp $D/r02_pmc_traffic_f32_g64.json $D/r02_pmc_traffic_bf16_g64.json $R/profiles/      # what bench.py reads for `traffic`
timeout -k 10 300 python3 bench.py --steps 10 --warmup 2 > $D/r02_bench_f32_g64.json 2> $D/bench_f32.err || exit 1
timeout -k 10 300 python3 bench.py --steps 10 --warmup 2 --dtype bf16 --no-cpu > $D/r02_bench_bf16_g64.json 2> $D/bench_bf16.err || exit 1
timeout -k 10 300 python3 bench.py --steps 10 --warmup 2 --dtype bf16x3 --no-cpu > $D/r02_bench_bf16x3_g64.json 2> $D/bench_x3.err || exit 1
timeout -k 10 300 python3 bench.py --steps 10 --warmup 2 --workload longform > $D/r02_bench_longform_f32.json 2> $D/bench_lf.err || exit 1
timeout -k 10 300 python3 bench.py --steps 5 --warmup 1 --workload batch4096 > $D/r02_bench_batch4096_bf16.json 2> $D/bench_b4096.err || exit 1
echo bench done
timeout -k 10 200 python3 tools/profile_steps.py > $D/r02_steps_f32_g64.txt 2>/dev/null || exit 1
timeout -k 10 200 python3 tools/bf16_steps.py > $D/r02_steps_bf16_g64.txt 2>/dev/null || exit 1
timeout -k 10 200 python3 tools/bf16_steps.py x3 > $D/r02_steps_bf16x3_g64.txt 2>/dev/null || exit 1
HIPVOC_WINO=0 timeout -k 10 200 python3 tools/profile_steps.py > $D/r02_steps_f32_g64_direct_kernels.txt 2>/dev/null || exit 1
timeout -k 10 200 python3 tools/latency_breakdown.py > $D/r02_latency_single_utterance.txt 2>/dev/null || exit 1
tools/prof_worker_r02.sh || exit 1
echo r02 profiles done
