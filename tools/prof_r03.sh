#!/bin/bash
# Everything profiles/ holds for round 3, in one or two GPU calls (run on the GPU box from the repo root):
#   tools/prof_r03.sh stats    rocprofv3 kernel-trace stats of the three bench lines (+ a roctx marker trace), HBM traffic per step
#                              (FETCH / WRITE PMC passes + launch trace), SQ counters
#   tools/prof_r03.sh bench    the un-profiled bench lines of every workload, per-step times, latency breakdown, worker load
# Outputs: gpurun_out/r03p/.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
D=$O/r03p
mkdir -p $D
what=${1:-all}
if [ "$what" = stats ] || [ "$what" = all ]; then
  cd /tmp && export TMPDIR=/tmp
  for leg in f32: bf16:--dtype=bf16 bf16x3:--dtype=bf16x3; do
    name=${leg%%:*}; flag=${leg##*:}
    rm -rf $O/prof_$name
    # (one lane: the per-kernel durations of the bench line's roofline are one lane's; with two passes in flight launches stretch)
    HIPVOC_LANES=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$name -o $name -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu --no-host --no-latency $flag > $D/r03_bench_under_rocprof_${name}_g64.json 2> $D/prof_$name.err || exit 1
    cp $O/prof_$name/${name}_kernel_stats.csv $D/r03_kernel_stats_bench_${name}_g64.csv
    rm -rf $O/prof_$name
  done
  echo stats done
  # roctx ranges (HIPVOC_ROCTX=1): the marker trace names every launch of a pass by plan step
  rm -rf $O/prof_roctx
  HIPVOC_ROCTX=1 timeout -k 10 300 rocprofv3 --marker-trace --kernel-trace --stats --output-format csv -d $O/prof_roctx -o roctx -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu --no-host --no-latency --utterances 32 > $D/roctx_bench.json 2> $D/prof_roctx.err || exit 1
  ls $O/prof_roctx > $D/roctx_files.txt
  for f in $O/prof_roctx/roctx_marker_api_stats.csv $O/prof_roctx/roctx_marker_api_trace.csv; do [ -f $f ] && head -n 60 $f > $D/r03_$(basename $f); done
  rm -rf $O/prof_roctx
  echo roctx done
  cd $R
  tools/prof_traffic.sh f32 profile_steps.py > $D/traffic_f32.log 2>&1 || exit 1
  tools/prof_traffic.sh bf16 bf16_steps.py > $D/traffic_bf16.log 2>&1 || exit 1
  cp $O/pmc_traffic_f32.txt $D/r03_pmc_traffic_per_step_f32_g64.txt; cp $O/pmc_traffic_f32.json $D/r03_pmc_traffic_f32_g64.json
  cp $O/pmc_traffic_bf16.txt $D/r03_pmc_traffic_per_step_bf16_g64.txt; cp $O/pmc_traffic_bf16.json $D/r03_pmc_traffic_bf16_g64.json
  echo traffic done
  tools/prof_sq_one.sh f32 f32_r03 || exit 1
  tools/prof_sq_one.sh bf16 bf16_r03 || exit 1
  cp $O/sq_f32_r03.txt $D/r03_pmc_sq_f32.txt; cp $O/sq_bf16_r03.txt $D/r03_pmc_sq_bf16.txt
  echo sq done
  # the clock the chip holds under each kernel (GRBM_GUI_ACTIVE / 8 / duration)
  cd /tmp
  for leg in f32:profile_steps.py bf16:bf16_steps.py; do
    name=${leg%%:*}; script=${leg##*:}
    rm -rf $O/prof_clk_$name
    timeout -s KILL 300 rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $O/prof_clk_$name -o clk -- python3 $R/tools/$script > $O/clk_$name.out 2>&1 || exit 1
    python3 $R/tools/pmc_clock.py $O/prof_clk_$name/clk_counter_collection.csv > $D/r03_pmc_clock_$name.txt || exit 1
    rm -rf $O/prof_clk_$name
  done
  cd $R
  echo clock done
fi
if [ "$what" = bench ] || [ "$what" = all ]; then
  cd $R
  ls profiles/r03_pmc_traffic_f32_g64.json > /dev/null 2>&1 || cp $D/r03_pmc_traffic_f32_g64.json $D/r03_pmc_traffic_bf16_g64.json profiles/ 2>/dev/null      # what bench.py reads for `traffic`
  timeout -k 10 300 python3 bench.py --steps 10 --warmup 2 > $D/r03_bench_f32_g64.json 2> $D/bench_f32.err || exit 1
  timeout -k 10 300 python3 bench.py --steps 10 --warmup 2 --dtype bf16 --no-cpu > $D/r03_bench_bf16_g64.json 2> $D/bench_bf16.err || exit 1
  timeout -k 10 300 python3 bench.py --steps 10 --warmup 2 --dtype bf16x3 --no-cpu > $D/r03_bench_bf16x3_g64.json 2> $D/bench_x3.err || exit 1
  timeout -k 10 300 python3 bench.py --steps 10 --warmup 2 --workload longform > $D/r03_bench_longform_f32.json 2> $D/bench_lf.err || exit 1
  timeout -k 10 300 python3 bench.py --steps 5 --warmup 1 --workload batch4096 > $D/r03_bench_batch4096_bf16.json 2> $D/bench_b4096.err || exit 1
  echo bench done
  timeout -k 10 200 python3 tools/profile_steps.py > $D/r03_steps_f32_g64.txt 2>/dev/null || exit 1
  timeout -k 10 200 python3 tools/bf16_steps.py > $D/r03_steps_bf16_g64.txt 2>/dev/null || exit 1
  timeout -k 10 200 python3 tools/bf16_steps.py x3 > $D/r03_steps_bf16x3_g64.txt 2>/dev/null || exit 1
  timeout -k 10 200 python3 tools/latency_breakdown.py > $D/r03_latency_single_utterance.txt 2>/dev/null || exit 1
  W=$D/r03_worker_load.txt; : > $W
  timeout -k 10 200 python3 tools/worker_load.py 8 40 4 f32 >> $W 2>&1 || exit 1
  timeout -k 10 200 python3 tools/worker_load.py 12 60 6 bf16 >> $W 2>&1 || exit 1
  timeout -k 10 200 python3 tools/worker_load.py 1 40 1 f32 >> $W 2>&1 || exit 1
  timeout -k 10 200 python3 tools/worker_load.py 1 40 1 bf16 >> $W 2>&1 || exit 1
  echo r03 bench done
fi
