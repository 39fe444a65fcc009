#!/bin/bash
# only the rocprofv3 --kernel-trace --stats leg of tools/prof_r02.sh (run on the GPU box from the repo root)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
D=$O/r02
mkdir -p $D
cd /tmp && export TMPDIR=/tmp
for leg in f32: bf16:--dtype=bf16 bf16x3:--dtype=bf16x3; do
  name=${leg%%:*}; flag=${leg##*:}
  rm -rf $O/prof_$name
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$name -o $name -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu --no-host --no-latency $flag > $D/r02_bench_under_rocprof_${name}_g64.json 2> $D/prof_$name.err || exit 1
  cp $O/prof_$name/${name}_kernel_stats.csv $D/r02_kernel_stats_bench_${name}_g64.csv
  rm -rf $O/prof_$name
done
echo stats done
