#!/bin/bash
# on the GPU box: per-step times of the Winograd layers for each library under qwen3-tts_amd/variants/ (x each value of $MODES for HIPVOC_WINO_RT)
for lib in qwen3-tts_amd/variants/libhipvoc_*.so; do
  tag=$(basename $lib .so); tag=${tag#libhipvoc_}
  for mode in ${MODES:-1}; do
    HIPVOC_WINO_RT=$mode HIPVOC_LIBRARY=$PWD/$lib timeout -k 10 200 python3 tools/profile_steps.py > gpurun_out/steps_var_${tag}_$mode.txt 2>/dev/null || exit 1
    echo "== $tag mode $mode: $(sed -n 1p gpurun_out/steps_var_${tag}_$mode.txt)"
    grep -E "wino_" gpurun_out/steps_var_${tag}_$mode.txt | awk '{printf "   %s %s", $1, $2} END {print ""}'
  done
done
