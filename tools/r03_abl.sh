#!/bin/bash
# timing ablations of the fused block kernel (libraries under qwen3-tts_amd/variants/), and the v_sin_f32 accuracy check
mkdir -p gpurun_out/r03
timeout -k 10 60 tools/ubench/vsin_err > gpurun_out/r03/vsin_err.txt 2>&1; cat gpurun_out/r03/vsin_err.txt
for lib in qwen3-tts_amd/variants/libhipvoc_*.so; do
  tag=$(basename $lib .so); tag=${tag#libhipvoc_}
  HIPVOC_LIBRARY=$PWD/$lib timeout -k 10 200 python3 tools/bf16_steps.py > gpurun_out/r03/steps_bf16_var_$tag.txt 2>/dev/null || exit 1
  echo "== $tag: $(sed -n 1p gpurun_out/r03/steps_bf16_var_$tag.txt) | $(grep resblock gpurun_out/r03/steps_bf16_var_$tag.txt)"
done
