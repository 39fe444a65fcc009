#!/bin/bash
# timing ablations of the bf16 ring kernels (libraries under qwen3-tts_amd/variants/): per-step lines of the kernels named in $PAT
mkdir -p gpurun_out/ring
PAT=${PAT:-ring_kernel}
for lib in qwen3-tts_amd/variants/libhipvoc_*.so; do
  tag=$(basename $lib .so); tag=${tag#libhipvoc_}
  HIPVOC_LIBRARY=$PWD/$lib timeout -k 10 200 python3 tools/bf16_steps.py > gpurun_out/ring/steps_var_$tag.txt 2>/dev/null || exit 1
  echo "== $tag: $(sed -n 1p gpurun_out/ring/steps_var_$tag.txt)"
  grep -E "$PAT" gpurun_out/ring/steps_var_$tag.txt
done
