#!/bin/bash
# tools/build_variants.sh "<tag>:<-D flags>" ... : libhipvoc_<tag>.so with kernels_wino.hip compiled under those flags (experiments;
# run with HIPVOC_LIBRARY=qwen3-tts_amd/variants/libhipvoc_<tag>.so)
cd "$(dirname "$0")/../qwen3-tts_amd/csrc" || exit 1
mkdir -p ../variants
for v in "$@"; do
  tag=${v%%:*}; flags=${v#*:}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -fno-slp-vectorize $flags -c kernels_wino.hip -o /tmp/kernels_wino_$tag.o || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ../variants/libhipvoc_$tag.so kernels.o kernels_bf16.o kernels_enc.o kernels_gemm1.o /tmp/kernels_wino_$tag.o scratch.o hipvoc.o safetensors.o wire.o discovery.o || exit 1
  echo built $tag
done
