#!/bin/bash
# tools/build_variant.sh <tag> <source file(s) in csrc, comma-separated> "<-D flags>": libhipvoc_<tag>.so with those files compiled
# under the flags (timing experiments; run with HIPVOC_LIBRARY=$PWD/qwen3-tts_amd/variants/libhipvoc_<tag>.so).  `make` first.
cd "$(dirname "$0")/../qwen3-tts_amd/csrc" || exit 1
mkdir -p ../variants
tag=$1; srcs=$2; flags=$3
objs=" kernels.o kernels_bf16.o kernels_bf16_ring.o kernels_blk3.o kernels_enc.o kernels_gemm1.o kernels_wino.o scratch.o hipvoc.o safetensors.o wire.o discovery.o "
for src in ${srcs//,/ }; do
  base=${src%.*}
  extra=""; [ "$base" = kernels_wino ] && extra="-fno-slp-vectorize"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $extra $flags -c $src -o /tmp/${base}_$tag.o || exit 1
  objs=${objs/ $base.o / /tmp/${base}_$tag.o }
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ../variants/libhipvoc_$tag.so $objs || exit 1
echo built $tag
