#!/bin/bash
# tools/build_variant.sh <tag> <source file in csrc> "<-D flags>": libhipvoc_<tag>.so with that one file compiled under the
# flags (timing experiments; run with HIPVOC_LIBRARY=$PWD/qwen3-tts_amd/variants/libhipvoc_<tag>.so).  `make` first.
cd "$(dirname "$0")/../qwen3-tts_amd/csrc" || exit 1
mkdir -p ../variants
tag=$1; src=$2; flags=$3
base=${src%.*}
extra=""; [ "$base" = kernels_wino ] && extra="-fno-slp-vectorize"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $extra $flags -c $src -o /tmp/${base}_$tag.o || exit 1
objs=""
for o in kernels.o kernels_bf16.o kernels_blk3.o kernels_enc.o kernels_gemm1.o kernels_wino.o scratch.o hipvoc.o safetensors.o wire.o discovery.o; do
  if [ "$o" = "$base.o" ]; then objs="$objs /tmp/${base}_$tag.o"; else objs="$objs $o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ../variants/libhipvoc_$tag.so $objs || exit 1
echo built $tag
