#!/bin/bash
# Development only: build libdrm_hip.so variants with extra compile-time switches into tools/variants/
# (selected at run time with DRM_HIP_LIBRARY=...), to A/B kernels on the GPU box in one gpurun call.  Goes through the
# product Makefile, so every unit gets the SAME per-unit flags as the shipped library plus the variant's switches.
#   usage: tools/build_variants.sh NAME "EXTRA HIPCC FLAGS" [NAME "FLAGS" ...]
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CSRC=$ROOT/differentiable-robot-model_amd/csrc
OUT=$ROOT/tools/variants
mkdir -p "$OUT"
while [ $# -ge 2 ]; do
  name=$1; extra=$2; shift 2
  tmp=$(mktemp -d)
  make -s -j10 -C "$CSRC" BUILD="$tmp" LIB="$OUT/libdrm_$name.so" EXTRA="$extra -w" > /dev/null
  rm -rf "$tmp"
  echo "built $OUT/libdrm_$name.so  [$extra]"
done
