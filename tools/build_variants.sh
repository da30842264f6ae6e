#!/bin/bash
# Development only: build libdrm_hip.so variants with different compile-time switches into tools/variants/
# (selected at run time with DRM_HIP_LIBRARY=...), to A/B kernels on the GPU box in one gpurun call.
#   usage: tools/build_variants.sh NAME "EXTRA HIPCC FLAGS" [NAME "FLAGS" ...]
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CSRC=$ROOT/differentiable-robot-model_amd/csrc
OUT=$ROOT/tools/variants
mkdir -p "$OUT"
BASE="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=fast -w"
while [ $# -ge 2 ]; do
  name=$1; extra=$2; shift 2
  tmp=$(mktemp -d)
  for f in drm_host drm_arm_kernels drm_arm_dynamics drm_fk_jacobian drm_fk drm_rnea drm_fk_backward drm_crba drm_rnea_backward drm_forward_dynamics; do
    /opt/rocm/bin/hipcc $BASE $extra -c -o "$tmp/$f.o" "$CSRC/$f.hip" &
  done
  wait
  /opt/rocm/bin/hipcc -fPIC --offload-arch=gfx950 -shared -o "$OUT/libdrm_$name.so" "$tmp"/*.o
  rm -rf "$tmp"
  echo "built $OUT/libdrm_$name.so  [$extra]"
done
