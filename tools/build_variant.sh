#!/bin/bash
# Builds a tuning variant of the library:  tools/build_variant.sh NAME [-DFLAG=VALUE ...]
# -> pathtracer_amd/lib/variants/libptx_NAME.so  (select it with PTX_AMD_LIB=<path> for A/B timing; never shipped)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; shift
mkdir -p $R/pathtracer_amd/lib/variants
C=$R/pathtracer_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -std=c++17 -O3 -ffp-contract=off -fPIC -Wall -Wno-unused-function "$@" -shared \
  -o $R/pathtracer_amd/lib/variants/libptx_$NAME.so $C/ptx_api.hip $C/render.hip $C/frame.hip $C/bvh_build_gpu.hip $C/bvh_build.cpp $C/js3d.cpp $C/png_write.cpp -ldl
echo built $NAME
