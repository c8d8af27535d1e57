#!/bin/bash
# A/B of the render legs between the shipped library and tuning variants, alternating on one box:
#   [CONFIGS="2 5 5f d"] bash tools/ab_render.sh VARIANT [VARIANT ...]
cd $(dirname "$0")/..
V=$PWD/pathtracer_amd/lib/variants
for rep in 1 2; do
  echo "== shipped"; timeout -k 10 200 python3 tools/time_render.py ${CONFIGS:-2 5 d} 2>&1 | grep -v amdgpu.ids || exit 1
  for v in "$@"; do echo "== $v"; PTX_AMD_LIB=$V/libptx_$v.so timeout -k 10 200 python3 tools/time_render.py ${CONFIGS:-2 5 d} 2>&1 | grep -v amdgpu.ids || exit 1; done
done
