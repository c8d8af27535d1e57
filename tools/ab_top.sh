#!/bin/bash
# A/B of the config-3 closest-hit kernel between the shipped library and tuning variants, alternating on one box:
#   bash tools/ab_top.sh VARIANT [VARIANT ...]     (variants built by tools/build_variant.sh)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
for rep in 1 2; do
  timeout -k 10 120 python3 $R/tools/time_closest.py 16777216 10
  for v in "$@"; do
    PTX_AMD_LIB=$R/pathtracer_amd/lib/variants/libptx_$v.so timeout -k 10 120 python3 $R/tools/time_closest.py 16777216 10
  done
done
