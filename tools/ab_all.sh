#!/bin/bash
# A/B of the three ray queries of the bench between the shipped library and tuning variants on one box:
#   bash tools/ab_all.sh VARIANT [VARIANT ...]
cd $(dirname "$0")/..
for v in "$@"; do bash tools/ab_bench.sh $v 1 2>&1 | grep -v amdgpu.ids || exit 1; done
