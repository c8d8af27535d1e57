#!/bin/bash
# Sweep of the leaf-parking thresholds on the config-3 workload.
set -e
PTX_AMD_LIB=$PWD/scratch/bin/libptx_prof.so timeout -k 10 120 python scratch/prof_closest.py 16777216 1 c 2>&1 | grep lanes
for t in 16 24 32 40 48; do for s in 8 16 32; do
  echo "park_min=$t stuck_max=$s"
  PTX_TRI_MIN_LANES=$t PTX_STUCK_MAX_LANES=$s timeout -k 10 120 python scratch/sweep.py 2>/dev/null | grep -v without
done; done
