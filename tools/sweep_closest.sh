#!/bin/bash
# Sweep of the headline kernel's scheduling thresholds (environment knobs read by launch_persistent) on one box.
cd $(dirname "$0")/..
run() { echo "$*"; env "$@" timeout -k 10 120 python3 tools/time_closest.py 16777216 6 2>&1 | grep Mrays || exit 1; }
run PTX_TRI_MIN_LANES=32
run PTX_TRI_MIN_LANES=24
run PTX_TRI_MIN_LANES=28
run PTX_TRI_MIN_LANES=36
run PTX_TRI_MIN_LANES=40
run PTX_STUCK_MAX_LANES=12
run PTX_STUCK_MAX_LANES=20
run PTX_STUCK_MAX_LANES=24
run PTX_REFILL_MIN_IDLE=12
run PTX_REFILL_MIN_IDLE=16
run PTX_REFILL_MIN_IDLE=28
run PTX_BLOCKS_PER_CU=6
run PTX_BLOCKS_PER_CU=7
