#!/bin/bash
# Triangle-phase thresholds of the culled mesh kernel (config 3 timing, one process per setting).
R=$GRAFT_REPO_ROOT; L=$R/gpurun_out/park.log; : > $L
for t in 24 32 40 48; do for s in 8 16 24; do
  echo "park_min $t stuck_max $s" >> $L
  PTX_TRI_MIN_LANES=$t PTX_STUCK_MAX_LANES=$s timeout -k 10 100 python $R/scratch/sweep.py 2>&1 | grep "refill" >> $L || exit 1
done; done
cat $L
