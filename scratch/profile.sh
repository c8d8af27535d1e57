#!/bin/bash
# Profiling recipe for the config-3 ray-query workload (run on the GPU box through gpurun):
#   1. kernel trace + stats of the default bench command
#   2. HBM traffic counters for the culled closest-hit kernel, one --pmc pass per counter group
set -e
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $OUT/trace -o bench -- python3 $R/bench.py --steps 10 --warmup 2 > $OUT/bench_under_rocprof.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o p -- python3 $R/scratch/prof_closest.py 16777216 2 c > $OUT/pmc_fetch.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write -o p -- python3 $R/scratch/prof_closest.py 16777216 2 c > $OUT/pmc_write.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum -d $OUT/pmc_l2 -o p -- python3 $R/scratch/prof_closest.py 16777216 2 c > $OUT/pmc_l2.log 2>&1
ls -R $OUT | head -50
