#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/tr; rm -rf $OUT; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --kernel-trace --stats -d $OUT -o t -- python3 $R/scratch/prof_closest.py 16777216 3 c > $OUT/log.txt 2>&1
