#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/pmc3; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES -d $OUT/sq1 -o p -- python3 $R/scratch/prof_render.py 16 > $OUT/sq1.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc TA_TA_BUSY_sum TD_TD_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE -d $OUT/mem1 -o p -- python3 $R/scratch/prof_render.py 16 > $OUT/mem1.log 2>&1
timeout -s KILL 120 rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- python3 $R/scratch/prof_render.py 64 > $OUT/trace.log 2>&1
cat $OUT/trace.log | tail -2
