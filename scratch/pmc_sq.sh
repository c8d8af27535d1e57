#!/bin/bash
# Instruction-issue counters for the culled closest-hit kernel (one --pmc pass per group).
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/pmc2; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 -L > $OUT/avail.txt 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_BUSY_CYCLES -d $OUT/sq1 -o p -- python3 $R/scratch/prof_closest.py 16777216 1 c > $OUT/sq1.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE -d $OUT/sq2 -o p -- python3 $R/scratch/prof_closest.py 16777216 1 c > $OUT/sq2.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc TA_TA_BUSY_sum TD_TD_BUSY_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum GRBM_GUI_ACTIVE -d $OUT/mem1 -o p -- python3 $R/scratch/prof_closest.py 16777216 1 c > $OUT/mem1.log 2>&1
ls $OUT/*
