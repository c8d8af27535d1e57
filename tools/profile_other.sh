#!/bin/bash
# Only step 2b of tools/profile_gpu.sh (the reference-order and occlusion kernels' counters), keeping what is under gpurun_out/prof/.
set -e
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
OTHER="python3 $R/tools/prof_closest.py 16777216 2 ra"
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE -d $OUT/o_fetch -o p -- $OTHER > $OUT/o_fetch.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum -d $OUT/o_l2 -o p -- $OTHER > $OUT/o_l2.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE -d $OUT/o_sq -o p -- $OTHER > $OUT/o_sq.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc TA_TA_BUSY_sum TD_TD_BUSY_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE -d $OUT/o_mem -o p -- $OTHER > $OUT/o_mem.log 2>&1
echo "[profile] other ray queries' counters done"
