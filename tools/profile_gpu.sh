#!/bin/bash
# Profiling recipe, run on the GPU box through gpurun:  bash tools/profile_gpu.sh
# Leaves rocprofv3 databases under gpurun_out/prof/; tools/summarise_profiles.py turns them into the summaries
# kept under profiles/.  Counter passes are runs of their own (kernel trace and counters are never combined),
# one pass per counter group, each under its own time limit; the program stands directly after `--`.
set -e
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CLOSEST="python3 $R/tools/prof_closest.py 16777216 2 c"
RENDER="python3 $R/tools/prof_render.py 16"
# 1. kernel trace + stats of the default bench command
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $OUT/trace -o bench -- python3 $R/bench.py --steps 10 --warmup 2 --no-4k > $OUT/bench_under_rocprof.log 2>&1
echo "[profile] bench trace done"
# 2. culled closest-hit kernel: fabric-side traffic, L2 hit rate, issue counters, vector-memory path
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE -d $OUT/c_fetch -o p -- $CLOSEST > $OUT/c_fetch.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE -d $OUT/c_write -o p -- $CLOSEST > $OUT/c_write.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum -d $OUT/c_l2 -o p -- $CLOSEST > $OUT/c_l2.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_BUSY_CYCLES -d $OUT/c_sq1 -o p -- $CLOSEST > $OUT/c_sq1.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE -d $OUT/c_sq2 -o p -- $CLOSEST > $OUT/c_sq2.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc TA_TA_BUSY_sum TD_TD_BUSY_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum GRBM_GUI_ACTIVE -d $OUT/c_mem -o p -- $CLOSEST > $OUT/c_mem.log 2>&1
echo "[profile] closest-hit counters done"
# 2b. the other two ray queries of the bench (closest hit in the reference order, occlusion in the culled order)
OTHER="python3 $R/tools/prof_closest.py 16777216 2 ra"
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE -d $OUT/o_fetch -o p -- $OTHER > $OUT/o_fetch.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum -d $OUT/o_l2 -o p -- $OTHER > $OUT/o_l2.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE -d $OUT/o_sq -o p -- $OTHER > $OUT/o_sq.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc TA_TA_BUSY_sum TD_TD_BUSY_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE -d $OUT/o_mem -o p -- $OTHER > $OUT/o_mem.log 2>&1
echo "[profile] other ray queries' counters done"
# 3. render kernels (refract_cow_cbox 1024x1024, 16 spp, depth 8): trace, then the same counter groups
timeout -s KILL 200 rocprofv3 --kernel-trace --stats -d $OUT/r_trace -o t -- python3 $R/tools/prof_render.py 256 > $OUT/r_trace.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES -d $OUT/r_sq1 -o p -- $RENDER > $OUT/r_sq1.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_INSTS_FLAT GRBM_GUI_ACTIVE -d $OUT/r_sq2 -o p -- $RENDER > $OUT/r_sq2.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc TA_TA_BUSY_sum TD_TD_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE -d $OUT/r_mem -o p -- $RENDER > $OUT/r_mem.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE -d $OUT/r_fetch -o p -- $RENDER > $OUT/r_fetch.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum -d $OUT/r_l2 -o p -- $RENDER > $OUT/r_l2.log 2>&1
echo "[profile] render counters done"
du -sh $OUT
