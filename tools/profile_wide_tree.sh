set -e
R=$GRAFT_REPO_ROOT; V=$R/pathtracer_amd/lib/variants; OUT=$R/gpurun_out/wide; rm -rf $OUT; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
PTX_WIDE_TREE=0 PTX_AMD_LIB=$V/libptx_prof.so timeout -k 10 120 python3 $R/tools/prof_closest.py 16777216 1 c > $OUT/lanes_binary.txt 2>&1
PTX_WIDE_TREE=1 PTX_BLOCKS_PER_CU=5 PTX_AMD_LIB=$V/libptx_prof.so timeout -k 10 120 python3 $R/tools/prof_closest.py 16777216 1 c > $OUT/lanes_wide.txt 2>&1
export PTX_WIDE_TREE=1 PTX_BLOCKS_PER_CU=5 PTX_AMD_LIB=$V/libptx_wide5.so
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE -d $OUT/sq -o p -- python3 $R/tools/prof_closest.py 16777216 2 c > $OUT/sq.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc TA_TA_BUSY_sum TD_TD_BUSY_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE -d $OUT/mem -o p -- python3 $R/tools/prof_closest.py 16777216 2 c > $OUT/mem.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum -d $OUT/l2 -o p -- python3 $R/tools/prof_closest.py 16777216 2 c > $OUT/l2.log 2>&1
timeout -s KILL 200 rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- python3 $R/tools/prof_closest.py 16777216 4 c > $OUT/trace.log 2>&1
grep lanes $OUT/lanes_*.txt
