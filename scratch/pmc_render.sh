#!/bin/bash
# Counter passes over one short render; per-kernel sums printed from the rocpd databases.
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/pmc3; rm -rf $OUT; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES -d $OUT/sq1 -o p -- python3 $R/scratch/prof_render.py 16 > $OUT/sq1.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_INSTS_FLAT -d $OUT/sq2 -o p -- python3 $R/scratch/prof_render.py 16 > $OUT/sq2.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc TA_TA_BUSY_sum TD_TD_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE -d $OUT/mem1 -o p -- python3 $R/scratch/prof_render.py 16 > $OUT/mem1.log 2>&1 || exit 1
python3 - <<PY
import sqlite3, glob, collections
for d in ("sq1", "sq2", "mem1"):
    db = glob.glob("$OUT/%s/**/*.db" % d, recursive=True)[0]
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
    rows = c.execute("select kernel_name, counter_name, sum(value), count(*) from counters_collection group by kernel_name, counter_name").fetchall()
    agg = collections.defaultdict(dict)
    for k, n, v, cnt in rows:
        agg[k[:60]][n] = v
    for k, m in agg.items():
        if "rocclr" in k: continue
        print(k)
        for n, v in sorted(m.items()): print(f"    {n:32s} {v:.4g}")
PY
