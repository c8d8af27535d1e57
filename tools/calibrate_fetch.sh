#!/bin/bash
# FETCH_SIZE against a known byte count in the traversal's access pattern: one random 64-byte record per
# lane (four 16-byte loads) from a 1 GB table, far larger than the L2s.
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/calib; rm -rf $OUT; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE -d $OUT/fetch -o p -- $R/tools/bin/gather_bench 16000000 > $OUT/run.log 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum -d $OUT/l2 -o p -- $R/tools/bin/gather_bench 16000000 > $OUT/run2.log 2>&1 || exit 1
python3 - <<PY
import sqlite3, glob
for d in ("fetch", "l2"):
    c = sqlite3.connect(glob.glob("$OUT/%s/**/*.db" % d, recursive=True)[0])
    for k, n, v, cnt in c.execute("select kernel_name, counter_name, sum(value), count(distinct dispatch_id) from counters_collection group by kernel_name, counter_name"):
        print(k[:40], n, v / cnt)
print("bytes requested per launch:", 2048 * 256 * 400 * 64)
PY
grep mode $OUT/run.log
