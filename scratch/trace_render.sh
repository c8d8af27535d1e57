#!/bin/bash
# Kernel-level breakdown of one render (refract_cow_cbox 1024x1024 256 spp).
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/trender; rm -rf $OUT; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- python3 $R/scratch/prof_render.py 256 > $OUT/trace.log 2>&1 || exit 1
python3 - <<PY
import sqlite3, glob
db = glob.glob("$OUT/trace/**/*.db", recursive=True)[0]
c = sqlite3.connect(db)
rows = c.execute("select name, count(*), sum(end-start)/1e6, avg(end-start)/1e3 from kernels group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
with open("$OUT/summary.txt", "w") as f:
    for r in rows:
        f.write(f"{r[2]:9.2f} ms {100*r[2]/tot:5.1f}% n={r[1]:6d} avg={r[3]:9.1f} us  {r[0][:90]}\n")
print(open("$OUT/summary.txt").read())
PY
