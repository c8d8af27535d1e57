"""Runs on the GPU box at the end of tools/profile_gpu.sh: every rocprofv3 database under gpurun_out/prof/ is reduced
to a small JSON file beside it (per kernel: counter sums and dispatch counts; for kernel traces: per-kernel duration
statistics and the start / end of every traversal-kernel launch) and then deleted, so that what travels back stays
far below the size limit on gpurun_out/.   python3 tools/reduce_profiles.py [dir]"""
import glob
import json
import os
import shutil
import sqlite3
import sys

G = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "prof")


def tables(c):
    return {r[0] for r in c.execute("select name from sqlite_master where type in ('table', 'view')")}


for d in sorted(glob.glob(os.path.join(G, "*"))):
    if not os.path.isdir(d):
        continue
    dbs = glob.glob(os.path.join(d, "**", "*.db"), recursive=True)
    if not dbs:
        continue
    c = sqlite3.connect(dbs[0])
    t = tables(c)
    out = {}
    if "counters_collection" in t:
        agg = {}
        for k, n, v, cnt in c.execute("select kernel_name, counter_name, sum(value), count(distinct dispatch_id) from counters_collection group by kernel_name, counter_name"):
            agg.setdefault(k, {})[n] = v
            agg[k]["dispatches"] = cnt
        if agg:
            out["counters"] = agg
    if "kernels" in t:
        rows = c.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by name order by 3 desc").fetchall()
        if rows:
            out["kernel_stats"] = [list(r) for r in rows]
            out["traversal_launches"] = {}
            for (name,) in c.execute("select distinct name from kernels where name like '%k_mesh_persistent%'").fetchall():
                out["traversal_launches"][name] = c.execute("select start, end from kernels where name = ? order by start", (name,)).fetchall()
    c.close()
    json.dump(out, open(d + ".json", "w"))
    shutil.rmtree(d)
    print(f"[reduce] {os.path.basename(d)}: {os.path.getsize(d + '.json') // 1024} KiB")
