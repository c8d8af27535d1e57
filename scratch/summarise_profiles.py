"""Turns the rocprofv3 databases left under gpurun_out/ by scratch/profile.sh and scratch/pmc_sq.sh into the
summaries kept under profiles/ (kernel stats CSV of the bench command, PMC sums of the culled closest-hit kernel,
HBM-side traffic per launch).  Run here after the gpurun call has merged gpurun_out/ back."""
import sqlite3, glob, json, os, sys, csv
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "gpurun_out")
P = os.path.join(ROOT, "profiles")
RND = "r01"
KERNEL = "k_mesh_persistent<1, false>"

def db(path):
    f = glob.glob(os.path.join(G, path, "**", "*.db"), recursive=True)
    return sqlite3.connect(f[0]) if f else None

# ---- kernel stats of the bench command
c = db("prof/trace")
rows = c.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
with open(os.path.join(P, f"{RND}_kernel_stats_bench.csv"), "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for r in rows:
        w.writerow([r[0], r[1], r[2], round(r[3], 1), round(100.0 * r[2] / tot, 3), r[4], r[5]])
line = [l for l in open(os.path.join(G, "prof", "bench_under_rocprof.log")) if l.startswith("{")][-1]
json.dump(json.loads(line), open(os.path.join(P, f"{RND}_bench_line.json"), "w"), indent=1)

# ---- PMC sums per launch of the culled closest-hit kernel
def sums(path):
    c = db(path)
    if c is None: return {}
    out = {}
    for name, v, n in c.execute("select counter_name, sum(value), count(distinct dispatch_id) from counters_collection where kernel_name like ? group by counter_name", ("%" + KERNEL + "%",)):
        out[name] = v / max(n, 1)
    return out
cnt = {}
for d in ("prof/pmc_fetch", "prof/pmc_write", "prof/pmc_l2", "pmc2/sq1", "pmc2/sq2", "pmc2/mem1"):
    cnt.update(sums(d))
if "FETCH_SIZE" in cnt:
    hit, miss = cnt.get("TCC_HIT_sum", 0.0), cnt.get("TCC_MISS_sum", 0.0)
    traffic = {
        "round": 1,
        "kernel": "ptx::k_mesh_persistent<1,false> (culled closest-hit), 16777216 rays vs 1M-triangle soup",
        "command": "rocprofv3 --pmc <counters> -- python3 scratch/prof_closest.py 16777216 2 c   (one pass per counter group: scratch/profile.sh, scratch/pmc_sq.sh)",
        "FETCH_SIZE_KB_per_launch": cnt["FETCH_SIZE"], "WRITE_SIZE_KB_per_launch": cnt.get("WRITE_SIZE"),
        "TCC_HIT_sum": hit, "TCC_MISS_sum": miss, "l2_hit_rate": hit / (hit + miss) if hit + miss else None,
        "correction": "KB = 1024 B.  FETCH_SIZE = 64 B per L2 read miss.  The guide's gfx950 rule (double FETCH_SIZE) holds for wide coalesced streaming reads, "
                      "whose 128-B requests are tallied at 64 B; this kernel reads one 64-byte node or 48-byte triangle record per lane, and on that pattern "
                      "the counter was calibrated against a known byte count (profiles/r01_fetch_calibration.json: 13.42 GB of random 64-B records requested, "
                      "FETCH_SIZE 13.78 GB): taken as is.  WRITE_SIZE taken as is.  The counters sit on the L2's fabric side, so Infinity Cache hits are included: "
                      "with a 69 MB working set this is an upper bound on DRAM traffic, not DRAM traffic.",
        "closest_culled_hbm_bytes_per_launch": (cnt["FETCH_SIZE"] + cnt.get("WRITE_SIZE", 0.0)) * 1024.0,
        "bytes_per_launch_if_fetch_size_were_doubled": (2.0 * cnt["FETCH_SIZE"] + cnt.get("WRITE_SIZE", 0.0)) * 1024.0,
    }
    json.dump(traffic, open(os.path.join(P, "traffic.json"), "w"), indent=1)
if "SQ_INSTS_VALU" in cnt and "GRBM_GUI_ACTIVE" in cnt:
    cyc = cnt["GRBM_GUI_ACTIVE"] / 8.0
    n = 16777216.0
    d = {"cycles_per_xcd": cyc,
         "valu_busy_frac": 4.0 * cnt["SQ_ACTIVE_INST_VALU"] / (cyc * 1024.0),
         "valu_lane_utilisation": cnt["SQ_THREAD_CYCLES_VALU"] / (cnt["SQ_INSTS_VALU"] * 64.0) if "SQ_THREAD_CYCLES_VALU" in cnt else None,
         "td_busy_frac": cnt["TD_TD_BUSY_sum"] / (cyc * 256.0), "ta_busy_frac": cnt["TA_TA_BUSY_sum"] / (cyc * 256.0),
         "l2_hit_rate": cnt.get("TCC_HIT_sum", 0.0) / max(1.0, cnt.get("TCC_HIT_sum", 0.0) + cnt.get("TCC_MISS_sum", 0.0)),
         "valu_instructions_per_ray": cnt["SQ_INSTS_VALU"] / n, "vmem_read_instructions_per_ray": cnt["SQ_INSTS_VMEM_RD"] / n}
    json.dump({"round": 1, "kernel": "ptx::k_mesh_persistent<1,false> (culled closest-hit), 16777216 rays vs 1M-triangle soup",
               "note": "sums over the chip per launch; SQ_ACTIVE_INST_* count quad-cycles (x4 = cycles); busy fractions are relative to GRBM_GUI_ACTIVE/8 (cycles per XCD) measured in the same pass",
               "counters": cnt, "derived": d}, open(os.path.join(P, f"{RND}_pmc_closest.json"), "w"), indent=1)
    print(json.dumps(d, indent=1))
print({k: cnt[k] for k in ("FETCH_SIZE", "WRITE_SIZE", "TCC_HIT_sum", "TCC_MISS_sum") if k in cnt})
