"""Turns the rocprofv3 databases that tools/profile_gpu.sh left under gpurun_out/prof/ into the summaries kept under
profiles/ for one round:  python tools/summarise_profiles.py r02
  rNN_kernel_stats_bench.csv   per-kernel time of the default bench command (kernel trace)
  rNN_bench_line.json          the line that command printed
  rNN_traffic.json             L2<->fabric bytes per launch of the culled closest-hit kernel, L2 hit rate
  rNN_pmc_closest.json         per-launch counter sums of that kernel + derived busy fractions
  rNN_pmc_other_queries.json   the same for closest hit in the reference order (the API's default) and for occlusion queries
  rNN_kernel_stats_render.csv  per-kernel time of a config-2 render (refract_cow_cbox 1024x1024, 256 spp)
  rNN_pmc_scene.json           counter sums of the render's kernels (two-level extend, shade, raygen) over a 16 spp render
Run here after the gpurun call has merged gpurun_out/ back."""
import collections
import csv
import glob
import json
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "gpurun_out", "prof")
P = os.path.join(ROOT, "profiles")
RND = sys.argv[1] if len(sys.argv) > 1 else "r02"
KERNEL = "k_mesh_persistent<1, false, false, false>"      # <culled, closest, binary tree, first pass>
N_RAYS = 16777216.0
CUS = 256


def db(path):
    f = glob.glob(os.path.join(G, path, "**", "*.db"), recursive=True)
    return sqlite3.connect(f[0]) if f else None


def kernel_stats(path, out_csv):
    c = db(path)
    if c is None:
        return None
    rows = c.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows)
    with open(os.path.join(P, out_csv), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for r in rows:
            w.writerow([r[0], r[1], r[2], round(r[3], 1), round(100.0 * r[2] / tot, 3), r[4], r[5]])
    return rows


def sums(path, like):
    """Per-launch mean of every counter over the dispatches of kernels whose name matches."""
    c = db(path)
    if c is None:
        return {}
    out = {}
    for name, v, n in c.execute("select counter_name, sum(value), count(distinct dispatch_id) from counters_collection where kernel_name like ? group by counter_name", ("%" + like + "%",)):
        out[name] = v / max(n, 1)
    return out


def totals(path):
    """Per kernel: sum of every counter over all dispatches, and the dispatch count."""
    c = db(path)
    agg = collections.defaultdict(dict)
    if c is None:
        return agg
    for k, n, v, cnt in c.execute("select kernel_name, counter_name, sum(value), count(distinct dispatch_id) from counters_collection group by kernel_name, counter_name"):
        agg[k][n] = v
        agg[k]["dispatches"] = cnt
    return agg


def derived(cnt, per_unit=None):
    cyc = cnt["GRBM_GUI_ACTIVE"] / 8.0          # the counter sums the eight XCDs
    d = {"cycles_per_xcd": cyc,
         "valu_busy_frac": 4.0 * cnt["SQ_ACTIVE_INST_VALU"] / (cyc * CUS * 4),     # quad-cycles -> cycles, over 1024 SIMDs
         "valu_lane_utilisation": cnt["SQ_THREAD_CYCLES_VALU"] / (cnt["SQ_INSTS_VALU"] * 64.0),
         "valu_lane_throughput_frac": cnt["SQ_THREAD_CYCLES_VALU"] / (CUS * 64.0 * cyc)}
    if "TD_TD_BUSY_sum" in cnt:
        d["td_busy_frac"] = cnt["TD_TD_BUSY_sum"] / (cyc * CUS)
        d["ta_busy_frac"] = cnt["TA_TA_BUSY_sum"] / (cyc * CUS)
    if "TCC_HIT_sum" in cnt:
        d["l2_hit_rate"] = cnt["TCC_HIT_sum"] / max(1.0, cnt["TCC_HIT_sum"] + cnt["TCC_MISS_sum"])
    if per_unit:
        d["valu_instructions_per_ray"] = cnt["SQ_INSTS_VALU"] / per_unit
        d["vmem_read_instructions_per_ray"] = cnt["SQ_INSTS_VMEM_RD"] / per_unit
    return d


os.makedirs(P, exist_ok=True)
# ---- bench command
if kernel_stats("trace", f"{RND}_kernel_stats_bench.csv"):
    lines = [l for l in open(os.path.join(G, "bench_under_rocprof.log")) if l.startswith("{")]
    if lines:
        json.dump(json.loads(lines[-1]), open(os.path.join(P, f"{RND}_bench_line.json"), "w"), indent=1)

# ---- culled closest-hit kernel
cnt = {}
for d in ("c_fetch", "c_write", "c_l2", "c_sq1", "c_sq2", "c_mem"):
    cnt.update(sums(d, KERNEL))
what = "ptx::k_mesh_persistent<1,false> (culled closest-hit), 16777216 rays vs 1M-triangle soup"
cmd = "rocprofv3 --pmc <counters> -- python3 tools/prof_closest.py 16777216 2 c   (one pass per counter group: tools/profile_gpu.sh)"
if "FETCH_SIZE" in cnt:
    hit, miss = cnt.get("TCC_HIT_sum", 0.0), cnt.get("TCC_MISS_sum", 0.0)
    json.dump({
        "round": RND, "kernel": what, "command": cmd,
        "FETCH_SIZE_KB_per_launch": cnt["FETCH_SIZE"], "WRITE_SIZE_KB_per_launch": cnt.get("WRITE_SIZE"),
        "TCC_HIT_sum": hit, "TCC_MISS_sum": miss, "l2_hit_rate": hit / (hit + miss) if hit + miss else None,
        "correction": "KB = 1024 B.  FETCH_SIZE = 64 B per L2 read miss.  The gfx950 rule of doubling FETCH_SIZE holds for wide coalesced streaming reads, "
                      "whose 128-B requests are tallied at 64 B; this kernel reads one 64-byte node or 48-byte triangle record per lane, and on that pattern "
                      "the counter was calibrated against a known byte count (profiles/r01_fetch_calibration.json: 13.42 GB of random 64-B records requested, "
                      "FETCH_SIZE 13.78 GB): taken as is.  WRITE_SIZE taken as is.  The counters sit on the L2's fabric side, so Infinity Cache hits are included: "
                      "with a 69 MB working set this is an upper bound on DRAM traffic, not DRAM traffic.",
        "closest_culled_hbm_bytes_per_launch": (cnt["FETCH_SIZE"] + cnt.get("WRITE_SIZE", 0.0)) * 1024.0,
        "bytes_per_launch_if_fetch_size_were_doubled": (2.0 * cnt["FETCH_SIZE"] + cnt.get("WRITE_SIZE", 0.0)) * 1024.0,
    }, open(os.path.join(P, f"{RND}_traffic.json"), "w"), indent=1)
if "SQ_INSTS_VALU" in cnt and "GRBM_GUI_ACTIVE" in cnt:
    d = derived(cnt, N_RAYS)
    json.dump({"round": RND, "kernel": what, "command": cmd,
               "note": "per-launch sums over the chip; SQ_ACTIVE_INST_* count quad-cycles (x4 = cycles); busy fractions are relative to GRBM_GUI_ACTIVE/8 "
                       "(cycles per XCD) measured in the same pass; valu_lane_throughput_frac = SQ_THREAD_CYCLES_VALU / (256 CUs x 64 lanes x cycles)",
               "counters": cnt, "derived": d}, open(os.path.join(P, f"{RND}_pmc_closest.json"), "w"), indent=1)
    print("closest:", json.dumps(d, indent=1))

# ---- the other two ray queries of the bench: reference-order closest hit, culled occlusion
others = {}
for label, like in (("closest hit, reference order: k_mesh_persistent<0, false, false, false>", "k_mesh_persistent<0, false, false, false>"),
                    ("occlusion, culled order: k_mesh_persistent<1, true, false, false>", "k_mesh_persistent<1, true, false, false>")):
    c2 = {}
    for d in ("o_fetch", "o_l2", "o_sq", "o_mem"):
        c2.update(sums(d, like))
    if "SQ_INSTS_VALU" in c2 and "GRBM_GUI_ACTIVE" in c2:
        d2 = derived(c2, N_RAYS)
        d2["l1_access_rate_frac"] = c2["TCP_TOTAL_CACHE_ACCESSES_sum"] / (d2["cycles_per_xcd"] * CUS)
        d2["l1_accesses_per_ray"] = c2["TCP_TOTAL_CACHE_ACCESSES_sum"] / N_RAYS
        d2["fabric_read_bytes_per_launch"] = c2.get("FETCH_SIZE", 0.0) * 1024.0
        others[label] = {"counters": c2, "derived": d2}
if others:
    json.dump({"round": RND, "workload": "16777216 rays vs the 1M-triangle soup (closest-hit rays / shadow rays of the bench)",
               "command": "rocprofv3 --pmc <counters> -- python3 tools/prof_closest.py 16777216 2 ra   (one pass per counter group: tools/profile_gpu.sh)",
               "note": "per-launch sums; derived fractions as in the culled closest-hit summary; l1_access_rate_frac = TCP_TOTAL_CACHE_ACCESSES_sum / (256 CUs x cycles)",
               "kernels": others}, open(os.path.join(P, f"{RND}_pmc_other_queries.json"), "w"), indent=1)
    for k, e in others.items():
        print(k[:40], {a: round(b, 3) for a, b in e["derived"].items()})

# ---- the bench line's roofline block again, from the summaries just written (the run itself used the summaries
# committed before it: same formulas, this run's counters), so that the line and the files beside it agree exactly
line_path = os.path.join(P, f"{RND}_bench_line.json")
if os.path.exists(line_path) and "SQ_INSTS_VALU" in cnt:
    sys.path.insert(0, ROOT)
    import bench
    line = json.load(open(line_path))
    old = line["roofline"]
    line["roofline"] = bench.roofline_block(old["kernel_ms"], line["config"]["rays_per_gpu"], old["culled_traversal"],
                                            old["reference_semantics_equiv"]["nodes_per_ray"], old["reference_semantics_equiv"]["tris_per_ray"])
    json.dump(line, open(line_path, "w"), indent=1)

# ---- render kernels
rows = kernel_stats("r_trace", f"{RND}_kernel_stats_render.csv")
per_kernel = collections.defaultdict(dict)
for d in ("r_sq1", "r_sq2", "r_mem", "r_fetch", "r_l2"):
    for k, m in totals(d).items():
        if "rocclr" in k or "Cijk" in k:
            continue
        per_kernel[k].update(m)
out = {}
for k, m in per_kernel.items():
    e = {"counters_summed_over_dispatches": m}
    if "GRBM_GUI_ACTIVE" in m and "SQ_INSTS_VALU" in m and m.get("SQ_INSTS_VALU", 0) > 0:
        e["derived"] = derived(m)
    if "FETCH_SIZE" in m:
        e["fabric_read_bytes"] = m["FETCH_SIZE"] * 1024.0
    out[k] = e
if out:
    json.dump({"round": RND, "workload": "refract_cow_cbox.js3d 1024x1024, 16 spp, depth 8 (tools/prof_render.py 16): every kernel of ptx_render_tiles",
               "command": "rocprofv3 --pmc <counters> -- python3 tools/prof_render.py 16   (one pass per counter group: tools/profile_gpu.sh)",
               "note": "sums over all dispatches of each kernel in the render; derived fractions as in the closest-hit summary",
               "kernels": out}, open(os.path.join(P, f"{RND}_pmc_scene.json"), "w"), indent=1)
    for k, e in out.items():
        if "derived" in e:
            print(k[:70], {a: round(b, 3) for a, b in e["derived"].items()})
