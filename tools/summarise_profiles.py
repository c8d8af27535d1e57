"""Turns the rocprofv3 databases that tools/profile_gpu.sh left under gpurun_out/prof/ into the summaries kept under
profiles/ for one round:  python tools/summarise_profiles.py r03
  rNN_kernel_stats_bench.csv   per-kernel time of the default bench command (kernel trace)
  rNN_kernel_stats_bench_by_leg.json  the traversal kernels' launches of that trace split by the leg (mesh) they ran on
  rNN_bench_line.json          the line that command printed
  rNN_traffic.json             L2<->fabric bytes per launch of the culled closest-hit kernel (1M-triangle soup), L2 hit rate
  rNN_pmc_closest.json         per-launch counter sums of that kernel + derived busy fractions
  rNN_pmc_other_queries.json   the same for closest hit in the reference order (the API's default) and for occlusion queries
  rNN_traffic_10m.json         fabric bytes / L2 hit rate of the closest-hit and occlusion kernels on the two 10 M-triangle meshes
  rNN_pmc_closest_10m.json     counter sums + derived fractions of the same
  rNN_kernel_stats_render.csv  per-kernel time of a config-2 render (refract_cow_cbox 1024x1024, 256 spp)
  rNN_pmc_render.json          per benchmark render leg: counter sums of every kernel of a 16 spp render, fabric bytes per path
Every summary carries `source_hash` (pathtracer_amd/buildinfo.py, taken on the GPU box when the counters were collected)
and `commit` (git HEAD here when the summary was written).  Sections whose databases are missing are left as they are.
Run here after the gpurun call has merged gpurun_out/ back."""
import collections
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "gpurun_out", "prof")
P = os.path.join(ROOT, "profiles")
RND = sys.argv[1] if len(sys.argv) > 1 else "r03"
CLOSEST = "k_mesh_persistent<1, false, false, false>"      # <culled, closest, binary tree, first pass>
ANYHIT = "k_mesh_persistent<1, true, false, false>"
REFORDER = "k_mesh_persistent<0, false, false, false>"
N_RAYS = 16777216.0
CUS = 256
sys.path.insert(0, ROOT)


def stamp():
    try:
        src = open(os.path.join(G, "source_hash.txt")).read().strip()
    except OSError:
        src = None
    try:
        commit = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short=12", "HEAD"], text=True).strip()
        dirty = bool(subprocess.check_output(["git", "-C", ROOT, "status", "--porcelain", "--", "pathtracer_amd/csrc"], text=True).strip())
    except (OSError, subprocess.CalledProcessError):
        commit, dirty = None, None
    return {"round": RND, "source_hash": src, "commit": commit, "csrc_modified_since_commit": dirty}


def reduced(path):
    """The reduction tools/reduce_profiles.py left of one profiler pass (gpurun_out/prof/<path>.json), or None."""
    f = os.path.join(G, path + ".json")
    return json.load(open(f)) if os.path.exists(f) else None


def kernel_stats(path, out_csv):
    r = reduced(path)
    if not r or "kernel_stats" not in r:
        return None
    rows = r["kernel_stats"]
    tot = sum(x[2] for x in rows)
    with open(os.path.join(P, out_csv), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for x in rows:
            w.writerow([x[0], x[1], x[2], round(x[3], 1), round(100.0 * x[2] / tot, 3), x[4], x[5]])
    return rows


def sums(path, like):
    """Per-launch mean of every counter over the dispatches of kernels whose name matches."""
    r = reduced(path)
    out, n = {}, 0
    for k, m in ((r or {}).get("counters") or {}).items():
        if like in k:
            n += m.get("dispatches", 0)
            for c, v in m.items():
                if c != "dispatches":
                    out[c] = out.get(c, 0.0) + v
    return {c: v / max(n, 1) for c, v in out.items()}


def totals(path):
    """Per kernel: sum of every counter over all dispatches, and the dispatch count."""
    r = reduced(path)
    return {k: dict(m) for k, m in ((r or {}).get("counters") or {}).items()}


def derived(cnt, per_unit=None):
    cyc = cnt["GRBM_GUI_ACTIVE"] / 8.0          # the counter sums the eight XCDs
    d = {"cycles_per_xcd": cyc,
         "valu_busy_frac": 4.0 * cnt["SQ_ACTIVE_INST_VALU"] / (cyc * CUS * 4),     # quad-cycles -> cycles, over 1024 SIMDs
         "valu_lane_utilisation": cnt["SQ_THREAD_CYCLES_VALU"] / (cnt["SQ_INSTS_VALU"] * 64.0),
         "valu_lane_throughput_frac": cnt["SQ_THREAD_CYCLES_VALU"] / (CUS * 64.0 * cyc)}
    if "TD_TD_BUSY_sum" in cnt:
        d["td_busy_frac"] = cnt["TD_TD_BUSY_sum"] / (cyc * CUS)
        d["ta_busy_frac"] = cnt["TA_TA_BUSY_sum"] / (cyc * CUS)
    if "TCP_TOTAL_CACHE_ACCESSES_sum" in cnt:
        d["l1_access_rate_frac"] = cnt["TCP_TOTAL_CACHE_ACCESSES_sum"] / (cyc * CUS)
    if "TCC_HIT_sum" in cnt:
        d["l2_hit_rate"] = cnt["TCC_HIT_sum"] / max(1.0, cnt["TCC_HIT_sum"] + cnt["TCC_MISS_sum"])
    if per_unit:
        d["valu_instructions_per_ray"] = cnt["SQ_INSTS_VALU"] / per_unit
        if "SQ_INSTS_VMEM_RD" in cnt:
            d["vmem_read_instructions_per_ray"] = cnt["SQ_INSTS_VMEM_RD"] / per_unit
        if "TCP_TOTAL_CACHE_ACCESSES_sum" in cnt:
            d["l1_accesses_per_ray"] = cnt["TCP_TOTAL_CACHE_ACCESSES_sum"] / per_unit
    return d


FETCH_NOTE = ("KB = 1024 B.  FETCH_SIZE = 64 B per L2 read miss.  The gfx950 rule of doubling FETCH_SIZE holds for wide coalesced streaming reads, "
              "whose 128-B requests are tallied at 64 B; these kernels read one 64-byte node or 48-byte triangle record per lane, and on that pattern "
              "the counter was calibrated against a known byte count (profiles/r01_fetch_calibration.json: 13.42 GB of random 64-B records requested, "
              "FETCH_SIZE 13.78 GB): taken as is.  WRITE_SIZE taken as is.  The counters sit on the L2's fabric side, so Infinity Cache hits are included: "
              "an upper bound on DRAM traffic, not DRAM traffic.")


def traffic_entry(cnt):
    hit, miss = cnt.get("TCC_HIT_sum", 0.0), cnt.get("TCC_MISS_sum", 0.0)
    return {"FETCH_SIZE_KB_per_launch": cnt["FETCH_SIZE"], "WRITE_SIZE_KB_per_launch": cnt.get("WRITE_SIZE"),
            "TCC_HIT_sum": hit, "TCC_MISS_sum": miss, "l2_hit_rate": hit / (hit + miss) if hit + miss else None,
            "hbm_bytes_per_launch": (cnt["FETCH_SIZE"] + cnt.get("WRITE_SIZE", 0.0)) * 1024.0,
            "bytes_per_launch_if_fetch_size_were_doubled": (2.0 * cnt["FETCH_SIZE"] + cnt.get("WRITE_SIZE", 0.0)) * 1024.0}


os.makedirs(P, exist_ok=True)
STAMP = stamp()

# ---- bench command
if kernel_stats("trace", f"{RND}_kernel_stats_bench.csv"):
    lines = [l for l in open(os.path.join(G, "bench_under_rocprof.log")) if l.startswith("{")]
    if lines:
        json.dump(json.loads(lines[-1]), open(os.path.join(P, f"{RND}_bench_line.json"), "w"), indent=1)
    # The command runs the culled closest-hit kernel on three meshes in turn (the 1M-triangle soup, then the two
    # 10 M-triangle meshes), seconds apart (scene construction, CPU baselines): its launches split by the gaps.
    launches = reduced("trace")["traversal_launches"]

    def launches_of(like):
        return sorted((st, en) for name, rows in launches.items() if like in name for st, en in rows)

    by_leg = []
    legs3 = ("config 3 (1M-triangle soup)", "cows10m", "soup10m")
    windows = []     # a leg's window opens with its first closest-hit launch
    for st, en in launches_of(CLOSEST):
        if not windows or st - last_end > 2.0e9:
            windows.append(st)
        last_end = en
    for like, label in ((CLOSEST, "closest"), (ANYHIT, "occlusion")):
        rows = launches_of(like)
        for k, leg in enumerate(legs3[:len(windows)]):
            lo, hi = windows[k], windows[k + 1] if k + 1 < len(windows) else float("inf")
            d = [(en - st) / 1e6 for st, en in rows if lo <= st < hi]
            if d:
                by_leg.append({"kernel": like, "query": label, "leg": leg, "launches": len(d), "avg_ms": sum(d) / len(d), "min_ms": min(d), "max_ms": max(d)})
    json.dump({**STAMP, "what": "launches of the culled closest-hit and occlusion kernels in the traced bench command, split by leg (a leg opens with the first closest-hit launch after a pause of more than 2 s: scene construction and ray generation lie between legs)",
               "legs": by_leg}, open(os.path.join(P, f"{RND}_kernel_stats_bench_by_leg.json"), "w"), indent=1)

# ---- culled closest-hit kernel, 1M-triangle soup
cnt = {}
for d in ("c_fetch", "c_write", "c_l2", "c_sq1", "c_sq2", "c_mem"):
    cnt.update(sums(d, CLOSEST))
what = "ptx::k_mesh_persistent<1,false> (culled closest-hit), 16777216 rays vs 1M-triangle soup"
cmd = "rocprofv3 --pmc <counters> -- python3 tools/prof_closest.py 16777216 2 c soup1m   (one pass per counter group: tools/profile_gpu.sh closest)"
if "FETCH_SIZE" in cnt:
    e = traffic_entry(cnt)
    e["closest_culled_hbm_bytes_per_launch"] = e["hbm_bytes_per_launch"]
    json.dump({**STAMP, "kernel": what, "command": cmd, **e, "correction": FETCH_NOTE}, open(os.path.join(P, f"{RND}_traffic.json"), "w"), indent=1)
if "SQ_INSTS_VALU" in cnt and "GRBM_GUI_ACTIVE" in cnt:
    d = derived(cnt, N_RAYS)
    json.dump({**STAMP, "kernel": what, "command": cmd,
               "note": "per-launch sums over the chip; SQ_ACTIVE_INST_* count quad-cycles (x4 = cycles); busy fractions are relative to GRBM_GUI_ACTIVE/8 "
                       "(cycles per XCD) measured in the same pass; valu_lane_throughput_frac = SQ_THREAD_CYCLES_VALU / (256 CUs x 64 lanes x cycles)",
               "counters": cnt, "derived": d}, open(os.path.join(P, f"{RND}_pmc_closest.json"), "w"), indent=1)
    print("closest:", json.dumps(d, indent=1))

# ---- the other two ray queries of the bench: reference-order closest hit, culled occlusion
others = {}
for label, like in (("closest hit, reference order: " + REFORDER, REFORDER), ("occlusion, culled order: " + ANYHIT, ANYHIT)):
    c2 = {}
    for d in ("o_fetch", "o_l2", "o_sq", "o_mem"):
        c2.update(sums(d, like))
    if "SQ_INSTS_VALU" in c2 and "GRBM_GUI_ACTIVE" in c2:
        d2 = derived(c2, N_RAYS)
        d2["fabric_read_bytes_per_launch"] = c2.get("FETCH_SIZE", 0.0) * 1024.0
        others[label] = {"counters": c2, "derived": d2}
if others:
    json.dump({**STAMP, "workload": "16777216 rays vs the 1M-triangle soup (closest-hit rays / shadow rays of the bench)",
               "command": "rocprofv3 --pmc <counters> -- python3 tools/prof_closest.py 16777216 2 ra soup1m   (one pass per counter group: tools/profile_gpu.sh other)",
               "note": "per-launch sums; derived fractions as in the culled closest-hit summary; l1_access_rate_frac = TCP_TOTAL_CACHE_ACCESSES_sum / (256 CUs x cycles)",
               "kernels": others}, open(os.path.join(P, f"{RND}_pmc_other_queries.json"), "w"), indent=1)
    for k, e in others.items():
        print(k[:40], {a: round(b, 3) for a, b in e["derived"].items()})

# ---- the 10 M-triangle meshes: culled closest hit and occlusion
traffic10, pmc10 = {}, {}
for mesh, label in (("cows10m", "1,728 cows baked into one 10,119,168-triangle mesh (tests/scenes.py flattened_cows(12))"),
                    ("soup10m", "the soup generator at 10,000,000 triangles (pathtracer_amd/synth.py soup(10_000_000))")):
    for query, like in (("closest", CLOSEST), ("occlusion", ANYHIT)):
        c3 = {}
        for d in ("fetch", "write", "l2", "sq1", "sq2", "mem"):
            c3.update(sums(f"m_{mesh}_{d}", like))
        if "FETCH_SIZE" in c3:
            traffic10.setdefault(mesh, {"mesh": label})[query] = traffic_entry(c3)
        if "SQ_INSTS_VALU" in c3 and "GRBM_GUI_ACTIVE" in c3:
            pmc10.setdefault(mesh, {"mesh": label})[query] = {"counters": c3, "derived": derived(c3, N_RAYS)}
            print(mesh, query, {a: round(b, 3) for a, b in pmc10[mesh][query]["derived"].items()})
cmd10 = "rocprofv3 --pmc <counters> -- python3 tools/prof_closest.py 16777216 2 ca <cows10m|soup10m>   (one pass per counter group: tools/profile_gpu.sh closest10m)"
if traffic10:
    json.dump({**STAMP, "workload": "16777216 rays (origins uniform in the mesh's box, directions uniform) per launch, culled order", "command": cmd10, "correction": FETCH_NOTE,
               "meshes": traffic10}, open(os.path.join(P, f"{RND}_traffic_10m.json"), "w"), indent=1)
if pmc10:
    json.dump({**STAMP, "workload": "16777216 rays per launch, culled order", "command": cmd10,
               "note": "per-launch sums; derived fractions as in the 1M-triangle summary", "meshes": pmc10}, open(os.path.join(P, f"{RND}_pmc_closest_10m.json"), "w"), indent=1)

# ---- render kernels
kernel_stats("r_trace", f"{RND}_kernel_stats_render.csv")
LEGS = {"cbox": (256, 16, "config 1: cbox.js3d 256x256, depth 4"), "refract_cow_cbox": (1024, 16, "config 2 / 4: refract_cow_cbox.js3d 1024x1024, depth 8"),
        "instanced_cows": (1024, 16, "config 5, instanced: 1,728 instances of the cow, 1024x1024, depth 8"),
        "flat_cows": (1024, 16, "config 5, flattened: the cows baked into one 10.1 M-triangle mesh, 1024x1024, depth 8")}
legs = {}
for leg, (w, spp, label) in LEGS.items():
    per_kernel = collections.defaultdict(dict)
    for d in ("sq1", "sq2", "mem", "fetch", "write", "l2"):
        for k, m in totals(f"r_{leg}_{d}").items():
            if "rocclr" in k or "Cijk" in k or "k_gb_" in k or "(anonymous namespace)" in k:   # copies, library kernels, the tree builder (scene creation)
                continue
            per_kernel[k].update(m)
    out = {}
    fabric = 0.0
    for k, m in per_kernel.items():
        e = {"counters_summed_over_dispatches": m}
        if "GRBM_GUI_ACTIVE" in m and m.get("SQ_INSTS_VALU", 0) > 0:
            e["derived"] = derived(m)
        if "FETCH_SIZE" in m:
            e["fabric_bytes"] = (m["FETCH_SIZE"] + m.get("WRITE_SIZE", 0.0)) * 1024.0
            fabric += e["fabric_bytes"]
        out[k] = e
    if out:
        paths = w * w * (spp + 1)      # the driver renders one warm-up sample per pixel before the spp samples
        log = [l for l in open(os.path.join(G, f"r_{leg}_sq1.log")) if "rays per path" in l]
        legs[leg] = {"workload": label, "paths_profiled": paths, "fabric_bytes_per_path": fabric / paths if fabric else None,
                     "driver_output": log[-1].strip() if log else None, "kernels": out}
        for k, e in out.items():
            if "derived" in e and e["counters_summed_over_dispatches"].get("GRBM_GUI_ACTIVE", 0) > 0.02 * sum(x["counters_summed_over_dispatches"].get("GRBM_GUI_ACTIVE", 0) for x in out.values()):
                print(leg, k[:60], {a: round(b, 3) for a, b in e["derived"].items()})
if legs:
    json.dump({**STAMP, "command": "rocprofv3 --pmc <counters> -- python3 tools/prof_render.py 16 <leg>   (one pass per counter group: tools/profile_gpu.sh render)",
               "note": "sums over all dispatches of each kernel in one render (16 spp plus the driver's 1-spp warm-up); derived fractions as in the closest-hit summary; "
                       "fabric_bytes = (FETCH_SIZE + WRITE_SIZE) x 1024, taken as is (see the traffic summaries); fabric_bytes_per_path = sum over kernels / paths",
               "legs": legs}, open(os.path.join(P, f"{RND}_pmc_render.json"), "w"), indent=1)

# ---- the bench line's roofline blocks again, from the summaries just written (the run itself used the summaries
# committed before it: same formulas, this run's counters), so that the line and the files beside it agree exactly
line_path = os.path.join(P, f"{RND}_bench_line.json")
if os.path.exists(line_path):
    import bench
    line = json.load(open(line_path))
    bench.refresh_roofline_blocks(line)
    json.dump(line, open(line_path, "w"), indent=1)
