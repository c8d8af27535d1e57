"""Roofline blocks of the benchmark line.

Timings come from the run that prints the line (HIP events around each launch, wall clocks around each render); hardware
counters cannot be read from inside that run, so they come from the newest committed summaries under profiles/
(separate rocprofv3 --pmc passes over the same kernels and workloads, tools/profile_gpu.sh).  Every block names its
sources and says whether the summary was taken from the kernel sources now in the tree (`sources_match`).

Peaks: HBM3E 8 TB/s, aggregate L2 34.5 TB/s, 256 CUs x 4 SIMDs, one vector-L1 line lookup per CU per clock.
"""
import glob
import json
import os
import sys

from . import buildinfo

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HBM_PEAK_GBS = 8000.0    # MI355X HBM3E spec peak
L2_PEAK_GBS = 34500.0    # aggregate L2 bandwidth of the eight XCDs
CUS, LANES_PER_CU_CYCLE = 256, 64   # four 16-lane SIMDs per CU: 64 lane-operations per CU per cycle

_warned = set()


def latest_profile(stem):
    """(contents, repo-relative path) of the newest profiles/rNN_<stem>.json, or (None, None)."""
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r[0-9][0-9]_{stem}.json")))
    return (json.load(open(files[-1])), os.path.relpath(files[-1], ROOT)) if files else (None, None)


def source_of(profile, path):
    """What a block records about a counter summary: file, the commit and source hash it was taken at, and whether
    that hash is the one of the sources in the tree now.  A mismatch is reported on stderr once per file."""
    if profile is None:
        return None
    now = buildinfo.csrc_hash()
    match = profile.get("source_hash") == now if profile.get("source_hash") else None
    if match is not True and path not in _warned:
        _warned.add(path)
        why = "carries no source hash" if match is None else f"was taken at sources {profile.get('source_hash')}, the tree holds {now}"
        print(f"[roofline] WARNING: {path} {why}: its counters may belong to other kernels than the ones timed now", file=sys.stderr, flush=True)
    return {"file": path, "commit": profile.get("commit"), "source_hash": profile.get("source_hash"), "sources_match": match}


def counter_roofs(c, sec=None, n_rays=None):
    """Roofs that follow from one kernel's counter sums `c` (any consistent scope: per launch or per render)."""
    roofs = {}
    if "GRBM_GUI_ACTIVE" not in c:
        return roofs
    cycles = c["GRBM_GUI_ACTIVE"] / 8.0                                  # the counter sums the eight XCDs
    if "SQ_ACTIVE_INST_VALU" in c:
        roofs["valu_issue"] = {"frac": round(4.0 * c["SQ_ACTIVE_INST_VALU"] / (cycles * CUS * 4), 4),   # quad-cycles -> cycles, over 1024 SIMDs
                               "SQ_ACTIVE_INST_VALU": c["SQ_ACTIVE_INST_VALU"], "cycles_per_xcd": cycles}
    if "SQ_THREAD_CYCLES_VALU" in c and c.get("SQ_INSTS_VALU"):
        roofs["valu_lanes"] = {"frac": round(c["SQ_THREAD_CYCLES_VALU"] / (CUS * LANES_PER_CU_CYCLE * cycles), 4), "SQ_THREAD_CYCLES_VALU": c["SQ_THREAD_CYCLES_VALU"],
                               "lanes_active_per_instruction": round(c["SQ_THREAD_CYCLES_VALU"] / c["SQ_INSTS_VALU"], 1)}
        if sec:
            roofs["valu_lanes"].update(achieved=round(c["SQ_THREAD_CYCLES_VALU"] / sec / 1e12, 2), peak=round(CUS * LANES_PER_CU_CYCLE * 2.4e9 / 1e12, 2), unit="T lane-ops/s at 2.4 GHz")
    if "TCP_TOTAL_CACHE_ACCESSES_sum" in c:
        # every lane of a traversal step reads its own node or triangle record: four 16-byte accesses per node, three
        # per triangle, each a line lookup of its own in the CU's vector L1, which serves about one per clock
        acc = c["TCP_TOTAL_CACHE_ACCESSES_sum"]
        roofs["l1_accesses"] = {"frac": round(acc / (cycles * CUS), 4), "TCP_TOTAL_CACHE_ACCESSES_sum": acc, "peak": "one vector-L1 access per clock per CU"}
        if n_rays:
            roofs["l1_accesses"]["accesses_per_ray"] = round(acc / n_rays, 1)
    if "TD_TD_BUSY_sum" in c:
        roofs["vector_memory_return_busy"] = {"busy_frac": round(c["TD_TD_BUSY_sum"] / (cycles * CUS), 4), "TD_TD_BUSY_sum": c["TD_TD_BUSY_sum"]}
    return roofs


def nearest(roofs):
    best = max((k for k in roofs if isinstance(roofs[k], dict) and "frac" in roofs[k]), key=lambda k: roofs[k]["frac"])
    return best, roofs[best]["frac"]


def query_block(kernel, kernel_ms, n, work, ref_nodes_per_ray, ref_tris_per_ray, traffic, traffic_src, pmc, pmc_src, resident_note):
    """Roofs of one ray-query kernel launch (k_mesh_persistent, culled order) over `n` rays.
      hbm_fabric   L2<->fabric bytes (FETCH_SIZE + WRITE_SIZE) / HBM peak: an upper bound on DRAM traffic, since
                   Infinity Cache hits are counted
      l2_requests  bytes the traversal requests (64 B per two-child node, 48 B per triangle, ray, hit, order index;
                   device-side visit counts of this run) / L2 aggregate peak
      valu_issue   cycles in which a SIMD issues a vector-ALU instruction / all SIMD cycles
      valu_lanes   lane-operations executed / (CUs x 64 lanes x cycles)
      l1_accesses  vector-L1 accesses (one per lane per 16-byte load of its own record) / (CUs x cycles)
    `frac` of the block is hbm_fabric's (the memory roof); `nearest_roof` says which one binds."""
    sec = kernel_ms * 1e-3
    roofs = {}
    fabric_bytes = traffic.get("hbm_bytes_per_launch", traffic.get("closest_culled_hbm_bytes_per_launch")) if traffic else None
    if fabric_bytes:
        gbs = fabric_bytes / sec / 1e9
        roofs["hbm_fabric"] = {"achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
                               "bytes_per_launch": fabric_bytes, "l2_hit_rate": round(traffic.get("l2_hit_rate") or 0.0, 4)}
    if work:
        req_gbs = work["bytes_per_ray"] * n / sec / 1e9
        roofs["l2_requests"] = {"achieved": round(req_gbs, 1), "peak": L2_PEAK_GBS, "unit": "GB/s", "frac": round(req_gbs / L2_PEAK_GBS, 4),
                                "bytes_per_ray": work["bytes_per_ray"], "bytes_per_launch": round(work["bytes_per_ray"] * n)}
    if pmc:
        roofs.update(counter_roofs(pmc["counters"], sec, n))
    if not roofs:
        return None
    near, near_frac = nearest(roofs)
    primary = roofs.get("hbm_fabric") or roofs.get("l2_requests")
    block = {"bound": "hbm", "achieved": primary["achieved"], "peak": primary["peak"], "unit": "GB/s", "frac": primary["frac"],
             "traffic": fabric_bytes, "kernel": kernel, "kernel_ms": round(kernel_ms, 4), "rays_per_launch": int(n),
             "nearest_roof": near, "nearest_roof_frac": near_frac, "roofs": roofs,
             "sources": {"kernel_ms": "HIP events around each launch, this run", "traffic": source_of(traffic, traffic_src) if traffic else None,
                         "pmc": source_of(pmc, pmc_src) if pmc else None, "visit_counts": "device-side counters, 64k-ray prefix, this run"},
             "culled_traversal": work}
    if ref_nodes_per_ray is not None:
        ref_bytes = 32 + 8 + 32 * ref_nodes_per_ray + 36 * ref_tris_per_ray
        block["reference_semantics_equiv"] = {
            "bytes_per_ray": round(ref_bytes, 1), "nodes_per_ray": round(ref_nodes_per_ray, 2), "tris_per_ray": round(ref_tris_per_ray, 2), "GBps": round(ref_bytes * n / sec / 1e9, 1),
            "note": "bytes a traversal with the reference's visit semantics (no culling: every node whose ancestors pass the slab test, 32 B per box + 36 B per triangle, "
                    "counts from the CPU checker) would move for the same batch, over this kernel's time: an equivalent-work figure, not bytes this kernel moves"}
    block["note"] = ("achieved/peak/frac = L2<->fabric bytes per launch over live kernel time against HBM peak (Infinity Cache hits included, so an upper bound on DRAM traffic); "
                     "nearest_roof names the roof that binds. " + resident_note)
    return block


def _stamped(profile, entry):
    """An entry of a per-mesh / per-leg summary with the file's stamps, as source_of() expects them."""
    return dict(entry, source_hash=profile.get("source_hash"), commit=profile.get("commit")) if entry is not None else None


def soup_1m_block(kernel_ms, n, work, ref_nodes, ref_tris):
    traffic, tsrc = latest_profile("traffic")
    pmc, psrc = latest_profile("pmc_closest")
    return query_block("k_mesh_persistent<culled, closest>", kernel_ms, n, work, ref_nodes, ref_tris, traffic, tsrc, pmc, psrc,
                       "Nodes + triangles (69 MB) stay in L2 / Infinity Cache: the vector L1's access rate (one record piece per lane per clock) and vector-ALU issue are both "
                       "near saturation, with under half the lanes active per instruction; DRAM is not.")


def mesh_10m_block(mesh, query, kernel_ms, n, work, ref_nodes, ref_tris):
    """`mesh`: cows10m | soup10m; `query`: closest | occlusion."""
    traffic, tsrc = latest_profile("traffic_10m")
    pmc, psrc = latest_profile("pmc_closest_10m")
    t = _stamped(traffic, (traffic or {}).get("meshes", {}).get(mesh, {}).get(query))
    p = _stamped(pmc, (pmc or {}).get("meshes", {}).get(mesh, {}).get(query))
    kernel = "k_mesh_persistent<culled, closest>" if query == "closest" else "k_mesh_persistent<culled, any-hit>"
    return query_block(kernel, kernel_ms, n, work, ref_nodes, ref_tris, t, tsrc, p, psrc,
                       "Nodes + triangles (about 700 MB) exceed the 256 MB Infinity Cache: this is the workload on which the tree lives in HBM.")


def render_block(leg, paths, wall_s, traced_rays, reference=None):
    """Roofline of one render leg.  `traced_rays`: rays the extend rounds traced (device counters, this run);
    `reference`: {rays_per_path, nodes_per_ray, tris_per_ray} of the reference's own recursion on a sample of the same
    render (CPU checker), for the algorithmic-bytes figure  B_path = sum over rays (32 + 8 + 32 N_nodes + 36 N_tris) + 28."""
    prof, src = latest_profile("pmc_render")
    entry = (prof or {}).get("legs", {}).get(leg)
    block = {"rays_per_path": round(traced_rays / paths, 3) if traced_rays else None,
             "Mrays_per_s": round(traced_rays / wall_s / 1e6, 1) if traced_rays else None,
             "sources": {"wall_s": "this run", "rays": "device counters, this run (ptx_get_counters)", "pmc": source_of(prof, src) if entry else None}}
    if reference:
        b_ray = 32 + 8 + 32 * reference["nodes_per_ray"] + 36 * reference["tris_per_ray"]
        b_path = reference["rays_per_path"] * b_ray + 28
        block["reference_semantics"] = {"rays_per_path": round(reference["rays_per_path"], 3), "nodes_per_ray": round(reference["nodes_per_ray"], 1),
                                        "tris_per_ray": round(reference["tris_per_ray"], 1), "B_ray": round(b_ray, 1), "B_path": round(b_path, 1),
                                        "GBps_equiv": round(b_path * paths / wall_s / 1e9, 1),
                                        "note": "the reference's recursion on a sample of the same render (CPU checker): every ray it traces, every node and triangle its "
                                                "uncut traversal visits; an equivalent-work figure -- this pipeline leaves out rays that cannot reach the image and culls by distance"}
    if not entry:
        return block
    if entry.get("fabric_bytes_per_path"):
        gbs = entry["fabric_bytes_per_path"] * paths / wall_s / 1e9
        block.update(bound="hbm", achieved=round(gbs, 1), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(gbs / HBM_PEAK_GBS, 4),
                     traffic_per_path=round(entry["fabric_bytes_per_path"], 1))
    kernels = entry["kernels"]
    total_cycles = sum(k["counters_summed_over_dispatches"].get("GRBM_GUI_ACTIVE", 0.0) for k in kernels.values()) or 1.0
    per_kernel = {}
    for name, k in kernels.items():
        c = k["counters_summed_over_dispatches"]
        share = c.get("GRBM_GUI_ACTIVE", 0.0) / total_cycles
        if share < 0.05:
            continue
        roofs = counter_roofs(c)
        if "TCC_HIT_sum" in c:
            roofs["l2_hit_rate"] = round(c["TCC_HIT_sum"] / max(1.0, c["TCC_HIT_sum"] + c["TCC_MISS_sum"]), 4)
        if k.get("fabric_bytes"):
            roofs["fabric_bytes_per_path"] = round(k["fabric_bytes"] / entry["paths_profiled"], 1)
        near, frac = nearest(roofs)
        per_kernel[short_kernel_name(name)] = {"share_of_gpu_cycles": round(share, 3), "nearest_roof": near, "nearest_roof_frac": frac, "roofs": roofs}
    block["kernels"] = per_kernel
    if per_kernel:
        top = max(per_kernel, key=lambda k: per_kernel[k]["share_of_gpu_cycles"])
        block["nearest_roof"] = f"{per_kernel[top]['nearest_roof']} of {top}"
        block["nearest_roof_frac"] = per_kernel[top]["nearest_roof_frac"]
    block["note"] = ("achieved = fabric-side bytes per path (FETCH_SIZE + WRITE_SIZE over every kernel of a profiled 16 spp render of this leg) x this run's paths / this run's wall time, "
                     "against HBM peak; per-kernel roofs are fractions of that kernel's own busy cycles in the profiled render")
    return block


def short_kernel_name(name):
    """`void ptx::k_shade<false>(ptx::SceneRec, ...)` -> `k_shade<false>`."""
    name = name.replace("ptx::", "").replace("void ", "")
    depth = 0
    for i, ch in enumerate(name):
        depth += ch == "<"
        depth -= ch == ">"
        if ch == "(" and depth == 0:
            return name[:i].strip()
    return name.strip()


def refresh(line):
    """Recomputes every roofline block of a benchmark line from the newest summaries (same formulas, same timings):
    used after a profiling session, so that the committed line and the summaries beside it agree exactly."""
    if "roofline" in line and line["roofline"]:
        old = line["roofline"]
        ref = old.get("reference_semantics_equiv", {})
        line["roofline"] = soup_1m_block(old["kernel_ms"], old.get("rays_per_launch", line["config"]["rays_per_gpu"]), old["culled_traversal"], ref.get("nodes_per_ray"), ref.get("tris_per_ray"))
    n = line["config"]["rays_per_gpu"]
    for mesh, leg in (line.get("closest_10m") or {}).items():
        for query, key, ms_key in (("closest", "roofline", "closest_kernel_ms"), ("occlusion", "roofline_occlusion", "anyhit_kernel_ms")):
            old = leg.get(key) or {}
            ref = old.get("reference_semantics_equiv", {})
            leg[key] = mesh_10m_block(mesh, query, old.get("kernel_ms", leg[ms_key]), old.get("rays_per_launch", n), old.get("culled_traversal"), ref.get("nodes_per_ray"), ref.get("tris_per_ray"))
    for r in line.get("renders", []):
        if r.get("leg") and r.get("traced_rays") is not None:
            ref = (r.get("roofline") or {}).get("reference_semantics")
            r["roofline"] = render_block(r["leg"], r["paths"], r["wall_s"], r["traced_rays"], ref and {k: ref[k] for k in ("rays_per_path", "nodes_per_ray", "tris_per_ray")})
    return line
