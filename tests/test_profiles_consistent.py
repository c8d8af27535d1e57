"""The committed measurement evidence hangs together: every fraction of the bench line's `roofline` blocks can be
recomputed from the counter summaries and the kernel trace under profiles/ (no GPU needed), and the summaries say
which kernel sources they were taken from."""
import csv
import glob
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")
CUS, N_RAYS = 256, 16777216


def latest(stem, ext="json"):
    files = sorted(glob.glob(os.path.join(P, f"r[0-9][0-9]_{stem}.{ext}")))
    if not files:
        pytest.skip(f"no profiles/rNN_{stem}.{ext}")
    return files[-1]


def check_query_block(roof, pmc, traffic, n_rays):
    ms = roof["kernel_ms"]
    cycles = pmc["GRBM_GUI_ACTIVE"] / 8.0
    # HBM-side traffic over live kernel time against 8 TB/s
    bytes_per_launch = (traffic["FETCH_SIZE_KB_per_launch"] + traffic["WRITE_SIZE_KB_per_launch"]) * 1024.0
    assert roof["traffic"] == pytest.approx(bytes_per_launch, rel=1e-3)
    assert roof["achieved"] == pytest.approx(bytes_per_launch / (ms * 1e-3) / 1e9, rel=2e-3)
    assert roof["peak"] == 8000.0
    assert roof["frac"] == pytest.approx(roof["achieved"] / roof["peak"], rel=2e-3) and roof["frac"] <= 1.0
    r = roof["roofs"]
    assert r["valu_issue"]["frac"] == pytest.approx(4.0 * pmc["SQ_ACTIVE_INST_VALU"] / (cycles * CUS * 4), rel=2e-3)
    assert r["l1_accesses"]["frac"] == pytest.approx(pmc["TCP_TOTAL_CACHE_ACCESSES_sum"] / (cycles * CUS), rel=2e-3)
    assert r["valu_lanes"]["frac"] == pytest.approx(pmc["SQ_THREAD_CYCLES_VALU"] / (CUS * 64.0 * cycles), rel=2e-2)
    assert r["hbm_fabric"]["l2_hit_rate"] == pytest.approx(pmc["TCC_HIT_sum"] / (pmc["TCC_HIT_sum"] + pmc["TCC_MISS_sum"]), rel=2e-3)
    assert all(v.get("frac", v.get("busy_frac")) <= 1.0 for v in r.values())
    assert roof["nearest_roof"] in r and roof["nearest_roof_frac"] == max(v["frac"] for v in r.values() if "frac" in v)
    if "l2_requests" in r:
        assert r["l2_requests"]["achieved"] == pytest.approx(roof["culled_traversal"]["bytes_per_ray"] * n_rays / (ms * 1e-3) / 1e9, rel=2e-3)


def test_roofline_fractions_recompute_from_the_profiles():
    line = json.load(open(latest("bench_line")))
    pmc = json.load(open(latest("pmc_closest")))["counters"]
    traffic = json.load(open(latest("traffic")))
    roof = line["roofline"]
    assert roof["sources"]["pmc"]["file"].endswith(os.path.basename(latest("pmc_closest")))
    # the counters are per launch of the first-pass culled closest-hit kernel: a launch walks all 16.7 M rays
    assert 300 < pmc["TCP_TOTAL_CACHE_ACCESSES_sum"] / N_RAYS < 500
    assert 300 < pmc["SQ_INSTS_VALU"] / N_RAYS < 450
    check_query_block(roof, pmc, traffic, N_RAYS)
    # the kernel trace of the same command agrees with the live timing, leg by leg (the same kernel runs on three meshes)
    by_leg = {(e["leg"], e["query"]): e for e in json.load(open(latest("kernel_stats_bench_by_leg")))["legs"]}
    assert by_leg[("config 3 (1M-triangle soup)", "closest")]["avg_ms"] == pytest.approx(line["dominant_kernel_ms"], rel=0.02)
    for mesh, leg in line.get("closest_10m", {}).items():
        assert by_leg[(mesh, "closest")]["avg_ms"] == pytest.approx(leg["closest_kernel_ms"], rel=0.02), mesh
        assert by_leg[(mesh, "occlusion")]["avg_ms"] == pytest.approx(leg["anyhit_kernel_ms"], rel=0.02), mesh
    rows = list(csv.reader(open(latest("kernel_stats_bench", "csv"))))[1:]
    assert sum("k_mesh_persistent<1, false, false, false>" in x[0] for x in rows) == 1
    # value is rays over step time
    assert line["value"] == pytest.approx(line["config"]["rays_per_gpu"] * line["n_gpus"] / (line["ms_per_step"] * 1e-3) / 1e6, rel=1e-3)


def test_ten_million_triangle_blocks_recompute():
    line = json.load(open(latest("bench_line")))
    if "closest_10m" not in line:
        pytest.skip("the committed line has no 10 M-triangle legs")
    pmc = json.load(open(latest("pmc_closest_10m")))["meshes"]
    traffic = json.load(open(latest("traffic_10m")))["meshes"]
    for mesh, leg in line["closest_10m"].items():
        assert leg["triangles"] >= 10_000_000 and leg["parity"]["culled_vs_reference_mode_mismatches"] == 0
        assert leg["device_bytes"]["nodes"] + leg["device_bytes"]["triangle_geometry"] > 256 * (1 << 20)     # beyond the Infinity Cache
        check_query_block(leg["roofline"], pmc[mesh]["closest"]["counters"], traffic[mesh]["closest"], N_RAYS)
        check_query_block(leg["roofline_occlusion"], pmc[mesh]["occlusion"]["counters"], traffic[mesh]["occlusion"], N_RAYS)
        assert leg["closest_mrays_per_s"] == pytest.approx(N_RAYS / (leg["ms_per_step"] * 1e-3) / 1e6, rel=1e-3)
        assert leg["closest_kernel_ms"] <= leg["ms_per_step"]


def test_render_leg_blocks_recompute():
    line = json.load(open(latest("bench_line")))
    prof = json.load(open(latest("pmc_render")))
    seen = 0
    for r in line["renders"]:
        roof = r.get("roofline")
        if not roof or "achieved" not in roof:
            continue
        seen += 1
        entry = prof["legs"][r["leg"]]
        fabric = sum(k.get("fabric_bytes", 0.0) for k in entry["kernels"].values())
        assert entry["fabric_bytes_per_path"] == pytest.approx(fabric / entry["paths_profiled"], rel=1e-6)
        assert roof["achieved"] == pytest.approx(entry["fabric_bytes_per_path"] * r["paths"] / r["wall_s"] / 1e9, rel=2e-3)
        assert roof["frac"] == pytest.approx(roof["achieved"] / 8000.0, rel=2e-3) and roof["frac"] <= 1.0
        assert roof["rays_per_path"] == pytest.approx(r["traced_rays"] / r["paths"], rel=1e-3)
        assert roof["Mrays_per_s"] == pytest.approx(r["traced_rays"] / r["wall_s"] / 1e6, rel=2e-3)
        # the profiled render traced the same rays per path as the timed one (same kernels, same scene)
        assert f"{roof['rays_per_path']:.2f}"[:3] in entry["driver_output"]
        ref = roof.get("reference_semantics")
        if ref:
            assert ref["B_ray"] == pytest.approx(40 + 32 * ref["nodes_per_ray"] + 36 * ref["tris_per_ray"], rel=1e-2)
            assert ref["B_path"] == pytest.approx(ref["rays_per_path"] * ref["B_ray"] + 28, rel=1e-2)
            assert ref["rays_per_path"] >= roof["rays_per_path"]        # rays that cannot reach the image are left out
        for name, k in roof["kernels"].items():
            full = [n for n in entry["kernels"] if name in n.replace("ptx::", "")]
            assert full, name
            c = entry["kernels"][full[0]]["counters_summed_over_dispatches"]
            cycles = c["GRBM_GUI_ACTIVE"] / 8.0
            assert k["roofs"]["valu_issue"]["frac"] == pytest.approx(4.0 * c["SQ_ACTIVE_INST_VALU"] / (cycles * CUS * 4), rel=2e-3, abs=1e-4)   # fractions are kept to four decimals
            assert k["roofs"]["valu_lanes"]["frac"] == pytest.approx(c["SQ_THREAD_CYCLES_VALU"] / (CUS * 64.0 * cycles), rel=2e-2, abs=1e-4)
            assert all(v.get("frac", v.get("busy_frac", 0)) <= 1.0 for v in k["roofs"].values() if isinstance(v, dict))
    assert seen >= 4


def test_summaries_name_their_sources():
    """The newest counter summaries carry the hash of the kernel sources they were taken from, and the bench line
    records for each block whether that is what was timed."""
    for stem in ("traffic", "pmc_closest", "traffic_10m", "pmc_closest_10m", "pmc_render"):
        d = json.load(open(latest(stem)))
        assert d.get("source_hash") and len(d["source_hash"]) == 16, stem
        assert d.get("commit"), stem
    line = json.load(open(latest("bench_line")))
    assert line["roofline"]["sources"]["pmc"]["source_hash"]
    assert line["roofline"]["sources"]["pmc"]["sources_match"] in (True, False)
