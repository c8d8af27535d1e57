"""The committed measurement evidence hangs together: every fraction of the bench line's `roofline` block can be
recomputed from the counter summaries and the kernel trace under profiles/ (no GPU needed)."""
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


def test_roofline_fractions_recompute_from_the_profiles():
    line = json.load(open(latest("bench_line")))
    pmc = json.load(open(latest("pmc_closest")))["counters"]
    traffic = json.load(open(latest("traffic")))
    roof = line["roofline"]
    ms = roof["kernel_ms"]
    assert roof["sources"]["pmc"].endswith(os.path.basename(latest("pmc_closest")))
    # the counters are per launch of the first-pass culled closest-hit kernel: a launch walks all 16.7 M rays
    assert 300 < pmc["TCP_TOTAL_CACHE_ACCESSES_sum"] / N_RAYS < 500
    assert 300 < pmc["SQ_INSTS_VALU"] / N_RAYS < 450
    cycles = pmc["GRBM_GUI_ACTIVE"] / 8.0
    # HBM-side traffic over live kernel time against 8 TB/s
    bytes_per_launch = (traffic["FETCH_SIZE_KB_per_launch"] + traffic["WRITE_SIZE_KB_per_launch"]) * 1024.0
    assert roof["traffic"] == pytest.approx(bytes_per_launch, rel=1e-3)
    assert roof["achieved"] == pytest.approx(bytes_per_launch / (ms * 1e-3) / 1e9, rel=2e-3)
    assert roof["frac"] == pytest.approx(roof["achieved"] / roof["peak"], rel=2e-3) and roof["frac"] <= 1.0
    # the roofs that bind
    r = roof["roofs"]
    assert r["valu_issue"]["frac"] == pytest.approx(4.0 * pmc["SQ_ACTIVE_INST_VALU"] / (cycles * CUS * 4), rel=2e-3)
    assert r["l1_accesses"]["frac"] == pytest.approx(pmc["TCP_TOTAL_CACHE_ACCESSES_sum"] / (cycles * CUS), rel=2e-3)
    assert r["valu_lanes"]["frac"] == pytest.approx(pmc["SQ_THREAD_CYCLES_VALU"] / (CUS * 64.0 * cycles), rel=2e-2)
    assert r["hbm_fabric"]["l2_hit_rate"] == pytest.approx(pmc["TCC_HIT_sum"] / (pmc["TCC_HIT_sum"] + pmc["TCC_MISS_sum"]), rel=2e-3)
    assert all(v.get("frac", v.get("busy_frac")) <= 1.0 for v in r.values())
    assert roof["nearest_roof"] in r
    # the kernel trace of the same command agrees with the live timing
    rows = list(csv.reader(open(latest("kernel_stats_bench", "csv"))))[1:]
    avg_ns = [float(x[3]) for x in rows if "k_mesh_persistent<1, false, false, false>" in x[0]]
    assert len(avg_ns) == 1 and avg_ns[0] / 1e6 == pytest.approx(line["dominant_kernel_ms"], rel=0.02)
    # value is rays over step time
    assert line["value"] == pytest.approx(line["config"]["rays_per_gpu"] * line["n_gpus"] / (line["ms_per_step"] * 1e-3) / 1e6, rel=1e-3)
