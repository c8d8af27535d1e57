"""bench.py's multi-rank path with real kernels: two ranks sharing the one GPU of the box (--rehearse: gloo collectives,
rank r on device r % device_count), started as fresh child processes the way `python bench.py --gpus 2` starts them.
The integer frame reduce makes every render leg's mean radiance equal to the single-rank run's to the last digit
(pathtracer.h:80-83: fixed-point sums do not depend on who adds what first)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ARGS = ["--no-4k", "--no-cpu", "--rays", "1048576", "--tris", "100000", "--steps", "2", "--warmup", "1"]


def run_bench(extra):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra + ARGS, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-4000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


def test_two_ranks_render_the_single_rank_frame():
    one = run_bench([])
    two = run_bench(["--gpus", "2", "--rehearse"])
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2 and two["rccl_ranks"] == 2
    assert len(two["rank_devices"]) == 2 and {r["rank"] for r in two["rank_devices"]} == {0, 1}
    for line in (one, two):
        assert line["parity"]["culled_vs_reference_mode_mismatches"] == 0 and line["parity"]["anyhit_culled_vs_reference_mismatches"] == 0
        for mesh in ("cows10m", "soup10m"):
            p = line["closest_10m"][mesh]["parity"]
            assert p["culled_vs_reference_mode_mismatches"] == 0 and p["anyhit_culled_vs_reference_mismatches"] == 0
            assert line["closest_10m"][mesh]["triangles"] >= 10_000_000
    legs1 = {r["leg"]: r for r in one["renders"]}
    legs2 = {r["leg"]: r for r in two["renders"]}
    assert set(legs1) == set(legs2) == {"cbox", "refract_cow_cbox", "instanced_cows", "flat_cows"}
    for leg, r2 in legs2.items():
        r1 = legs1[leg]
        assert r2["mean_radiance"] == r1["mean_radiance"], leg          # integer reduce: the same frame
        assert r2["traced_rays"] == r1["traced_rays"], leg               # the same rays, dealt to two ranks
        assert len(r2["per_rank"]) == 2 and sum(r["tiles"] for r in r2["per_rank"]) > 0
        assert all(r["compute_s"] > 0 and r["reduce_ms"] >= 0 for r in r2["per_rank"])
        assert r2["compute_s"] == max(r["compute_s"] for r in r2["per_rank"])
        assert r2["frame_reduce"].startswith("gloo")
    # every render leg carries its rays per path and a roofline block
    for r in one["renders"]:
        assert r["roofline"]["rays_per_path"] > 1.0 and r["roofline"]["Mrays_per_s"] > 0


def test_ranks_started_by_torch_distributed_run():
    """The other way the bench is started: `python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2` hands
    it RANK / LOCAL_RANK / WORLD_SIZE itself (here with --rehearse, two ranks on the one GPU).  A short line, the same
    checks: both ranks joined, the frames are the single-rank ones."""
    args = ["--no-10m", "--no-4k", "--no-cpu", "--rays", "262144", "--tris", "50000", "--steps", "1", "--warmup", "1"]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29533",
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rehearse"] + args
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-4000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    two = json.loads(lines[0])
    assert two["n_gpus"] == 2 and two["rccl_ranks"] == 2 and two["collective_backend"] == "gloo"
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=600)
    assert one.returncode == 0, one.stderr[-4000:]
    one = json.loads([l for l in one.stdout.splitlines() if l.startswith("{")][0])
    for r1, r2 in zip(one["renders"], two["renders"]):
        assert r1["leg"] == r2["leg"] and r1["mean_radiance"] == r2["mean_radiance"] and r1["traced_rays"] == r2["traced_rays"]
    # a wrong --gpus is refused rather than silently run with another world size
    bad = subprocess.run(cmd[:cmd.index("--gpus") + 1] + ["3", "--rehearse"] + args, capture_output=True, text=True, timeout=600)
    assert bad.returncode != 0
