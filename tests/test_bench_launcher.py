"""`python bench.py --gpus N` from a plain shell starts its own N ranks (no torch.distributed.run needed), relays
rank 0's single line and fails when a rank fails.  Runs the launcher's plumbing on CPU with gloo."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env["MASTER_ADDR"] = "127.0.0.1"
    return env


def test_plain_invocation_starts_n_ranks_and_relays_one_line():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--launch-check"], capture_output=True, text=True, env=_env(), timeout=240)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["ranks_joined"] == 2 and out["rank_sum"] == 3
    assert sorted(x["rank"] for x in out["ranks"]) == [0, 1]
    assert len({x["pid"] for x in out["ranks"]}) == 2          # two fresh processes
    assert [x["local_rank"] for x in sorted(out["ranks"], key=lambda x: x["rank"])] == [0, 1]


def test_failing_rank_fails_the_job():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--launch-check", "--launch-check-fail-rank", "1"], capture_output=True, text=True, env=_env(), timeout=240)
    assert r.returncode != 0
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_world_size_must_match_gpus():
    """Under an external launcher (torch.distributed.run sets WORLD_SIZE) a --gpus that disagrees is an error, not a
    silently smaller job."""
    env = dict(_env(), RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_PORT="29533")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--launch-check"], capture_output=True, text=True, env=env, timeout=240)
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)
