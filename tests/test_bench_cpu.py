"""CPU: `python bench.py --gpus 2` started WITHOUT torchrun launches two ranks itself and reports n_gpus == 2.
The hot path cannot run here (no GPU), so the ranks rehearse everything around it with --dry-run: rendezvous on
127.0.0.1, the gloo process group, the barriers, the pose all-gather and rank 0's single JSON line."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_launches_its_own_ranks():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["ELO_BENCH_BACKEND"] = "gloo"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run", "--steps", "5",
                          "--warmup", "2"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 5 and d["config"]["parallelism"] == "dp2" and d["config"]["global_batch"] == 2
    assert d["value"] is None and d["data"].startswith("dry-run")           # no throughput claim without a GPU
    assert "2 ranks" in out.stderr and "launching 2 ranks" in out.stderr
    # N > 1: the collectives of the path under their own clock + what the backend reported, in the line the driver parses
    r = d["rccl"]
    assert r["world_size"] == 2 and r["backend"] == "gloo" and r["all_reduce_us"] > 0 and r["all_gather_us"] > 0
    assert r["all_reduce_bytes"] == 899134 * 4 and "numa" in r


def test_bench_single_rank_dry_run_needs_no_launcher():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-run", "--steps", "3", "--warmup", "1"],
                         cwd=ROOT, env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    assert d["n_gpus"] == 1 and "launching" not in out.stderr
