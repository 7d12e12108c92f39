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


def test_bench_dry_run_with_eight_ranks():
    """The line of the node configs[3] / SCALE_rNN names: `--gpus 8` self-launches eight ranks (gloo here), n_gpus == 8, the
    global batch and the rccl object carry the world size, and `warmup` / `warmup_requested` explain the driver's warm-up."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["ELO_BENCH_BACKEND"] = "gloo"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--dry-run", "--steps", "20",
                          "--warmup", "5", "--batch", "8"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["steps"] == 20 and d["config"]["parallelism"] == "dp8" and d["config"]["global_batch"] == 64
    assert d["warmup_requested"] == 5 and d["warmup"] >= 5 and d["scaling"] == "weak" and d["value"] is None
    r = d["rccl"]
    assert r["world_size"] == 8 and r["backend"] == "gloo" and r["all_reduce_bytes"] == 899134 * 4
    assert r["all_gather_bytes"] == 8 * 20 * 8 * 7 * 4                 # every rank's 20 steps x 8 pairs x 7 floats
    assert set(r) >= {"world_size", "backend", "all_reduce_us", "all_reduce_bytes", "all_gather_us", "all_gather_bytes", "numa"}


def test_bench_single_rank_dry_run_needs_no_launcher():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-run", "--steps", "3", "--warmup", "1"],
                         cwd=ROOT, env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    assert d["n_gpus"] == 1 and "launching" not in out.stderr


def test_bench_one_rank_process_group_rehearsal():
    """ELO_BENCH_FORCE_DIST=1: a ONE-rank process group (the 1-GPU box's rehearsal of the distributed path through RCCL,
    tests/test_bench_gpu.py) -- here over gloo and --dry-run: the group comes up without a launcher and without MASTER_* in the
    environment, the collectives run on the world of one, the line carries the `rccl` object, n_gpus stays 1."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(ELO_BENCH_FORCE_DIST="1", ELO_BENCH_BACKEND="gloo")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-run", "--steps", "4", "--warmup", "1"],
                         cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    assert d["n_gpus"] == 1 and d["config"]["parallelism"] == "dp1" and "launching" not in out.stderr
    r = d["rccl"]
    assert r["world_size"] == 1 and r["backend"] == "gloo" and r["all_reduce_us"] > 0 and r["all_gather_bytes"] == 4 * 7 * 4


RING_LOG = """\
node0:4242:4242 [0] NCCL INFO cudaDriverVersion 12000
node0:4242:4242 [0] NCCL INFO RCCL version 2.22.3+hip7.0 HEAD:abcdef0
node0:4242:4300 [0] NCCL INFO comm 0x55d0c8a0 rank 0 nranks 8 cudaDev 0 busId c000 commId 0x9d1c2f3e4a5b6c7d - Init START
node0:4242:4300 [0] NCCL INFO Channel 00/16 :    0   1   2   3   4   5   6   7
node0:4242:4300 [0] NCCL INFO Channel 01/16 :    0   2   4   6   7   5   3   1
node0:4242:4300 [0] NCCL INFO Trees [0] 1/-1/-1->0->-1 [1] 2/-1/-1->0->-1
node0:4242:4300 [0] NCCL INFO Channel 00 : 0[c000] -> 1[d000] via P2P/IPC
node0:4242:4300 [0] NCCL INFO Channel 01 : 0[c000] -> 2[e000] via P2P/IPC
node0:4242:4300 [0] NCCL INFO Connected all rings
node0:4242:4300 [0] NCCL INFO Connected all trees
node0:4242:4300 [0] NCCL INFO 16 coll channels, 0 collnet channels, 0 nvls channels, 16 p2p channels, 2 p2p channels per peer
node0:4242:4300 [0] NCCL INFO comm 0x55d0c8a0 rank 0 nranks 8 cudaDev 0 busId c000 commId 0x9d1c2f3e4a5b6c7d - Init COMPLETE
node0:4242:4242 [0] NCCL INFO AllReduce: 3596536 Bytes -> Algo 1 proto 2 time 61.250000
node0:4242:4242 [0] NCCL INFO AllGather: 224 Bytes -> Algo 1 proto 0 time 9.500000
"""

TREE_LOG = """\
gpu-box:77:77 [3] NCCL INFO comm 0x1 rank 3 nranks 4 cudaDev 3 busId 2f000 commId 0xfeed - Init START
gpu-box:77:91 [3] NCCL INFO Trees [0] -1/-1/-1->3->2 [1] -1/-1/-1->3->2
gpu-box:77:91 [3] NCCL INFO Channel 00/0 : 3[2f000] -> 2[2e000] via SHM/direct/direct
gpu-box:77:91 [3] NCCL INFO 8 coll channels, 0 collnet channels, 0 nvls channels, 8 p2p channels, 2 p2p channels per peer
gpu-box:77:91 [3] NCCL INFO Connected all trees
gpu-box:77:77 [3] NCCL INFO AllReduce: 3596536 Bytes -> Algorithm Tree Protocol LL128 time 40.1
"""


def test_parse_rccl_log_on_canned_excerpts(tmp_path):
    """bench.py's reading of an RCCL init / tuning log (NCCL_DEBUG=INFO, INIT,GRAPH,TUNING): rank count, channels, transports,
    ring / tree, algorithm / protocol -- on the ring spelling (numeric algo / proto) and the tree spelling (names)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    ring, tree = tmp_path / "ring.log", tmp_path / "tree.log"
    ring.write_text(RING_LOG)
    tree.write_text(TREE_LOG)
    r = bench.parse_rccl_log(str(ring))
    assert r["nranks"] == 8 and r["channels"] == 16 and r["transports"] == ["P2P/IPC"] and r["rings"] and r["trees"]
    assert r["algo_proto"] == ["AllGather:Ring/LL", "AllReduce:Ring/Simple"]
    q = bench.parse_rccl_log(str(tree))
    assert q["nranks"] == 4 and q["channels"] == 8 and q["transports"] == ["SHM/direct/direct"] and q["trees"]
    assert q["algo_proto"] == ["AllReduce:Tree/LL128"]
    missing = bench.parse_rccl_log(str(tmp_path / "absent.log"))
    assert missing["nranks"] is None and missing["algo_proto"] is None


def test_live_traffic_falls_back_with_a_reason(monkeypatch):
    """No GPU here: the counter pass cannot run, and bench._live_traffic must say why instead of raising (roofline_leg
    then reports the committed passes and puts the reason into traffic_source)."""
    sys.path.insert(0, ROOT)
    import bench
    monkeypatch.setattr(bench.shutil if hasattr(bench, "shutil") else __import__("shutil"), "which", lambda name: "/bin/false")
    got, why = bench._live_traffic(1, "f32", timeout=60)
    assert got is None and "FETCH_SIZE pass failed" in why
    import json as _json
    table = _json.load(open(bench.PMC_SUMMARY))
    assert bench._pmc_traffic("cv1_kernel", 1, "f32") == table["cv1_kernel/b1/f32"]["traffic_bytes"] > 0
