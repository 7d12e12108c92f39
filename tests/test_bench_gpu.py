"""GPU: bench.py keeps its output contract -- ONE JSON line with the driver's keys, the roofline and cpu_baseline
objects, finite positive numbers -- on a short run; `--gpus 2` started plainly becomes two ranks (sharing the one GPU
of the test box over gloo: a rehearsal of the launch path, not a scaling measurement)."""
import json
import math
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _line(out):
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]      # (gloo announces its connections on stdout)
    assert len(lines) == 1, out.stdout
    return json.loads(lines[0])


def test_bench_line_contract():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "40", "--warmup", "8", "--cpu-pairs", "2",
                          "--train-steps", "2"], cwd=ROOT, capture_output=True, text=True, timeout=900)
    d = _line(out)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 40 and d["warmup"] == 16 and d["higher_is_better"] is True   # warm-up >= 2 per lane
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    assert d["value"] > 0 and math.isfinite(d["value"]) and abs(d["value"] * d["ms_per_step"] / 1e3 - 1.0) < 1e-3   # B = 1
    assert d["repeats"] >= 2 and d["repeats"] * 40 * d["ms_per_step"] / 1e3 > 0.15          # >= 0.25 s of timed signal
    r = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_source"):
        assert key in r, key
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and 0 < r["frac"] < 1 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert r["mfma"]["peak"] == 2500.0 and 0 < r["mfma"]["frac"] < 1                          # the dtype ISSUED: fp16
    for leg in ("cost_volume_b8_f32", "cost_volume_b8_f16", "per_operator_b8_f32", "per_operator_b64_f16", "per_operator_all_levels_b8_f16"):
        assert r[leg]["bound"] == "hbm" and 0 < r[leg]["frac"] < 1, leg
    assert set(r["cost_volume_b8_f16"]["levels"]) == {"l0", "l1", "l2", "l2_origin"}
    c = d["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample", "one_core", "cores_available", "one_process_all_threads"):
        assert key in c, key
    assert c["kind"] in ("reference", "port") and c["cores"] == os.cpu_count() and c["value"] > 0 and c["one_core"] > 0
    assert d["from_raw_clouds"]["value"] > 0
    b8 = d["batch8"]
    assert b8["f32"] > 0 and b8["f16_features"] > 0 and b8["f16_features_f16_products"] > 0
    assert d["dense_f32"]["value"] and d["dense_f32"]["value"] > 0
    h = d["hires"]                                        # BASELINE configs[4]: 128x2048
    assert h["grid"] == "128x2048" and h["batch1"] > 0 and h["batch8_f32"] > 0 and h["batch8_f16_features"] > 0
    assert h["cost_volume_b8_f16"]["us"] > 0 and all(v >= 1 for v in h["workgroups_per_cu_by_lds"].values())
    assert d["roofline"]["carries_riders_in_the_forward"] in (True, False)
    # round 6: the HBM-cold readings beside the warm ones, the KITTI-density legs, the cost of submit()'s default ordering
    for leg in ("per_operator_hires_b8_f32", "per_operator_hires_b8_f16", "per_operator_hires_all_levels_b8_f32", "cost_volume_b8_f16"):
        assert 0 < r[leg]["frac_cold"] <= 0.85 and 0 < r[leg]["frac_warm"] < 1 and r[leg]["frac"] == r[leg]["frac_cold"], leg
    for term in r["per_operator_hires_b8_f32"]["terms"].values():
        assert term["between_uses_bytes"] >= 2 * 256 * 1024 * 1024 and term["ring"] >= 3 and 0 < term["frac_cold"] <= 0.85
    assert 0 < r["frac_cold"] < 1
    sp = d["sparse"]
    assert sp["batch1"] > 0 and sp["batch8_f16_features"] > 0 and sp["from_raw_clouds_60k_valid"] > 0 and set(sp["ratio_to_dense"]) == {"batch1", "batch8_f16_features", "from_raw_clouds"}
    so = d["submit_ordering"]
    assert so["default_ordered"] > 0.9 * so["caller_owned_ready_False"]
    t = d["train_dp"]
    assert t["n_gpus"] == 1 and t["value"] > 0 and t["batch_per_gpu"] == 8


def test_bench_gpus_2_launches_two_ranks():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["ELO_BENCH_BACKEND"] = "gloo"                    # two ranks on ONE GPU: RCCL would refuse the duplicate device
    env["ELO_BENCH_SHARE_GPU"] = "1"                     # (without it bench.py refuses --gpus 2 on a 1-GPU box)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "8",
                          "--lanes", "4", "--train-steps", "2"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    d = _line(out)
    assert d["n_gpus"] == 2 and d["config"]["parallelism"] == "dp2" and d["value"] > 0
    assert d["train_dp"]["n_gpus"] == 2 and d["train_dp"]["value"] > 0 and "gloo" in d["train_dp"]["collective"]
    assert "roofline" not in d                            # the N = 1 legs stay with N = 1
    assert d["rccl"]["world_size"] == 2 and d["rccl"]["all_reduce_us"] > 0


def test_bench_runs_its_distributed_path_through_rccl_as_one_rank(tmp_path):
    """No multi-GPU node is available to the builder, so RCCL had never been LOADED by this code.  On the 1-GPU box:
    ELO_BENCH_FORCE_DIST=1 makes bench.py a ONE-rank `nccl` group -- init_process_group("nccl", device_id=...), the barriers, the pose
    all-gather inside the timed region, train_dp's flat-bucket all-reduce (899 134 floats, between the two captured graphs), rccl_leg's
    clocks, and parse_rccl_log on the LIVE NCCL_DEBUG_FILE of this run.  A 1-rank communicator has no peers, so there is no ring /
    transport line to find; what must be there: the communicator's rank count, and the log itself."""
    log = tmp_path / "rccl.log"
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(ELO_BENCH_FORCE_DIST="1", NCCL_DEBUG="INFO", NCCL_DEBUG_SUBSYS="INIT,GRAPH,TUNING,COLL", NCCL_DEBUG_FILE=str(log),
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "8", "--train-steps", "2"],
                         cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    d = _line(out)
    assert d["n_gpus"] == 1 and d["value"] > 0
    r = d["rccl"]
    assert r["backend"] == "nccl" and r["world_size"] == 1 and r["all_reduce_us"] > 0 and r["all_gather_us"] > 0
    assert r["all_reduce_bytes"] == 899134 * 4
    assert r["nranks"] == 1, r                            # parsed from the live log ("... nranks 1 ... Init COMPLETE")
    text = log.read_text(errors="replace")
    assert "Init COMPLETE" in text or "Init START" in text, text[-2000:]
    assert "nccl" in d["train_dp"]["collective"]
    keep = os.path.join(ROOT, "gpurun_out", "r06")
    os.makedirs(keep, exist_ok=True)
    with open(os.path.join(keep, "rccl_one_rank.log"), "w") as f:      # evidence for profiles/: the head of a real RCCL log
        f.write(text[:20000])
    with open(os.path.join(keep, "rccl_one_rank.json"), "w") as f:
        json.dump({"rccl": r, "train_dp": d["train_dp"]}, f, indent=1)
