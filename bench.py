#!/usr/bin/env python
"""bench.py -- frame-pairs/s of the EfficientLO-Net hot path on MI355X.

    python bench.py [--gpus N --steps K --warmup W] [--batch B --height H --width W]

Workload (BASELINE.json configs[1]): the full 4-level PWC pyramid -- Siamese set-conv
pyramid, initial attentive cost volume, coarse pose, three warp-refinement levels
(warp + re-projection + cost volume + 2x set-upconv + pose head) -- on synthetic
KITTI-shaped 64x1800 range-image pairs, batch 1, fp32, random-init weights.
A "step" is one forward pass over one batch of pairs already resident in HBM
(a pool of pre-generated pairs; each step copies the next pair into the captured
graph's input buffers, device-to-device, inside the timed region).

One process per GPU.  Under torchrun (RANK / LOCAL_RANK / WORLD_SIZE in the environment) this process is one rank;
started plainly with --gpus N > 1 it re-launches itself as N ranks (torch.distributed.run, 127.0.0.1 rendezvous) and
relays rank 0's line.  Frame pairs are independent, so ranks shard the stream with no data-path collective; every
step's pose is logged on the device and the logs are all-gathered (7 floats per pair) once per timed repeat, inside
the timed region, so rank 0 can chain them (main.py:557-572): the only exchange inference has.  scaling = weak.

Timing: W untimed warm-up steps (at least 2 per lane), then repeats of EXACTLY K steps, each bracketed by a barrier +
torch.cuda.synchronize() on both sides and reduced with MAX over ranks; repeats continue until 0.25 s have been
measured (a 20-step run at batch 1 is 2.5 ms of signal otherwise) and `ms_per_step` / `value` come from the MEDIAN
repeat (`repeats`, `ms_per_step_repeats` show them all).

Throughput design: the whole forward is one hipGraph; `--lanes` graphs (default 8), each with its own static
buffers, dealt over one stream per hardware queue (model.distinct_queue_streams), keep independent forwards
in flight (step i rides lane i % lanes).

Besides the contract keys the line carries (rank 0):
  roofline     -- SURVEY.md section 8(d): the dominant kernel of the timed path (fused cost-volume stage 1 at l0) timed
                  live with HIP events, its algorithmic bytes against the 8 TB/s HBM roofline and its executed
                  matrix-core flops against the peak of the dtype it issues; plus `cost_volume_*`: ALL cost-volume
                  kernels of a forward (four levels) at batch 8, fp32 and fp16 feature storage (configs[2])
  cpu_baseline -- the CPU oracle (oracle/, kind "port") on one core and on all host cores
  batch8, from_raw_clouds, dense_f32, train_dp -- see the functions below
"""
import argparse
import importlib
import json
import os
import socket
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s peak
F16_MFMA_PEAK_TFLOPS = 2500.0  # ibid.: BF16/FP16 MFMA ~2.5 PFLOP/s dense -- the dtype the split products are ISSUED in
F32_MFMA_PEAK_TFLOPS = 157.3   # ibid.: v_mfma_f32_16x16x4_f32 dense peak -- only for the -DELO_DENSE_F32 comparison build
MFMA_PRODUCTS = 3              # fp16 matrix-core products per fp32-class product: hi*hi + hi*lo + lo*hi (elo_fused.hip)
MIN_TIMED_S = 0.25
PMC_SUMMARY = os.path.join(ROOT, "profiles", "r04_pmc", "summary.json")     # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes
PMC_HIRES_SUMMARY = os.path.join(ROOT, "profiles", "r05_pmc_hires", "summary.json")   # the per-operator kernels at the 128 x 2048 l0 shape


def pkg(sub=None):
    return importlib.import_module("efficientlo-net_amd" + ("." + sub if sub else ""))


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--batch", type=int, default=1, help="frame pairs per step per GPU (configs[1]: 1)")
    ap.add_argument("--height", type=int, default=64)
    ap.add_argument("--width", type=int, default=1800)
    ap.add_argument("--pool", type=int, default=7,
                    help="distinct synthetic pairs kept in HBM (coprime to --lanes, so a lane sees a different pair on every replay)")
    ap.add_argument("--inputs", choices=("copied", "in-place"), default="copied",
                    help="copied: every step copies a pooled pair into its lane's input buffer (a launch in the lane's chain); "
                         "in-place: the lanes' buffers are filled once, before the timed region (a producer writing in place)")
    ap.add_argument("--lanes", type=int, default=8,
                    help="independent forwards in flight (hipGraphs, dealt over the 4 hardware queues: use a multiple of 4)")
    ap.add_argument("--products", choices=("split", "half"), default="split",
                    help="dense products of the fused kernels: split = fp32-class (three fp16 MFMA products, the headline); "
                         "half = ONE fp16 product (fp16 arithmetic; not the fp32 parity path)")
    ap.add_argument("--features", choices=("f32", "f16"), default="f32",
                    help="storage of the feature tensors in HBM (f16: BASELINE configs[2]); arithmetic is unchanged")
    ap.add_argument("--fresh-orders", type=int, default=16,
                    help="pre-drawn window visiting orders per lane: every replay walks the next set (tf.random_shuffle per "
                         "sess.run, utils/pointnet_util.py:45,104,193,270: one tiny launch at the head of each replay); 0 = one "
                         "fixed draw for the life of the graph")
    ap.add_argument("--submit-threads", type=int, default=1,
                    help="host threads that enqueue the steps of a repeat (step i goes to thread i %% N).  Measured at the driver's "
                         "--steps 20: 1 thread 9.65 k pairs/s, 2 threads 9.4-9.8 k, 4 threads 9.3 k, 8 threads 8.7 k -- the "
                         "interpreter lock costs more than the staggered start of the four queues; default 1")
    ap.add_argument("--submit-order", choices=("caller", "stream"), default="caller",
                    help="caller: submit(..., ready=False) -- the pooled pairs are resident and synchronised before the timed region, the bench "
                         "owns the ordering (the unordered fast path); stream: submit()'s default -- every step first orders the lane's stream "
                         "behind the caller's current stream (one hipEventRecord + hipStreamWaitEvent inside the native submit) and "
                         "record_stream()s the pair")
    ap.add_argument("--check-every", type=int, default=64,
                    help="every N-th replay of a lane runs the graph recorded on the range-checked kernels (the fp16 split's "
                         "production guard, PWCLONet.capture(check_every=N)); 0: off")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of hipGraph replay")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-legs", action="store_true",
                    help="only the timed loop: no roofline / cpu_baseline / from_raw_clouds / batch8 / dense_f32 / train_dp legs")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="roofline.traffic from the committed counter passes instead of two rocprofv3 --pmc child runs (~30 s)")
    ap.add_argument("--cpu-pairs", type=int, default=24, help="pairs in the one-core CPU sample (all cores: 3 per core)")
    ap.add_argument("--train-steps", type=int, default=8, help="timed steps of the train_dp leg (0: skip)")
    ap.add_argument("--dry-run", action="store_true",
                    help="rehearse the multi-process path WITHOUT a GPU: ranks, barriers, the pose all-gather and the line, "
                         "with an empty step (value is null, data says so)")
    return ap.parse_args()


# ----------------------------------------------------------------------------- self-launch of N ranks
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(n):
    """`python bench.py --gpus N` without a torchrun environment: become N ranks on this node."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # dmabuf IPC: RCCL across processes needs it on this driver
    print("[bench] launching %d ranks: %s" % (n, " ".join(cmd[1:])), file=sys.stderr, flush=True)
    return subprocess.call(cmd, env=env)


# ----------------------------------------------------------------------------- SURVEY 8(d) byte / flop counts
def cost_volume_bytes(N, C, Kq, Kp, s=4):
    """algorithmic (operator-boundary) bytes of the cost volume per batch element."""
    A1 = 2 * N * 12 + 2 * N * C * s + N * Kq * 16 + N * Kq * (10 + 2 * C) * s
    P1 = 2 * N * Kq * 64 * s + N * Kq * 4 + N * 64 * s
    A2 = N * 12 + N * C * s + N * 64 * s + N * Kp * 16 + N * Kp * (10 + C + 64) * s
    P2 = 2 * N * Kp * 64 * s + N * Kp * 4 + N * 64 * s
    return dict(A1=A1, P1=P1, A2=A2, P2=P2)


def cv1_flops(N, C, Kq):
    """Multiply-adds x2 of the six 1x1 convolutions of cost-volume stage 1 (pointnet_util.py:72-90)."""
    per_row = (10 + 2 * C) * 128 + 128 * 64 + 64 * 64 + 10 * 64 + 128 * 128 + 128 * 64
    return 2 * N * Kq * per_row


def cv2_flops(N, C, Kp):
    """... of the three convolutions of stage 2 (pointnet_util.py:123-135)."""
    return 2 * N * Kp * (10 * 64 + (128 + C) * 128 + 128 * 64)


def _time_launches(fn, dev, reps):
    """Average duration of one launch of `fn`: `reps` back-to-back launches captured in a hipGraph (so the host's
    per-call cost, ~12 us from Python, is not in the number), replayed between two HIP events on torch's current
    stream == the stream _lib.stream_ptr() hands to the C ABI."""
    import torch
    for _ in range(3):
        fn()
    torch.cuda.synchronize(dev)
    graph = torch.cuda.CUDAGraph()
    with pkg("model").graph_capture(graph):
        for _ in range(reps):
            fn()
    graph.replay()
    torch.cuda.synchronize(dev)
    start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    start.record()
    graph.replay()
    stop.record()
    torch.cuda.synchronize(dev)
    sec = start.elapsed_time(stop) / 1e3 / reps
    del graph
    return sec


LLC_BYTES = 256 * 1024 * 1024     # MI355X_MICROARCH.md: 256 MB Infinity Cache in front of HBM (memory side, all XCDs)


def _tensors(tree):
    """The torch tensors of a (nested tuple / list / dict) argument tree."""
    import torch
    if torch.is_tensor(tree):
        return [tree]
    if isinstance(tree, dict):
        tree = list(tree.values())
    if isinstance(tree, (list, tuple)):
        return [t for v in tree for t in _tensors(v)]
    return []


def _clone_tensors(tree):
    """The same tree with every tensor cloned (new addresses, same contents); anything else -- packed weights, grouping
    descriptions, scalars -- is shared: weights are the small hot operand in the product too."""
    import torch
    if torch.is_tensor(tree):
        return tree.clone()
    if isinstance(tree, dict):
        return {k: _clone_tensors(v) for k, v in tree.items()}
    if isinstance(tree, (list, tuple)):
        return type(tree)(_clone_tensors(v) for v in tree)
    return tree


def _footprint(args, kwargs, out):
    seen, total = set(), 0
    for t in _tensors((args, kwargs, out)):
        key = t.untyped_storage().data_ptr()
        if key not in seen:
            seen.add(key)
            total += t.untyped_storage().nbytes()
    return total


def _time_ring(call, args, kwargs, dev, between=2 * LLC_BYTES, passes=3, max_ring=768):
    """COLD duration of one launch of call(*args, **kwargs): the launch is issued over a RING of R distinct tensor sets
    (tensor arguments cloned, outputs kept alive through the capture so that every launch writes its own), R chosen so
    that >= `between` bytes (default 2 x the 256 MB Infinity Cache) of OTHER sets' inputs and outputs pass between two
    uses of the same set -- by the time a set comes round again neither L2 nor the Infinity Cache can still hold it, so
    every launch reads its inputs from HBM and its writes cannot be absorbed by a later overwrite.  One pass over the
    ring is captured in a hipGraph; `passes` replays between two HIP events on the launch stream.
    -> (seconds per launch, {"footprint_bytes", "ring", "between_uses_bytes"})."""
    import torch
    kwargs = kwargs or {}
    foot = _footprint(args, kwargs, call(*args, **kwargs))
    ring = min(max_ring, max(3, -(-between // max(foot, 1)) + 1))
    sets = [(args, kwargs)] + [(_clone_tensors(args), _clone_tensors(kwargs)) for _ in range(ring - 1)]
    for a, k in sets:
        call(*a, **k)
    torch.cuda.synchronize(dev)
    graph = torch.cuda.CUDAGraph()
    with pkg("model").graph_capture(graph):
        keep = [call(*a, **k) for a, k in sets]
    graph.replay()
    torch.cuda.synchronize(dev)
    start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    start.record()
    for _ in range(passes):
        graph.replay()
    stop.record()
    torch.cuda.synchronize(dev)
    sec = start.elapsed_time(stop) / 1e3 / (ring * passes)
    del graph, keep, sets
    return sec, {"footprint_bytes": int(foot), "ring": ring, "between_uses_bytes": int(foot * (ring - 1))}


def _pmc_commit():
    """' @ <commit>' the committed counter passes were taken at (written into the summary when it was copied to profiles/)."""
    try:
        return " @ commit " + json.load(open(PMC_SUMMARY))["captured_at_commit"]
    except (OSError, KeyError, ValueError):
        return ""


def _pmc_traffic(kernel, batch, features="f32"):
    """HBM bytes per launch of `kernel` from the committed counter passes (profiles/r02_pmc/summary.json, collected by
    tools/pmc_collect.sh: FETCH_SIZE and WRITE_SIZE in separate rocprofv3 --pmc runs, FETCH_SIZE x2 per the gfx950
    note of MI355X_MICROARCH.md).  Counters cannot be read from inside this process: null when the shape was not captured."""
    try:
        table = json.load(open(PMC_SUMMARY))
    except (OSError, ValueError):
        return None
    hit = table.get("%s/b%d/%s" % (kernel, batch, features))
    return None if hit is None else hit.get("traffic_bytes")


def _live_traffic(batch, features, timeout=150):
    """HBM bytes per launch of the fused cost-volume stage 1 at l0 MEASURED DURING THIS RUN: two child processes, each
    `rocprofv3 --pmc <counter>` (FETCH_SIZE, then WRITE_SIZE: separate passes, no trace domain beside them) around
    tools/roofline_micro.py --kernel cv1_recorded (the launch roofline_leg times: stage 1 at l0 of a real forward on its own
    tensors), averaged over the last 20 of its 25 launches; FETCH_SIZE x2 (the gfx950 correction of MI355X_MICROARCH.md's HBM section), both in KB.
    (bytes, note) -- (None, why) when rocprofv3 is missing, a pass fails or times out: the caller then falls back to the
    committed passes.  The parent is idle meanwhile (called between the timed region and the other legs)."""
    import csv, glob, shutil, subprocess, tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    micro = os.path.join(ROOT, "tools", "roofline_micro.py")
    cmd = [sys.executable, micro, "--kernel", "cv1_recorded", "--batch", str(batch), "--reps", "25"] + (["--half"] if features == "f16" else [])
    got = {}
    work = tempfile.mkdtemp(prefix="elo_pmc_", dir="/tmp")
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(work, counter)
            env = dict(os.environ, TMPDIR="/tmp")
            try:
                subprocess.run([exe, "--pmc", counter, "--output-format", "csv", "-d", out, "-o", "c", "--"] + cmd, cwd="/tmp", env=env,
                               stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=timeout, check=True)
            except (subprocess.SubprocessError, OSError) as e:
                return None, "%s pass failed: %s" % (counter, type(e).__name__)
            # the stage's launches: the tile kernel (in-kernel grouping), or the select-k pre-pass + the register-resident kernel
            vals = {"cv1_kernel": [], "cv1_rr_kernel": [], "group_select_k": []}
            for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
                for r in csv.DictReader(open(f)):
                    if r.get("Counter_Name") != counter:
                        continue
                    for key in vals:
                        if key in r.get("Kernel_Name", ""):
                            vals[key].append(float(r["Counter_Value"]))
            chain = len(vals["cv1_rr_kernel"]) >= 20
            groups = [vals["cv1_rr_kernel"], vals["group_select_k"]] if chain else [vals["cv1_kernel"]]
            if any(len(g) < 20 for g in groups):
                return None, "%s pass: no cv1 launches in the counter file" % counter
            got[counter] = sum(sum(g[-20:]) / 20 for g in groups)   # the last 20 of the 25 repeats (the recording forwards come first)
    finally:
        shutil.rmtree(work, ignore_errors=True)
    return int((2 * got["FETCH_SIZE"] + got["WRITE_SIZE"]) * 1024), (
        "measured in this run: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) around tools/roofline_micro.py "
        "--kernel cv1_recorded --batch %d%s; 2 x FETCH_SIZE + WRITE_SIZE, KB (FETCH_SIZE %.0f, WRITE_SIZE %.0f per launch)"
        % (batch, " --half" if features == "f16" else "", got["FETCH_SIZE"], got["WRITE_SIZE"]))


def _hires_traffic(features):
    """HBM bytes per launch (2 x FETCH_SIZE + WRITE_SIZE, KB) of the four per-operator kernels at the 128 x 2048 l0 shape, batch
    8, from the committed counter passes (profiles/r05_pmc_hires/summary.json: tools/pmc_collect.sh with ELO_PMC_GRID=32x256);
    None per term when that pass is not in the file."""
    try:
        table = json.load(open(PMC_HIRES_SUMMARY))
    except (OSError, ValueError):
        return None
    out = {}
    for term, kernel, tag in (("A1", "cv_encode1", "encode1"), ("P1", "softmax_pool", "pool"), ("A2", "cv_encode2", "encode2"),
                              ("P2", "softmax_pool", "pool2")):
        row = next((r for r in table.get("rows", []) if r["tag"] == "%s%s_b8" % (tag, "_f16" if features == "f16" else "")), None)
        out[term] = None if row is None else {"traffic_bytes": int(row["hbm_MB_fetch_x2"] * 1e6), "algorithmic_bytes": int(row["algorithmic_MB"] * 1e6),
                                              "trace_us": row["avg_us"]}
    out["source"] = os.path.relpath(PMC_HIRES_SUMMARY, ROOT)
    return out


LEVELS = ("l2_origin", "l2", "l1", "l0")          # the order in which a forward issues its four cost volumes


def recorded_cost_volume(dev, B, H_in, W_in, half, seed=5):
    """The eight cost-volume launches of ONE REAL FORWARD at batch B (pwclo_model.py:170, :242, :316, :390), on the tensors
    that forward fed them: an eager forward of a fresh net (random-init weights, seed 0: the timed workload's) on a
    synthetic pair is recorded (fused.recording clones every argument), and each level's stage 1 / stage 2 call can be
    re-issued as often as a timing needs.  {level: dict(run1, run2, N, C, Kq, Kp, riders)}; `riders`: stage 1 carried
    set-upconv jobs in the forward (batch 1-2) -- run1 re-issues the cost volume alone."""
    import torch
    model, synth, fused = pkg("model"), pkg("synth"), pkg("fused")
    net = model.PWCLONet(dev, seed=0, feature_dtype=torch.float16 if half else torch.float32)
    f1, f2 = synth.frame_pair(B, H_in, W_in, seed=seed)
    both = torch.cat([torch.from_numpy(f1), torch.from_numpy(f2)], 0).to(dev)
    net.forward(both[:B], both[B:])                      # variables, packed weights, caches
    with fused.recording() as calls:
        net.forward(both[:B], both[B:])
    torch.cuda.synchronize(dev)
    s1 = [c for c in calls if c[0] == 1]
    s2 = [c for c in calls if c[0] == 2]
    assert len(s1) == 4 and len(s2) == 4, (len(s1), len(s2))
    out = {}
    for name, c1, c2 in zip(LEVELS, s1, s2):
        xyz1, feat1 = c1[1][0], c1[1][1]
        out[name] = dict(run1=(lambda c=c1: fused.cv_stage1(*c[1], **c[2])), run2=(lambda c=c2: fused.cv_stage2(*c[1], **c[2])),
                         call1=(fused.cv_stage1, c1[1], c1[2]), call2=(fused.cv_stage2, c2[1], c2[2]),
                         N=xyz1.shape[1], C=feat1.shape[-1], Kq=c1[2]["K"], Kp=c2[2]["K"], riders=c1[3])
    return out


def cost_volume_leg(dev, B, H_in, W_in, half, reps=20, cold=True):
    """SURVEY 8(d)'s cost-volume figure ON THE PATH THE VALUE RUNS: the eight fused launches (stage 1 + stage 2 at
    l2_origin, l2, l1, l0; from 24.6 k rows on a stage is a grouping pre-pass + the register-resident kernel, timed
    together) of one forward at batch B ON THAT FORWARD'S OWN TENSORS (recorded_cost_volume), each timed with HIP events;
    achieved = the operator-boundary bytes (A1+P1 / A2+P2, s = 4 or 2) of all of them x B / the sum of the times,
    against 8 TB/s.  The fused kernels never materialise those tensors (their HBM traffic is the compulsory bytes only),
    so this is the section-8(d) accounting figure, not a bandwidth they could reach: what bounds them is the matrix +
    vector work, `mfma` (executed fp16 products against the 2.5 PFLOP/s fp16 peak)."""
    s = 2 if half else 4
    rec = recorded_cost_volume(dev, B, H_in, W_in, half)
    levels, tot_b, tot_s, tot_w, tot_f = {}, 0, 0.0, 0.0, 0
    for lv in ("l0", "l1", "l2", "l2_origin"):
        L = rec[lv]
        w1, w2 = _time_launches(L["run1"], dev, reps), _time_launches(L["run2"], dev, reps)
        (t1, r1), (t2, r2) = (_time_ring(*L["call1"], dev), _time_ring(*L["call2"], dev)) if cold else ((w1, {}), (w2, {}))
        cb = cost_volume_bytes(L["N"], L["C"], L["Kq"], L["Kp"], s)
        b1, b2 = (cb["A1"] + cb["P1"]) * B, (cb["A2"] + cb["P2"]) * B
        fl = (cv1_flops(L["N"], L["C"], L["Kq"]) + cv2_flops(L["N"], L["C"], L["Kp"])) * B
        levels[lv] = {"cv1_us": round(t1 * 1e6, 2), "cv2_us": round(t2 * 1e6, 2), "cv1_us_warm": round(w1 * 1e6, 2), "cv2_us_warm": round(w2 * 1e6, 2),
                      "bytes": int(b1 + b2), "GBps": round((b1 + b2) / (t1 + t2) / 1e9, 1),
                      "frac": round((b1 + b2) / (t1 + t2) / 1e9 / HBM_PEAK_GBS, 4), "frac_warm": round((b1 + b2) / (w1 + w2) / 1e9 / HBM_PEAK_GBS, 4)}
        if cold:
            levels[lv]["footprint_bytes"] = [r1["footprint_bytes"], r2["footprint_bytes"]]
            levels[lv]["between_uses_bytes"] = min(r1["between_uses_bytes"], r2["between_uses_bytes"])
        tot_b, tot_s, tot_w, tot_f = tot_b + b1 + b2, tot_s + t1 + t2, tot_w + w1 + w2, tot_f + fl
    gbs = tot_b / tot_s / 1e9
    tfs = MFMA_PRODUCTS * tot_f / tot_s / 1e12
    out = {"kernel": "cost-volume stage 1 + stage 2 at l0, l1, l2, l2_origin (the launches of one forward, on its own tensors)",
           "batch": B, "features": "f16" if half else "f32", "bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS,
           "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 5), "bytes": int(tot_b), "us": round(tot_s * 1e6, 2),
           "frac_warm": round(tot_b / tot_w / 1e9 / HBM_PEAK_GBS, 5), "us_warm": round(tot_w * 1e6, 2),
           "target_us_at_60pct": round(tot_b / (0.6 * HBM_PEAK_GBS * 1e9) * 1e6, 1),
           "mfma": {"executed_TFLOPs": round(tfs, 1), "peak": F16_MFMA_PEAK_TFLOPS,
                    "frac": round(tfs / F16_MFMA_PEAK_TFLOPS, 5), "issued_as": "v_mfma_f32_16x16x32_f16 x3 per 32-k pair (v_mfma_f32_16x16x16_f16 x3 on a 16-k tail)"},
           "levels": levels}
    if cold:
        out.update(frac_cold=out["frac"], us_cold=out["us"], reading=COLD_NOTE)
    return out


# the four cost volumes of a 64 x 1800 forward: grid, feature channels, stage-1 neighbours, stage-1 window (pwclo_model.py:170,
# :242, :316, :390)
PER_OPERATOR_LEVELS = {"l0": (16, 225, 16, 6, (11, 41)), "l1": (8, 113, 32, 6, (7, 25)), "l2": (4, 57, 64, 6, (5, 15)),
                       "l2_origin": (4, 57, 64, 32, (5, 35))}
# ... and of a 128 x 2048 forward (BASELINE configs[4]'s scans: levels 32x256 / 16x128 / 8x64; same channels, K and windows)
PER_OPERATOR_LEVELS_HIRES = {"l0": (32, 256, 16, 6, (11, 41)), "l1": (16, 128, 32, 6, (7, 25)), "l2": (8, 64, 64, 6, (5, 15)),
                             "l2_origin": (8, 64, 64, 32, (5, 35))}


def per_operator_all_levels_leg(dev, batch, half, table=None):
    """The 16 per-operator cost-volume launches of a forward (A1, P1, A2, P2 at l0, l1, l2, l2_origin) against the sum of
    their algorithmic bytes (216 MB at batch 8 with fp16 features), as separate launches.  `table`: the
    level shapes (default: the 64 x 1800 forward's; PER_OPERATOR_LEVELS_HIRES: the 128 x 2048 forward's)."""
    table = table or PER_OPERATOR_LEVELS
    levels = {lv: per_operator_leg(dev, batch, half, level=lv, table=table) for lv in table}
    tot_b, tot_us = sum(r["bytes"] for r in levels.values()), sum(r["us"] for r in levels.values())
    tot_w = sum(r["us_warm"] for r in levels.values())
    gbs = tot_b / tot_us / 1e3
    return {"kernel": "ELO_FUSED=0 cost volume, all four levels: 4 x (cv_encode1 + softmax_pool + cv_encode2 + softmax_pool)",
            "batch": batch, "features": "f16" if half else "f32", "bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 5), "bytes": int(tot_b), "us": round(tot_us, 2),
            "frac_cold": round(gbs / HBM_PEAK_GBS, 5), "us_cold": round(tot_us, 2),
            "frac_warm": round(tot_b / tot_w / 1e3 / HBM_PEAK_GBS, 5), "us_warm": round(tot_w, 2), "reading": COLD_NOTE,
            "target_us_at_60pct": round(tot_b / (0.6 * HBM_PEAK_GBS * 1e3), 1),
            "levels": {lv: {"us": r["us"], "us_warm": r["us_warm"], "bytes": r["bytes"], "frac": r["frac"], "frac_warm": r["frac_warm"],
                            "terms_frac_cold": {t: v["frac"] for t, v in r["terms"].items()},
                            "between_uses_bytes": min(v["between_uses_bytes"] for v in r["terms"].values())}
                       for lv, r in levels.items()}}


def per_operator_leg(dev, batch, half, reps=20, level="l0", table=None, cold=True):
    """The four HBM-bound kernels of the ELO_FUSED=0 cost volume at one level (default l0) (SURVEY 8(d) terms A1, P1, A2, P2:
    gather/encode and masked softmax pooling with the GEMMs between them left to hipBLASLt), each against its own
    algorithmic bytes.  `table`: the level shapes (PER_OPERATOR_LEVELS, or PER_OPERATOR_LEVELS_HIRES for configs[4]'s scans)."""
    import torch
    ops, synth, elo = pkg("_ops"), pkg("synth"), pkg()
    H, W, C, Kq, win = (table or PER_OPERATOR_LEVELS)[level]
    Kp = 4
    N = H * W
    g = torch.Generator(device="cpu").manual_seed(3)
    cvb = cost_volume_bytes(N, C, Kq, Kp, 2 if half else 4)
    cast = (lambda x: x.half()) if half else (lambda x: x)
    fb1, fb2 = synth.frame_pair(batch, H, W, seed=6)
    x1, x2 = torch.from_numpy(fb1).to(dev), torch.from_numpy(fb2).to(dev)
    ft1, ft2 = (cast(torch.randn((batch, H, W, C), generator=g).to(dev)) for _ in range(2))
    hw = torch.from_numpy(synth.hw_index(batch, H, W)).to(dev)
    order = torch.randperm(win[0] * win[1], generator=g).to(torch.int32).to(dev)
    idx_q, _, _, m_q = elo.fused_conv_select_k(x1, x2, hw, order, H, W, N, win[0], win[1], Kq, 0, 1000.0, 1, 1, want_valid=False)
    order_p = torch.randperm(15, generator=g).to(torch.int32).to(dev)
    idx_p, _, _, m_p = elo.fused_conv_random_k(x1, x1, hw, order_p, H, W, N, 3, 5, Kp, 0, 1000.0, 1, 1, want_valid=False)
    m_q, m_p = m_q.reshape(batch, N, Kq), m_p.reshape(batch, N, Kp)
    cost = cast(torch.randn((batch, H, W, 64), generator=g).to(dev))
    lq, vq = (cast(torch.randn((batch, N, Kq, 64), generator=g).to(dev)) for _ in range(2))
    lp, vp = (cast(torch.randn((batch, N, Kp, 64), generator=g).to(dev)) for _ in range(2))
    legs = {"A1": (ops.cv_encode1, (x1.reshape(batch, N, 3), ft1.reshape(batch, N, C), x2, ft2, idx_q, m_q)),
            "P1": (ops.masked_softmax_pool, (lq, vq, m_q)),
            "A2": (ops.cv_encode2, (x1, ft1, cost, idx_p, m_p)),
            "P2": (ops.masked_softmax_pool, (lp, vp, m_p))}
    terms, tot_b, tot_w, tot_c = {}, 0, 0.0, 0.0
    for term, (fn, a) in legs.items():
        warm = _time_launches(lambda: fn(*a), dev, reps)
        cold_s, ring = _time_ring(fn, a, None, dev) if cold else (warm, {})
        nbytes = cvb[term] * batch
        terms[term] = {"bytes": int(nbytes), "us": round(cold_s * 1e6, 2), "frac": round(nbytes / cold_s / 1e9 / HBM_PEAK_GBS, 4),
                       "us_warm": round(warm * 1e6, 2), "frac_warm": round(nbytes / warm / 1e9 / HBM_PEAK_GBS, 4)}
        if cold:
            terms[term].update(us_cold=terms[term]["us"], frac_cold=terms[term]["frac"], **ring)
        tot_b, tot_w, tot_c = tot_b + nbytes, tot_w + warm, tot_c + cold_s
    gbs, gbs_w = tot_b / tot_c / 1e9, tot_b / tot_w / 1e9
    out = {"kernel": "ELO_FUSED=0 cost volume at %s (%dx%d grid): cv_encode1 + softmax_pool + cv_encode2 + softmax_pool" % (level, H, W), "batch": batch,
           "features": "f16" if half else "f32", "bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "frac": round(gbs / HBM_PEAK_GBS, 5), "bytes": int(tot_b), "us": round(tot_c * 1e6, 2),
           "frac_warm": round(gbs_w / HBM_PEAK_GBS, 5), "us_warm": round(tot_w * 1e6, 2), "terms": terms}
    if cold:
        out.update(frac_cold=out["frac"], us_cold=out["us"], reading=COLD_NOTE)
    return out


COLD_NOTE = ("frac / us / frac_cold = COLD: every launch on its own tensor set of a ring with >= 2 x 256 MB (Infinity Cache) of other "
             "sets' traffic between two uses of a set -- inputs come from HBM; frac_warm = 20 back-to-back launches on ONE set "
             "(Infinity-Cache resident below 256 MB: not an HBM reading)")


def roofline_leg(args, dev, net, reps=50):
    """Primary object: the dominant cost-volume launch of the timed path -- the fused cost-volume stage 1 at l0, in the form the
    forward runs it (fused.cv_stage1's own switch): from 20 000 rows per launch the select-k pre-pass + `cv1_rr_kernel` (batch 1
    of a 64 x 1800 pair since round 5), below that `cv1_kernel` (in-kernel select-k grouping + gather/encode + six 1x1
    convolutions + masked softmax pooling) -- at the timed batch, with
    SURVEY 8(d)'s accounting: achieved = its algorithmic bytes (A1 + P1) per launch / its average launch duration
    (HIP events on the launch stream), peak 8 TB/s.  `traffic` is its measured HBM traffic per launch from the committed
    counter passes (`traffic_source`).  The kernel does not move the algorithmic bytes (it keeps the (N,K,10+2C) /
    (N,K,64) tensors in LDS): `mfma` gives the executed matrix-core rate against the peak of the dtype issued (fp16,
    2.5 PFLOP/s)."""
    B = args.batch
    half = args.features == "f16"
    L = recorded_cost_volume(dev, B, args.height, args.width, half)["l0"]
    sec = _time_launches(L["run1"], dev, reps)
    cold_sec, cold_ring = _time_ring(*L["call1"], dev)
    cb = cost_volume_bytes(L["N"], L["C"], L["Kq"], L["Kp"], 2 if half else 4)
    nbytes = (cb["A1"] + cb["P1"]) * B
    flops = cv1_flops(L["N"], L["C"], L["Kq"]) * B
    gbs = nbytes / sec / 1e9
    ex = MFMA_PRODUCTS if args.products == "split" else 1
    chain = L["N"] * L["Kq"] * B >= pkg("fused")._prepass_rows(1, B)      # (fused.cv_stage1's own switch: what the timed path runs)
    traffic, traffic_source = (None, "--no-live-traffic") if args.no_live_traffic else _live_traffic(B, args.features)
    if traffic is None:                                 # the committed passes of the same kernel and shape
        why = traffic_source
        traffic = _pmc_traffic("cv1_rr_kernel" if chain else "cv1_kernel", B, args.features)
        traffic_source = (os.path.relpath(PMC_SUMMARY, ROOT) + _pmc_commit() + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; "
                          "not measured in this run: %s)" % why)
    out = {"bound": "hbm", "kernel": "%s (fused cost-volume stage 1 at l0 on the tensors of a real forward: %d points, K=%d, batch %d, %s features)"
                                     % ("group_select_k pre-pass + cv1_rr_kernel, timed together" if chain else "cv1_kernel", L["N"], L["Kq"], B, args.features),
           "achieved": round(gbs, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 5),
           "traffic": traffic, "traffic_source": traffic_source,
           "us_per_launch": round(sec * 1e6, 3), "algorithmic_bytes_per_launch": int(nbytes),
           # the same launch with its inputs out of every cache (ring of tensor sets, _time_ring); `frac` above is the launch as the
           # timed loop runs it: a batch-1 pair's tensors (8 lanes x a few MB) live in L2 / the Infinity Cache
           "frac_cold": round(nbytes / cold_sec / 1e9 / HBM_PEAK_GBS, 5), "us_per_launch_cold": round(cold_sec * 1e6, 3), "cold_ring": cold_ring,
           "carries_riders_in_the_forward": L["riders"],      # batch 1-2: the forward's launch also runs the level's two set-upconv jobs
           "in_the_forward": ("cv1_setconv_rr_kernel (this stage + the level's two set-upconv stage-1 jobs in one launch) behind group_select_k"
                              if chain and L["riders"] else "cv1_setconv_kernel (this stage + the level's two set-upconv stage-1 jobs in one launch)"
                              if L["riders"] else "the same launches"),
           "mfma": {"executed_TFLOPs": round(ex * flops / sec / 1e12, 2), "algorithmic_TFLOPs": round(flops / sec / 1e12, 2),
                    "peak": F16_MFMA_PEAK_TFLOPS, "frac": round(ex * flops / sec / 1e12 / F16_MFMA_PEAK_TFLOPS, 5),
                    "issued_as": "v_mfma_f32_16x16x32_f16 x%d per 32-k pair, v_mfma_f32_16x16x16_f16 x%d on a 16-k tail (%s)" % (
                        ex,
                        ex, "fp16 hi+lo split operands, fp32 accumulate" if ex == 3 else "fp16-rounded operands")}}
    if (args.height, args.width) == (64, 1800):
        out["cost_volume_b8_f32"] = cost_volume_leg(dev, 8, args.height, args.width, False)
        out["cost_volume_b8_f16"] = cost_volume_leg(dev, 8, args.height, args.width, True)
        out["per_operator_b8_f32"] = per_operator_leg(dev, 8, False)
        out["per_operator_b8_f16"] = per_operator_leg(dev, 8, True)
        out["per_operator_b64_f32"] = per_operator_leg(dev, 64, False)
        out["per_operator_b64_f16"] = per_operator_leg(dev, 64, True)
        out["per_operator_all_levels_b8_f16"] = per_operator_all_levels_leg(dev, 8, True)
        # "take the roofline reading at C3 / C5 batch sizes" -- configs[4]'s workload (128 x 2048 scans: l0 is
        # 32 x 256 = 8192 points) at batch 8 on the per-operator (HBM-bound) kernels, fp16 (configs[2]'s storage) and fp32
        out["per_operator_hires_b8_f16"] = per_operator_leg(dev, 8, True, table=PER_OPERATOR_LEVELS_HIRES)
        out["per_operator_hires_b8_f32"] = per_operator_leg(dev, 8, False, table=PER_OPERATOR_LEVELS_HIRES)
        out["per_operator_hires_all_levels_b8_f16"] = per_operator_all_levels_leg(dev, 8, True, table=PER_OPERATOR_LEVELS_HIRES)
        out["per_operator_hires_all_levels_b8_f32"] = per_operator_all_levels_leg(dev, 8, False, table=PER_OPERATOR_LEVELS_HIRES)
        for key in ("per_operator_hires_b8_f16", "per_operator_hires_b8_f32"):       # counter traffic beside it (committed passes)
            out[key]["traffic"] = _hires_traffic(out[key]["features"])
    return out


# ----------------------------------------------------------------------------- CPU baseline leg
def cpu_baseline_leg(args, net):
    """The numpy/C restatement of the SAME forward (oracle/, kind 'port') on the host: one core, and all cores
    (one single-threaded worker process per core, whole pairs each -- oracle/cpu_bench.py)."""
    from oracle import cpu_bench
    synth = pkg("synth")
    params = {k: v.detach().cpu().numpy() for k, v in net.store.state_dict().items()}
    f1, f2 = synth.frame_pair(1, args.height, args.width, seed=900)
    cores = os.cpu_count() or 1
    per_worker = 2
    r = cpu_bench.run(params, f1, f2, args.cpu_pairs, per_worker, cores)
    return {"value": round(r["all_cores"], 3), "unit": "frame-pairs/s", "cores": cores, "cores_available": cores, "kind": "port",
            "one_core": round(r["one_core"], 4), "one_process_all_threads": round(r["one_process_all_threads"], 4),
            "sample": "oracle/ops_np.get_model_from_projection (numpy fp32 + C grouping oracle) on one %dx%d pair: "
                      "one core %d pairs in %.1f s; all cores (`value`) = %d single-threaded worker processes x %d pairs in %.1f s "
                      "(frame pairs are independent: the split the GPUs use); one_process_all_threads = SURVEY 8(d)'s literal "
                      "recipe, C grouping with a thread per core + threaded BLAS in ONE process, %d pairs in %.1f s"
                      % (args.height, args.width, args.cpu_pairs, r["seconds"][0], cores, per_worker, r["seconds"][1],
                         r["pairs_threaded"], r["seconds"][2])}


def batch_rate(dev, B, H, W, lanes, products, features, steps=240, ready=False, profile="dense"):
    """frame-pairs/s of a fresh net at batch B through `lanes` captured graphs, inputs resident in HBM (`ready`: submit()'s ordering
    argument -- False: the caller owns it, the inputs were synchronised; None: submit()'s default, ordered behind the current stream)."""
    import torch
    model, synth, fused = pkg("model"), pkg("synth"), pkg("fused")
    with fused.products(products):
        net = model.PWCLONet(dev, seed=0, feature_dtype=torch.float16 if features == "f16" else torch.float32)
        pairs = []
        for i in range(4):
            f1, f2 = synth.frame_pair(B, H, W, seed=77 + 16 * i, profile=profile, starved=False)
            pairs.append(torch.cat([torch.from_numpy(f1), torch.from_numpy(f2)], 0).to(dev))
        net.capture(B, H, W, lanes=lanes)
        for i in range(2 * lanes):
            net.submit(i % lanes, pairs[i % 4], ready=ready)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for i in range(steps):
            net.submit(i % lanes, pairs[i % 4], ready=ready)
        torch.cuda.synchronize(dev)
        return round(B * steps / (time.perf_counter() - t0), 1)


def raw_cloud_rate(dev, B, H, W, lanes, points=150000, steps=400, r_max=60.0):
    """frame-pairs/s when a step starts from RAW clouds: 2 x `points` KITTI-shaped points per pair
    (5 % zero padding, ranges to `r_max` m so the 35 m crop bites) -> elo_input_stage -> the pyramid, all inside the lane's
    graph; clouds resident in HBM."""
    import numpy as np
    import torch
    model = pkg("model")
    net = model.PWCLONet(dev, seed=0)
    rng = np.random.default_rng(3)
    clouds = []
    for _ in range(4):
        az = rng.uniform(-np.pi, np.pi, (B, 2 * points))
        el = np.deg2rad(rng.uniform(-24.8, 2.0, (B, 2 * points)))
        r = rng.uniform(2.0, r_max, (B, 2 * points))
        c = np.stack([r * np.cos(el) * np.cos(az), r * np.cos(el) * np.sin(az), r * np.sin(el)], -1).astype(np.float32)
        c[rng.random((B, 2 * points)) < 0.05] = 0
        clouds.append(torch.from_numpy(c).to(dev))
    net.capture(B, H, W, lanes=lanes, num_points=points)
    for i in range(2 * lanes):
        net.submit_points(i % lanes, clouds[i % 4], ready=False)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for i in range(steps):
        net.submit_points(i % lanes, clouds[i % 4], ready=False)
    torch.cuda.synchronize(dev)
    return round(B * steps / (time.perf_counter() - t0), 1)


def hires_leg(dev, lanes):
    """BASELINE configs[4]'s scans: 128x2048 range images (levels 32x256 / 16x128 / 8x64 / 8x32).  Pairs/s at batch 1 and
    batch 8 through `lanes` captured graphs, the cost-volume launches of a batch-8 forward on its own tensors, and the
    LDS footprint of the kernels that stage neighbour windows / W rings there (bytes per workgroup -> workgroups per CU of
    160 KB; the register-resident kernels are held to 2 by their 116-122 VGPRs)."""
    H, W = 128, 2048
    cv = cost_volume_leg(dev, 8, H, W, True, reps=10)
    sel_words = lambda P, kH, kW: 4 * kH * (63 + kW) + 2 * 64 * 32 + max(P, 8) * 64 + 64 + 64 + ((kH * kW + 3) & ~3) + P * 128
    lds = {"select_k_dense_11x41 (16 waves per tile)": 4 * sel_words(16, 11, 41), "select_k_dense_11x41 (4 waves per tile)": 4 * sel_words(4, 11, 41),
           "random_k_dense_3x5_K4 (2 x 64 centres)": 4 * ((15 + 7) & ~7) + 16 * (1 + 3) * (63 + 5) + 4 * (129 * 4 + 256),
           "cv1_rr / cv2_rr (W ring 16 KB, aliased by the pooling scratch)": 4 * (2 * 128 * 36 + 128)}
    return {"grid": "%dx%d" % (H, W), "unit": "frame-pairs/s", "lanes": lanes,
            "batch1": batch_rate(dev, 1, H, W, lanes, "split", "f32", steps=160),
            "batch8_f32": batch_rate(dev, 8, H, W, lanes, "split", "f32", steps=48),
            "batch8_f16_features": batch_rate(dev, 8, H, W, lanes, "split", "f16", steps=48),
            "cost_volume_b8_f16": {k: cv[k] for k in ("us", "bytes", "frac", "frac_cold", "frac_warm", "us_warm", "levels", "mfma")},
            "lds_bytes_per_workgroup": lds, "workgroups_per_cu_by_lds": {k: min(160 * 1024 // v, 32) for k, v in lds.items()}}


def dense_f32_leg(args):
    """The same timed loop on the comparison build (libelo_hip_f32.so, -DELO_DENSE_F32: every 1x1 convolution on
    v_mfma_f32_16x16x4_f32, true fp32 products) in a child process: what the fp16 hi/lo split buys."""
    env = dict(os.environ, ELO_DENSE_F32="1")
    cmd = [sys.executable, os.path.abspath(__file__), "--no-legs", "--steps", str(max(args.steps, 200)), "--warmup",
           str(args.warmup), "--batch", str(args.batch), "--height", str(args.height), "--width", str(args.width),
           "--lanes", str(args.lanes)]
    try:
        out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
        d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
        return {"value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"],
                "mfma": "v_mfma_f32_16x16x4_f32 (peak %.1f TFLOP/s)" % F32_MFMA_PEAK_TFLOPS}
    except Exception as e:                                  # the comparison build is optional: say why it is missing
        return {"value": None, "error": "%s: %s" % (type(e).__name__, str(e)[:200])}


def train_dp_leg(args, dev, rank, world, dist, group=None):
    """BASELINE configs[3]'s exchange under a clock: per-GPU batch 8, one optimisation step = zero grads -> forward
    (batch-statistics BN, dropout) -> get_loss -> backward -> ONE all-reduce of the flat 899 134-float gradient bucket
    (RCCL over xGMI when world > 1) -> Adam (main.py:344-397 + SURVEY 8(e)), captured as hipGraphs around the eager
    collective (training.Trainer.step_graph).  All ranks run it; barrier + synchronize on both sides, MAX over ranks."""
    import torch
    group = world > 1 if group is None else group
    model, training, synth, mu = pkg("model"), pkg("training"), pkg("synth"), pkg("model_util")
    B, H, W = 8, args.height, args.width
    net = model.PWCLONet(dev, seed=0)
    tr = training.Trainer(net, capturable=True)
    f1, f2 = synth.frame_pair(B, H, W, seed=500 + rank)
    a, b = torch.from_numpy(f1).to(dev), torch.from_numpy(f2).to(dev)
    T = torch.eye(4, device=dev).repeat(B, 1, 1)
    T[:, 0, 3] = 0.8
    q_gt, t_gt = mu.preprocess_gt(T, T, T, [0] * B)
    tr.capture(a, b, q_gt, t_gt)
    for _ in range(2):
        tr.step_graph(a, b, q_gt, t_gt)
    torch.cuda.synchronize(dev)
    if group:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.train_steps):
        loss = tr.step_graph(a, b, q_gt, t_gt)
    torch.cuda.synchronize(dev)
    if group:
        dist.barrier()
    el = time.perf_counter() - t0
    if group:
        t = torch.tensor([el], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    assert torch.isfinite(loss).all()
    return {"value": round(world * B * args.train_steps / el, 2), "unit": "frame-pairs/s (training)", "n_gpus": world,
            "batch_per_gpu": B, "steps": args.train_steps, "ms_per_step": round(el / args.train_steps * 1e3, 3),
            "collective": "one all_reduce(SUM) of the flat gradient bucket per step, %d floats = %.2f MB, backend %s"
                          % (tr.bucket.flat.numel(), tr.bucket.flat.numel() * 4 / 1e6,
                             dist.get_backend() if group else "none (1 rank)")}


# ----------------------------------------------------------------------------- multi-rank evidence (N > 1)
def pin_to_numa_node(torch, local):
    """Bind this rank to the CPUs of its GPU's NUMA node (host-side submit latency; best effort: {} when sysfs does not
    say).  Returns {"numa_node", "cpus"} for the line."""
    try:
        p = torch.cuda.get_device_properties(local)
        bdf = "%04x:%02x:%02x.0" % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
        node = int(open("/sys/bus/pci/devices/%s/numa_node" % bdf).read())
        if node < 0:
            return {"numa_node": None}
        cpus = set()
        for part in open("/sys/devices/system/node/node%d/cpulist" % node).read().strip().split(","):
            lo, _, hi = part.partition("-")
            cpus.update(range(int(lo), int(hi or lo) + 1))
        os.sched_setaffinity(0, cpus)
        return {"numa_node": node, "cpus": len(cpus)}
    except Exception as e:          # no sysfs entry, a container without the node files, a torch without the pci ids
        return {"numa_node": None, "why": "%s: %s" % (type(e).__name__, str(e)[:80])}


RCCL_ALGO = {"0": "Tree", "1": "Ring", "2": "CollNetDirect", "3": "CollNetChain", "4": "NVLS", "5": "NVLSTree"}   # NCCL_ALGO_* (nccl devcomm.h)
RCCL_PROTO = {"0": "LL", "1": "LL128", "2": "Simple"}                                                                    # NCCL_PROTO_*


def parse_rccl_log(path):
    """What RCCL said about itself (NCCL_DEBUG=INFO, subsystems INIT,GRAPH,TUNING) in this rank's log file: the rank count
    of the communicator ("... rank 0 nranks 8 cudaDev 0 ... Init COMPLETE"), the channel count ("16 coll channels, ..."),
    the transports of the ring / tree links ("Channel 00 : 0[..] -> 1[..] via P2P/IPC"), whether rings and trees were
    connected, and the algorithm / protocol the tuner picked per collective ("AllReduce: 3596536 Bytes -> Algo 1 proto 2
    time 52.3": numbers are NCCL_ALGO_* / NCCL_PROTO_*; newer logs spell them out).  tests/test_bench_cpu.py feeds it canned
    excerpts of both spellings; it has not met a live RCCL log on this pool yet (no multi-GPU node)."""
    import re
    out = {"log": path, "nranks": None, "channels": None, "transports": [], "algo_proto": None, "rings": None, "trees": None}
    try:
        text = open(path, errors="replace").read()
    except OSError:
        return out
    m = re.search(r"nranks (\d+)", text)
    out["nranks"] = int(m.group(1)) if m else None
    m = re.search(r"(\d+) coll channels", text)
    out["channels"] = int(m.group(1)) if m else None
    out["transports"] = sorted(set(re.findall(r"via (\S+)", text)))[:8]
    out["rings"] = bool(re.search(r"Connected all rings", text)) or (True if re.search(r"Channel \d+/\d+ *:", text) else None)
    out["trees"] = bool(re.search(r"Connected all trees", text)) or (True if re.search(r"\bTrees \[", text) else None)
    picks = set()
    for coll, algo, proto in re.findall(r"(\w+): \d+ Bytes -> [Aa]lgo(?:rithm)? (\w+) [Pp]roto(?:col)? (\w+)", text):
        picks.add("%s:%s/%s" % (coll, RCCL_ALGO.get(algo, algo), RCCL_PROTO.get(proto, proto)))
    if not picks:
        for algo, proto in re.findall(r"[Aa]lgo(?:rithm)? (\w+) [Pp]roto(?:col)? (\w+)", text):
            picks.add("%s/%s" % (RCCL_ALGO.get(algo, algo), RCCL_PROTO.get(proto, proto)))
    out["algo_proto"] = sorted(picks)[:8] or None
    return out


def rccl_leg(dist, dev, backend, world, rank, pose_block, numa, iters=10):
    """N > 1: the two collectives of the path under a clock of their own -- one all_reduce(SUM) of the flat gradient bucket
    (899 134 floats, 3.6 MB: configs[3]'s exchange) and the all_gather of a repeat's pose log -- plus what the backend
    reported (rank count, channels, transports, algorithm / protocol where the log names them)."""
    import torch
    sync = (lambda: torch.cuda.synchronize(dev)) if dev.type == "cuda" else (lambda: None)
    bucket = torch.zeros(899134, device=dev)
    gathered = [torch.empty_like(pose_block) for _ in range(world)]

    def clock(fn):
        for _ in range(3):
            fn()
        sync()
        dist.barrier()
        t0 = time.perf_counter()
        for _ in range(iters):
            fn()
        sync()
        el = torch.tensor([(time.perf_counter() - t0) / iters], dtype=torch.float64, device=dev)
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        return round(float(el.item()) * 1e6, 1)

    ar = clock(lambda: dist.all_reduce(bucket))
    ag = clock(lambda: dist.all_gather(gathered, pose_block))
    out = {"world_size": dist.get_world_size(), "backend": backend, "all_reduce_us": ar, "all_reduce_bytes": bucket.numel() * 4,
           "all_gather_us": ag, "all_gather_bytes": pose_block.numel() * 4 * world, "numa": numa}
    if backend == "nccl":
        out.update(parse_rccl_log(os.environ.get("NCCL_DEBUG_FILE", "")))
    return out


# ----------------------------------------------------------------------------- main
def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus))
    import numpy as np
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        print("[bench] --gpus %d but the launcher started %d ranks: reporting n_gpus = %d" % (args.gpus, world, world),
              file=sys.stderr, flush=True)
    dry = args.dry_run
    if not dry and not torch.cuda.is_available():
        raise SystemExit("bench.py needs an AMD GPU: the hot path has no CPU fallback (--dry-run rehearses the "
                         "multi-process path without one)")
    if dry:
        dev = torch.device("cpu")
    else:
        have = torch.cuda.device_count()
        if local >= have:
            if os.environ.get("ELO_BENCH_SHARE_GPU") != "1":      # (set by the test that rehearses 2 ranks on a 1-GPU box)
                raise SystemExit("bench.py: rank %d has LOCAL_RANK %d but this node shows %d GPU(s): --gpus %d needs one GPU "
                                 "per rank (ELO_BENCH_SHARE_GPU=1 lets ranks double up, for rehearsals only)"
                                 % (rank, local, have, world))
            local %= have
        torch.cuda.set_device(local)
        dev = torch.device("cuda", local)
    numa = pin_to_numa_node(torch, local) if (not dry and world > 1) else {"numa_node": None}
    backend = None
    sync = (lambda: None) if dry else (lambda: torch.cuda.synchronize(dev))
    # ELO_BENCH_FORCE_DIST=1: a ONE-rank process group -- on a 1-GPU box the whole distributed path of this file
    # (init_process_group("nccl", device_id=...), the barriers, the pose all-gather inside the timed region, the flat-bucket
    # all-reduce of train_dp and rccl_leg, parse_rccl_log on the live NCCL_DEBUG_FILE) runs through RCCL itself
    group = world > 1 or os.environ.get("ELO_BENCH_FORCE_DIST") == "1"
    if group:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:
            os.environ.setdefault("MASTER_PORT", str(_free_port()))
            pkg("distributed").FlatGradBucket.force_collective = True      # (train_dp's all-reduce runs on the world of one)
        backend = os.environ.get("ELO_BENCH_BACKEND", "gloo" if dry else "nccl")          # nccl == RCCL on ROCm
        if backend == "nccl":
            os.environ.setdefault("NCCL_DEBUG", "INFO")                                   # ring / tree choice and the transport
            os.environ.setdefault("NCCL_DEBUG_SUBSYS", "INIT,GRAPH,TUNING")
            os.environ.setdefault("NCCL_DEBUG_FILE", "/tmp/elo_bench_rccl_%d.log" % os.getpid())   # (RCCL logs to stdout otherwise)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
        if rank == 0:
            print("[bench] %s process group up: %d ranks (dist.get_world_size() = %d)"
                  % (backend, world, dist.get_world_size()), file=sys.stderr, flush=True)

    B, H, W = args.batch, args.height, args.width
    lanes = 1 if (args.no_graph or dry) else max(1, args.lanes)
    warmup = max(args.warmup, 2 * lanes)
    if dry:
        net = None
        pose_log = torch.zeros((max(args.steps, warmup), B, 7))

        def step(i):
            pose_log[i, :, 0] = 1.0

        def begin_repeat(reset=False):
            pass

        def end_repeat(n):
            pass
    else:
        model, synth = pkg("model"), pkg("synth")
        pkg("fused").products(args.products).__enter__()          # for the whole run: packing, capture and the roofline leg
        net = model.PWCLONet(dev, seed=0, feature_dtype=torch.float16 if args.features == "f16" else torch.float32)
        pool = []
        for i in range(args.pool):                       # inputs resident in HBM before the timed region
            f1, f2 = synth.frame_pair(B, H, W, seed=1000 * rank + 10 * i)
            pool.append(torch.cat([torch.from_numpy(f1), torch.from_numpy(f2)], 0).to(dev))   # (2B,H,W,3) = [f1 | f2]
        if args.no_graph:
            net.forward(pool[0][:B], pool[0][B:])
        else:
            # a lane's l0 pose-head kernel writes replay r's [q | t] block into slot r of the lane's ring
            per_lane = -(-args.steps // lanes)            # replays of a lane per repeat: this or one less -> slots a multiple of both
            net.capture(B, H, W, lanes=lanes, pose_ring=max(2, -(-warmup // lanes), per_lane * max(1, per_lane - 1) * 2), sample=pool[0],
                        fresh_orders=args.fresh_orders, check_every=0 if args.products == "half" else args.check_every)
        # every step's l0 pose [q | t] is kept: one (B,7) row block per step in HBM, collected from the lanes' rings at
        # the end of each repeat INSIDE the timed region (one strided copy per lane; round 1-2 copied one block out per
        # step, a launch that sat ~14 us in every lane's serial chain); with N > 1 ranks the blocks are all-gathered ONCE
        # per repeat, inside the timed region too (7 floats per pair: the only exchange of the inference path,
        # main.py:557-572)
        pose_log = torch.empty((max(args.steps, warmup), B, 7), device=dev)
        if not args.no_graph and args.inputs == "in-place":
            for lane in range(lanes):                    # untimed: every lane's input buffer gets its own synthetic pair
                net.lane_input(lane).copy_(pool[lane % len(pool)])
            sync()

        ready = False if args.submit_order == "caller" else None

        def step(i):
            pair = pool[i % len(pool)]
            if args.no_graph:                            # the l0 pose-head kernel writes the log row itself
                return net.forward(pair[:B], pair[B:], pose_out=pose_log[i])
            if args.inputs == "in-place":                # the lane's buffer holds its pair (filled before the timed region)
                return net.submit(i % lanes, ready=ready)
            return net.submit(i % lanes, pair, ready=ready)   # step i rides lane i % lanes: one copy in (the stacked pair), graph replay

        def begin_repeat(reset=False):
            if not args.no_graph:
                for lane in range(lanes):                # (mark: host bookkeeping only -- no launch inside the timed region)
                    net.reset_poses(lane) if reset else net.mark_poses(lane)

        def end_repeat(n):
            if not args.no_graph:
                for lane in range(min(lanes, n)):
                    with torch.cuda.stream(net.lane_stream(lane)):
                        pose_log[lane:n:lanes].copy_(net.lane_poses(lane), non_blocking=True)
    gathered = [torch.empty_like(pose_log[:args.steps]) for _ in range(world)] if group else None

    # Host submission.  A step costs the host ~50 us (one copy + one graph launch), so with one submitting thread the four
    # hardware queues receive their first forward of a repeat 50 us apart -- at the driver's 20 steps per repeat (2 ms) the
    # last queue idles 150 us, 7 % of the repeat.  --submit-threads N (step i -> thread i % N; lane = i % lanes, queue =
    # lane % 4: with N = 4 a thread feeds one queue, in order) was built to enqueue side by side and measured SLOWER (the
    # Python part of a submit holds the interpreter lock: 9.65 k pairs/s with one thread, 9.3 k with four); kept as an option.
    nthreads = 1 if (dry or args.no_graph) else max(1, min(args.submit_threads, lanes))
    if lanes % nthreads:
        nthreads = 1                                       # (a lane is fed by ONE thread: its host-side counters are not shared)
    if nthreads > 1:
        import threading
        go, done = threading.Barrier(nthreads), threading.Barrier(nthreads)
        job = {"n": 0, "stop": False, "err": None}

        def worker(w):
            torch.cuda.set_device(dev)
            while True:
                go.wait()
                if job["stop"]:
                    return
                try:
                    for i in range(w, job["n"], nthreads):
                        step(i)
                except Exception as e:                     # noqa: BLE001 -- reported by the main thread
                    job["err"] = e
                done.wait()
        pool_threads = [threading.Thread(target=worker, args=(w,), daemon=True) for w in range(1, nthreads)]
        for th in pool_threads:
            th.start()

        def submit_all(n):
            job["n"] = n
            go.wait()
            for i in range(0, n, nthreads):
                step(i)
            done.wait()
            if job["err"] is not None:
                raise job["err"]
    else:
        def submit_all(n):
            for i in range(n):
                step(i)

    begin_repeat(reset=True)
    submit_all(warmup)
    end_repeat(warmup)
    sync()
    begin_repeat(reset=True)                              # (untimed: the rings start the timed repeats at slot 0)
    sync()
    if group:
        dist.all_gather(gathered, pose_log[:args.steps].contiguous())     # untimed: RCCL sets its channels up on first use
    repeats, total = [], 0.0
    while total < MIN_TIMED_S and len(repeats) < 200:
        sync()
        if group:
            dist.barrier()
        t0 = time.perf_counter()
        begin_repeat()
        submit_all(args.steps)
        end_repeat(args.steps)
        sync()
        if group:
            dist.all_gather(gathered, pose_log[:args.steps].contiguous())
            sync()
            dist.barrier()
        elapsed = time.perf_counter() - t0
        if group:                                     # the clock of a repeat is its slowest rank
            t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        repeats.append(elapsed)
        total += elapsed
        if dry:
            break
    assert torch.isfinite(pose_log[:args.steps]).all()
    elapsed = statistics.median(repeats)

    line = {
        "metric": "frame-pairs/sec (KITTI 64x1800 range image)" if (H, W) == (64, 1800)
                  else "frame-pairs/sec (%dx%d range image)" % (H, W),
        "value": None if dry else round(world * B * args.steps / elapsed, 3),
        "unit": "frame-pairs/s",
        "n_gpus": world, "steps": args.steps, "warmup": warmup,
        "warmup_requested": args.warmup,                  # `warmup` = max(--warmup, 2 per lane): every lane's graph has replayed twice
        "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "repeats": len(repeats), "ms_per_step_repeats": [round(r / args.steps * 1e3, 4) for r in repeats[:16]],
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": ("f32 (1x1 convolutions as 3 fp16-MFMA products of hi/lo-split fp32 operands, fp32 accumulate%s)"
                  % ("; true-fp32 MFMA comparison build" if os.environ.get("ELO_DENSE_F32") == "1" else ""))
                 if args.products == "split" else
                 "f16 products (--products half: 1x1 convolutions as ONE fp16-MFMA product of fp16-rounded operands, fp32 "
                 "accumulate) -- not the fp32 parity path",
        "data": "dry-run (no GPU work: rehearsal of the multi-process path)" if dry else "synthetic",
        "config": {"workload": "full 4-level PWC pyramid (set-conv + attentive cost volume + warp-refinement + "
                               "set-upconv), %dx%d range-image pairs, batch %d per GPU, %s feature storage, "
                               "random-init weights, %s" % (H, W, B, "fp16" if args.features == "f16" else "fp32",
                                                           "eager launches" if args.no_graph else
                                                           "hipGraph replay, %d forwards in flight fed by %d host thread%s (%s)%s" % (
                                                               lanes, nthreads, "s" if nthreads > 1 else "",
                                                               "submit ready=False: inputs resident and synchronised, the bench owns the ordering" if args.submit_order == "caller"
                                                               else "submit()'s default ordering behind the current stream", ", fresh visiting orders per replay (pool of %d)" % args.fresh_orders
                                                               if args.fresh_orders else ", one fixed draw of the visiting orders")),
                   "global_batch": B * world, "parallelism": "dp%d" % world,
                   # every choice of kernel form, as the value the timed graphs were captured under (efficientlo-net_amd/tuning.py)
                   "tuning": None if dry else (getattr(net, "captured_tuning", None) or pkg("tuning").snapshot())},
    }
    if not dry and not args.no_graph and args.products != "half" and args.check_every:
        # the fp16 split's production guard: every N-th replay of a lane ran on the range-checked kernels INSIDE the timed loop
        line["range_check"] = {"every_nth_replay": args.check_every, "violations": net.range_violations()}
        assert line["range_check"]["violations"] == 0
    legs = not args.no_legs and not dry
    if legs and args.train_steps > 0 and (H, W) == (64, 1800):
        train = train_dp_leg(args, dev, rank, world, dist, group)          # every rank takes part (one collective per step)
        line["train_dp"] = train
    if legs and rank == 0 and world == 1:
        line["roofline"] = roofline_leg(args, dev, net)
        if not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline_leg(args, net)
        if (B, H, W) == (1, 64, 1800) and not args.no_graph and args.products == "split" and args.features == "f32":
            line["from_raw_clouds"] = {"unit": "frame-pairs/s", "lanes": lanes, "points_per_frame": 150000,
                                       "value": raw_cloud_rate(dev, B, H, W, lanes)}
            # BASELINE configs[2]: batch 8 through the same lanes -- fp32 storage, fp16 feature storage (the configuration
            # configs[2] names), and fp16 storage with fp16 products on top
            line["batch8"] = {"unit": "frame-pairs/s", "lanes": lanes,
                              "f32": batch_rate(dev, 8, H, W, lanes, "split", "f32"),
                              "f16_features": batch_rate(dev, 8, H, W, lanes, "split", "f16"),
                              "f16_features_f16_products": batch_rate(dev, 8, H, W, lanes, "half", "f16")}
            # submit()'s DEFAULT (ordered behind the caller's current stream + record_stream) against the unordered fast path the timed
            # loop uses, same lanes, saturated (240 steps between two synchronisations)
            line["submit_ordering"] = {"unit": "frame-pairs/s", "batch": 1, "steps": 240,
                                       "caller_owned_ready_False": batch_rate(dev, 1, H, W, lanes, "split", "f32"),
                                       "default_ordered": batch_rate(dev, 1, H, W, lanes, "split", "f32", ready=None)}
            # the density the reference actually runs on (VERDICT r05 missing 5): synth profile "kitti" -- ~56 % of the grid valid, in
            # runs and blocks (dead beam rows, a sector without returns, sky rows) -- against the 95 %-filled scene every other leg uses;
            # raw clouds: 120 000 points per frame of which ~60 000 survive the 35 m crop (ranges to 68 m), as a projected HDL-64 scan
            sparse = {"unit": "frame-pairs/s", "lanes": lanes, "scene": "synth.range_image(profile='kitti'): ~56 % valid cells",
                      "batch1": batch_rate(dev, 1, H, W, lanes, "split", "f32", profile="kitti"),
                      "batch8_f16_features": batch_rate(dev, 8, H, W, lanes, "split", "f16", profile="kitti"),
                      "from_raw_clouds_60k_valid": raw_cloud_rate(dev, B, H, W, lanes, points=120000, r_max=68.0)}
            sparse["ratio_to_dense"] = {"batch1": round(sparse["batch1"] / line["submit_ordering"]["caller_owned_ready_False"], 3),
                                        "batch8_f16_features": round(sparse["batch8_f16_features"] / line["batch8"]["f16_features"], 3),
                                        "from_raw_clouds": round(sparse["from_raw_clouds_60k_valid"] / line["from_raw_clouds"]["value"], 3)}
            line["sparse"] = sparse
            line["hires"] = hires_leg(dev, lanes)
            if os.environ.get("ELO_DENSE_F32") != "1":
                line["dense_f32"] = dense_f32_leg(args)
    if group:                                         # every rank takes part; rank 0 reports
        line["rccl"] = rccl_leg(dist, dev, backend, world, rank, pose_log[:args.steps].contiguous(), numa)
    if rank == 0:
        print(json.dumps(line), flush=True)
    if group:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
