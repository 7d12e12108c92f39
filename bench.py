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

One process per GPU (torchrun sets RANK/LOCAL_RANK/WORLD_SIZE); frame pairs are
independent, so ranks shard the stream with no data-path collective; every step's pose is
logged on the device and the logs are all-gathered (7 floats per pair) once, inside the timed
region, so rank 0 can chain them (main.py:557-572): the only exchange the path has.  scaling = weak.

Throughput design: the whole forward is one hipGraph; `--lanes` graphs (default 8), each with its own static
buffers, dealt over one stream per hardware queue (model.distinct_queue_streams), keep independent forwards
in flight (step i rides lane i % lanes).

Besides the contract line this prints, on rank 0 at N=1:
  roofline     -- the dominant cost-volume kernel (fused stage 1 at l0) timed live with HIP events
  cpu_baseline -- the CPU oracle (oracle/, kind "port") timed on a bounded sample
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s peak


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
    ap.add_argument("--pool", type=int, default=8, help="distinct synthetic pairs kept in HBM")
    ap.add_argument("--lanes", type=int, default=8,
                    help="independent forwards in flight (hipGraphs, dealt over the 4 hardware queues: use a multiple of 4)")
    ap.add_argument("--products", choices=("split", "half"), default="split",
                    help="dense products of the fused kernels: split = fp32-class (three fp16 MFMA products, the headline); "
                         "half = ONE fp16 product (fp16 arithmetic, BASELINE configs[2]; not the fp32 parity path)")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of hipGraph replay")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-legs", action="store_true",
                    help="only the timed loop: no roofline / cpu_baseline / from_raw_clouds / batch8 legs (profiling runs)")
    ap.add_argument("--cpu-pairs", type=int, default=60, help="pairs in the CPU-oracle sample")
    return ap.parse_args()


# ----------------------------------------------------------------------------- roofline leg
FP32_MFMA_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 dense peak (= the fp32 vector peak)
F16_MFMA_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: BF16/FP16 MFMA ~2.5 PF dense (2495 TF measured)
MFMA_PRODUCTS = 3              # fp16 matrix-core products per fp32-class product: hi*hi + hi*lo + lo*hi (elo_fused.hip)
# HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/r01_pmc_v3 and r01_pmc_v4/summary.json, collected by
# tools/pmc_collect.sh: FETCH_SIZE x2 + WRITE_SIZE, the gfx950 correction of MI355X_MICROARCH.md); counters cannot
# be read from inside bench.py.  Only for the exact shapes captured (l0 = 16x225, i.e. 64x1800 inputs); otherwise null.
PMC_TRAFFIC = {("cv1", 1): 3.65e6, ("cv1", 8): 13.64e6, ("A1", 8): 36.98e6, ("P1", 8): 96.68e6, ("A2", 8): 52.56e6,
               ("P2", 8): 66.88e6, ("A1", 64): 298.34e6, ("P1", 64): 772.78e6, ("A2", 64): 420.23e6, ("P2", 64): 534.60e6,
               ("A1", 64, "f16"): 168.88e6, ("P1", 64, "f16"): 389.00e6, ("A2", 64, "f16"): 218.92e6,
               ("P2", 64, "f16"): 269.17e6}


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


def _time_launches(fn, dev, reps):
    """Average duration of one launch of `fn`: `reps` back-to-back launches captured in a hipGraph (so the host's
    per-call cost, ~12 us from Python, is not in the number), replayed between two HIP events on torch's current
    stream == the stream _lib.stream_ptr() hands to the C ABI."""
    for _ in range(3):
        fn()
    torch.cuda.synchronize(dev)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
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


def roofline_leg(args, dev, net, reps=50):
    """The dominant cost-volume kernel of the timed path, timed live with HIP events on the stream it is launched
    on: the fused stage-1 kernel at l0 (select-k grouping + encode + six 1x1 convs on the matrix cores + masked
    softmax pooling in one launch).  It never materialises the operator-boundary tensors, so it is on the MFMA
    roofline, not the HBM one.  Its products are fp32-class but run on the FP16 matrix cores (each operand split
    into fp16 hi + lo, three v_mfma_f32_16x16x16_f16 per 16-k block, fp32 accumulation).  `achieved` is the
    ALGORITHMIC fp32 flops of the launch (2 x MACs of the six convolutions) over its duration, `peak` the dense MFMA
    peak of the dtype the path computes in (fp32: 157.3 TFLOP/s -- what the native fp32 MFMA could deliver at best);
    `executed_mfma_TFLOPs` / `frac_of_fp16_mfma_peak` count the three fp16 products actually issued against the
    2.5 PFLOP/s fp16 peak -- small by construction: with the matrix work this cheap the kernel is bound by vector
    instruction issue and latency.  `hbm_equivalent_GBps` is SURVEY 8(d)'s algorithmic bytes (A1+P1) over the
    same duration for comparison with the per-operator kernels (`per_operator`: the four HBM-bound kernels of the
    ELO_FUSED=0 cost volume, each against its own algorithmic bytes)."""
    ops, fused, tf_util, perm, pm = pkg("_ops"), pkg("fused"), pkg("tf_util"), pkg("perm"), pkg("pwclo_model")
    oh, ow = pm.pyramid_sizes(args.height, args.width)
    B, H, W, C, Kq = args.batch, oh[2], ow[2], 16, 6
    N = H * W
    if (args.height, args.width) != (64, 1800):
        PMC_TRAFFIC.clear()
    g = torch.Generator(device="cpu").manual_seed(0)
    synth = pkg("synth")
    f1, f2 = synth.frame_pair(B, H, W, seed=5)
    xyz1, xyz2 = torch.from_numpy(f1).to(dev), torch.from_numpy(f2).to(dev)
    feat1 = torch.randn((B, H, W, C), generator=g).to(dev)
    feat2 = torch.randn((B, H, W, C), generator=g).to(dev)
    order = torch.randperm(11 * 41, generator=g).to(torch.int32).to(dev)
    with tf_util.default_store(net.store), torch.no_grad(), tf_util.variable_scope('flow_embedding_l0'):
        P = fused.packed_layer
        layers = (P('CV_0', 10 + 2 * C, 128), P('CV_1', 128, 64), P('CV_2', 64, 64), P('CV_xyz', 10, 64),
                  P('sum_CV_0', 128, 128, row_order=list(range(64, 128)) + list(range(64))), P('sum_CV_1', 128, 64))
    grouping = fused.Grouping(order, [11, 41], 1000)
    run = lambda: fused.cv_stage1(xyz1.reshape(B, N, 3), feat1.reshape(B, N, C), xyz2, feat2, None, None, *layers,
                                  group=grouping, K=Kq)
    sec = _time_launches(run, dev, reps)
    flops = cv1_flops(N, C, Kq) * B
    cvb = cost_volume_bytes(N, C, Kq, 4)
    tfs = flops / sec / 1e12
    LEVELS = {"l0": (oh[2], ow[2], 16, 6, (11, 41)), "l1": (oh[3], ow[3], 32, 6, (7, 25)),
              "l2": (oh[4], ow[4], 64, 6, (5, 15)), "l2_origin": (oh[4], ow[4], 64, 32, (5, 35))}      # pwclo_model.py cost_volume calls

    def per_operator_leg(batch, half=False, level="l0"):
        """The four cost-volume kernels of the ELO_FUSED=0 path at one level (SURVEY 8(d) terms A1, P1, A2, P2), each
        against its own algorithmic bytes, and the four together.  `half`: fp16 feature storage (s = 2; BASELINE
        configs[2]), fp32 arithmetic."""
        H, W, C, Kq, win = LEVELS[level]
        N = H * W
        order = torch.randperm(win[0] * win[1], generator=g).to(torch.int32).to(dev)
        cvb = cost_volume_bytes(N, C, Kq, 4, 2 if half else 4)
        cast = (lambda x: x.half()) if half else (lambda x: x)
        elo = pkg()
        Kp = 4
        fb1, fb2 = synth.frame_pair(batch, H, W, seed=6)
        x1, x2 = torch.from_numpy(fb1).to(dev), torch.from_numpy(fb2).to(dev)
        ft1 = cast(torch.randn((batch, H, W, C), generator=g).to(dev))
        ft2 = cast(torch.randn((batch, H, W, C), generator=g).to(dev))
        hw = torch.from_numpy(synth.hw_index(batch, H, W)).to(dev)
        idx_q, _, _, m_q = elo.fused_conv_select_k(x1, x2, hw, order, H, W, N, win[0], win[1], Kq, 0, 1000.0, 1, 1,
                                                  want_valid=False)
        order_p = torch.randperm(3 * 5, generator=g).to(torch.int32).to(dev)
        idx_p, _, _, m_p = elo.fused_conv_random_k(x1, x1, hw, order_p, H, W, N, 3, 5, Kp, 0, 1000.0, 1, 1,
                                                  want_valid=False)
        m_q, m_p = m_q.reshape(batch, N, Kq), m_p.reshape(batch, N, Kp)
        cost = cast(torch.randn((batch, H, W, 64), generator=g).to(dev))
        lq, vq = (cast(torch.randn((batch, N, Kq, 64), generator=g).to(dev)) for _ in range(2))
        lp, vp = (cast(torch.randn((batch, N, Kp, 64), generator=g).to(dev)) for _ in range(2))
        legs = {"A1": ("cv_encode1_col_kernel" if C == 16 else "cv_encode1_vec_kernel", lambda: ops.cv_encode1(x1.reshape(batch, N, 3), ft1.reshape(batch, N, C),
                                                                        x2, ft2, idx_q, m_q)),
                "P1": ("softmax_pool_vec_kernel", lambda: ops.masked_softmax_pool(lq, vq, m_q)),
                "A2": ("cv_encode2_vec_kernel", lambda: ops.cv_encode2(x1, ft1, cost, idx_p, m_p)),
                "P2": ("softmax_pool_vec_kernel", lambda: ops.masked_softmax_pool(lp, vp, m_p))}
        terms, tot_b, tot_s = {}, 0, 0.0
        for term, (name, fn) in legs.items():
            s = _time_launches(fn, dev, 20)
            nbytes = cvb[term] * batch
            terms[term] = {"kernel": name, "bytes": int(nbytes), "us": round(s * 1e6, 3),
                           "GBps": round(nbytes / s / 1e9, 1), "frac": round(nbytes / s / 1e9 / HBM_PEAK_GBS, 4),
                           "traffic": PMC_TRAFFIC.get((term, batch, "f16") if half else (term, batch)) if level == "l0" else None}
            tot_b, tot_s = tot_b + nbytes, tot_s + s
        gbs = tot_b / tot_s / 1e9
        return {"kernel": "cost volume at %s, ELO_FUSED=0 path: encode1 + pool + encode2 + pool (A1+P1+A2+P2)" % level +
                          (", fp16 feature storage" if half else ""),
                "batch": batch, "dtype": "f16 storage, f32 arithmetic" if half else "f32", "bound": "hbm", "achieved": round(gbs, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(gbs / HBM_PEAK_GBS, 5), "bytes_per_launch": int(tot_b), "us_per_launch": round(tot_s * 1e6, 3),
                "terms": terms}

    def all_levels_leg(batch, half=False):
        """SURVEY 8(d)'s "cost-volume kernel bytes per pair": the four cost_volume calls of a forward (l0, l1, l2,
        l2_origin), 16 launches, bytes x batch / the sum of the kernel times."""
        per = {lv: per_operator_leg(batch, half, lv) for lv in LEVELS}
        tot_b = sum(v["bytes_per_launch"] for v in per.values())
        tot_us = sum(v["us_per_launch"] for v in per.values())
        gbs = tot_b / tot_us / 1e3
        return {"kernel": "all four cost_volume calls of a forward, ELO_FUSED=0 path (16 launches)", "batch": batch,
                "dtype": "f16 storage, f32 arithmetic" if half else "f32", "bound": "hbm", "achieved": round(gbs, 2),
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 5), "bytes": int(tot_b),
                "us": round(tot_us, 3),
                "levels": {lv: {"bytes": v["bytes_per_launch"], "us": v["us_per_launch"], "frac": v["frac"]} for lv, v in per.items()}}

    return {"bound": "mfma", "kernel": "cv1_kernel (fused cost volume stage 1, l0: %dx%d, K=%d, batch %d)" % (H, W, Kq, B),
            "achieved": round(tfs, 3), "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": round(tfs / FP32_MFMA_PEAK_TFLOPS, 5), "traffic": PMC_TRAFFIC.get(("cv1", B)),
            "mfma": "v_mfma_f32_16x16x16_f16 x3 per 16-k block (fp16 hi+lo split operands, fp32 accumulate)",
            "executed_mfma_TFLOPs": round(MFMA_PRODUCTS * tfs, 3), "fp16_mfma_peak": F16_MFMA_PEAK_TFLOPS,
            "frac_of_fp16_mfma_peak": round(MFMA_PRODUCTS * tfs / F16_MFMA_PEAK_TFLOPS, 5),
            "flops_per_launch": int(flops), "executed_mfma_flops_per_launch": int(MFMA_PRODUCTS * flops),
            "us_per_launch": round(sec * 1e6, 3),
            "algorithmic_bytes_per_launch": int((cvb["A1"] + cvb["P1"]) * B),
            "hbm_equivalent_GBps": round((cvb["A1"] + cvb["P1"]) * B / sec / 1e9, 2),
            "per_operator": per_operator_leg(B), "per_operator_b8": per_operator_leg(8), "per_operator_b64": per_operator_leg(64),
            "per_operator_b8_f16": per_operator_leg(8, half=True), "per_operator_b64_f16": per_operator_leg(64, half=True),
            "per_operator_all_levels_b8": all_levels_leg(8), "per_operator_all_levels_b8_f16": all_levels_leg(8, half=True),
            "per_operator_all_levels_b64": all_levels_leg(64)}


# ----------------------------------------------------------------------------- CPU baseline leg
def cpu_baseline_leg(args, net, pairs):
    """The numpy/C restatement of the SAME forward (oracle/, kind 'port'), one core, on `pairs` pairs."""
    from threadpoolctl import threadpool_limits

    from oracle import ops_np as O
    synth = pkg("synth")
    params = {k: v.detach().cpu().numpy() for k, v in net.store.state_dict().items()}
    rng_perm = {}

    def shuffle(scope, tag, KT):
        key = (scope, tag, KT)
        if key not in rng_perm:
            rng_perm[key] = np.random.default_rng(len(rng_perm)).permutation(KT).astype(np.int32)
        return rng_perm[key]

    f1, f2 = synth.frame_pair(1, args.height, args.width, seed=900)
    with threadpool_limits(limits=1):
        t0 = time.perf_counter()
        for _ in range(pairs):
            O.get_model_from_projection(params, shuffle, f1, f2)
        sec = time.perf_counter() - t0
    return {"value": round(pairs / sec, 4), "unit": "frame-pairs/s", "cores": 1, "kind": "port",
            "sample": "%d x one %dx%d pair through oracle/ops_np.get_model_from_projection (numpy fp32 + C "
                      "grouping oracle), single thread, %.1f s" % (pairs, args.height, args.width, sec)}


def batch_rate(dev, B, H, W, lanes, products, steps=240):
    """frame-pairs/s of a fresh net at batch B through `lanes` captured graphs, inputs resident in HBM."""
    model, synth, fused = pkg("model"), pkg("synth"), pkg("fused")
    with fused.products(products):
        net = model.PWCLONet(dev, seed=0)
        pairs = []
        for i in range(4):
            f1, f2 = synth.frame_pair(B, H, W, seed=77 + i)
            pairs.append(torch.cat([torch.from_numpy(f1), torch.from_numpy(f2)], 0).to(dev))
        net.capture(B, H, W, lanes=lanes)
        for i in range(2 * lanes):
            net.submit(i % lanes, pairs[i % 4])
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for i in range(steps):
            net.submit(i % lanes, pairs[i % 4])
        torch.cuda.synchronize(dev)
        return round(B * steps / (time.perf_counter() - t0), 1)


def raw_cloud_rate(dev, B, H, W, lanes, points=150000, steps=400):
    """frame-pairs/s when a step starts from RAW clouds: 2 x `points` KITTI-shaped points per pair
    (5 % zero padding, ranges to 60 m so the 35 m crop bites) -> elo_input_stage -> the pyramid, all inside the lane's
    graph; clouds resident in HBM."""
    model = pkg("model")
    net = model.PWCLONet(dev, seed=0)
    rng = np.random.default_rng(3)
    clouds = []
    for _ in range(4):
        az = rng.uniform(-np.pi, np.pi, (B, 2 * points))
        el = np.deg2rad(rng.uniform(-24.8, 2.0, (B, 2 * points)))
        r = rng.uniform(2.0, 60.0, (B, 2 * points))
        c = np.stack([r * np.cos(el) * np.cos(az), r * np.cos(el) * np.sin(az), r * np.sin(el)], -1).astype(np.float32)
        c[rng.random((B, 2 * points)) < 0.05] = 0
        clouds.append(torch.from_numpy(c).to(dev))
    net.capture(B, H, W, lanes=lanes, num_points=points)
    for i in range(2 * lanes):
        net.submit_points(i % lanes, clouds[i % 4])
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for i in range(steps):
        net.submit_points(i % lanes, clouds[i % 4])
    torch.cuda.synchronize(dev)
    return round(B * steps / (time.perf_counter() - t0), 1)


# ----------------------------------------------------------------------------- main
def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an AMD GPU: the hot path has no CPU fallback")
    local %= torch.cuda.device_count()               # (only differs when a test runs several ranks on one GPU)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("ELO_BENCH_BACKEND", "nccl")                            # nccl == RCCL on ROCm
        if backend == "nccl":                        # (gloo: the 2-ranks-on-1-GPU rehearsal of this code path)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    model, synth = pkg("model"), pkg("synth")
    pkg("fused").products(args.products).__enter__()          # for the whole run: packing, capture and the roofline leg
    net = model.PWCLONet(dev, seed=0)
    B, H, W = args.batch, args.height, args.width
    pool = []
    for i in range(args.pool):                       # inputs resident in HBM before the timed region
        f1, f2 = synth.frame_pair(B, H, W, seed=1000 * rank + 10 * i)
        pool.append(torch.cat([torch.from_numpy(f1), torch.from_numpy(f2)], 0).to(dev))   # (2B,H,W,3) = [f1 | f2]
    lanes = 1 if args.no_graph else max(1, args.lanes)
    if args.no_graph:
        net.forward(pool[0][:B], pool[0][B:])
    else:
        net.capture(B, H, W, lanes=lanes)
    # every step's l0 pose [q | t] is kept (a lane's static outputs are overwritten `lanes` steps later): one (B,7)
    # row block per step in HBM; with N > 1 ranks the blocks are all-gathered ONCE, inside the timed region
    # (7 floats per pair: the only exchange of the inference path, main.py:557-572)
    pose_log = torch.empty((max(args.steps, args.warmup), B, 7), device=dev)
    gathered = [torch.empty_like(pose_log[:args.steps]) for _ in range(world)] if world > 1 else None

    def step(i):
        pair = pool[i % len(pool)]
        if args.no_graph:                            # the l0 pose-head kernel writes the log row itself
            return net.forward(pair[:B], pair[B:], pose_out=pose_log[i])
        lane = i % lanes                             # step i rides lane i % lanes; lanes overlap on the GPU
        out = net.submit(lane, pair)                 # one copy in (the stacked pair), graph replay ...
        with torch.cuda.stream(net.lane_stream(lane)):
            pose_log[i].copy_(net.lane_pose(lane), non_blocking=True)      # ... one copy out (B,7)
        return out

    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.all_gather(gathered, pose_log[:args.steps].contiguous())     # untimed: RCCL sets its channels up on first use
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for i in range(args.steps):
        out = step(i)
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.all_gather(gathered, pose_log[:args.steps].contiguous())
        torch.cuda.synchronize(dev)
        dist.barrier()
    elapsed = time.perf_counter() - t0
    assert torch.isfinite(pose_log[:args.steps]).all()
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    line = {
        "metric": "frame-pairs/sec (KITTI 64x1800 range image)" if (H, W) == (64, 1800)
                  else "frame-pairs/sec (%dx%d range image)" % (H, W),
        "value": round(world * B * args.steps / elapsed, 3),
        "unit": "frame-pairs/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32 (1x1 convolutions as 3 fp16-MFMA products of hi/lo-split fp32 operands, fp32 accumulate)"
                 if args.products == "split" else
                 "f16 products (--products half: 1x1 convolutions as ONE fp16-MFMA product of fp16-rounded operands, fp32 "
                 "accumulate; storage and everything else fp32) -- not the fp32 parity path",
        "data": "synthetic",
        "config": {"workload": "full 4-level PWC pyramid (set-conv + attentive cost volume + warp-refinement + "
                               "set-upconv), %dx%d range-image pairs, batch %d per GPU, fp32, random-init weights, "
                               "%s" % (H, W, B, "eager launches" if args.no_graph else
                                      "hipGraph replay, %d forwards in flight" % lanes),
                   "global_batch": B * world, "parallelism": "dp%d" % world},
    }
    if rank == 0 and world == 1 and not args.no_legs:
        line["roofline"] = roofline_leg(args, dev, net)
        if not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline_leg(args, net, args.cpu_pairs)
        if (B, H, W) == (1, 64, 1800) and not args.no_graph and args.products == "split":
            # BASELINE configs[2]'s batch (8 pairs per step), same pyramid: fp32-class products, and fp16 products
            line["from_raw_clouds"] = {"unit": "frame-pairs/s", "lanes": lanes, "points_per_frame": 150000,
                                       "value": raw_cloud_rate(dev, B, H, W, lanes)}
            line["batch8"] = {"unit": "frame-pairs/s", "lanes": lanes,
                              "f32": batch_rate(dev, 8, H, W, lanes, "split"),
                              "f16_products": batch_rate(dev, 8, H, W, lanes, "half")}
    if rank == 0:
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
