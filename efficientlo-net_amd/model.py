"""PWCLONet: the object a user holds -- variables + visiting orders + (optionally)
a captured HIP graph of the whole forward pass.

The reference builds a TF graph once (main.py:141-179) and calls sess.run per
batch.  The MI355X-native equivalent of "build once, run many" is a hipGraph:
one frame pair's forward is ~600 small kernels, so at batch 1 the GPU is
launch-bound unless the launches are replayed from a graph.  `capture()`
records get_model_from_projection on static input buffers; `__call__` copies the
new range images in and replays.
"""
import time

import torch

from . import _ops, fused, model_util, perm, pwclo_model, tf_util, tuning


def graph_capture(graph):
    """torch.cuda.graph(graph) -- in the capture mode a process with a process group needs.  torch captures in hipStreamCaptureModeGlobal
    by default: while a capture is open, a hipEventQuery from ANY thread fails -- and the RCCL watchdog thread of torch.distributed
    polls the events of the collectives still in flight (the pose all-gather of the timed loop, a gradient all-reduce) all the time:
    "Exception raised from query at HIPEvent.h" and an abort, found by the one-rank RCCL rehearsal of round 6.  With a process group
    initialised the capture is therefore thread-local (only the capturing thread is held to the capture rules)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return torch.cuda.graph(graph, capture_error_mode="thread_local")
    return torch.cuda.graph(graph)


def distinct_queue_streams(device, want, candidates=16, cycles=500_000):
    """Up to `want` torch streams that sit on DIFFERENT hardware queues.

    HIP multiplexes its streams onto GPU_MAX_HW_QUEUES (default 4) hardware queues, assigned at stream
    creation; two streams on one queue never overlap, so lanes must be spread over the queues by
    measurement, not by counting streams (measured on MI355X, 64x1800, B=1: 12 lanes on whatever streams the
    pool hands out 3990 pairs/s, 4 lanes on 4 distinct queues 4730; 5 queues and more are slower again).
    The probe runs pairs of ~0.2 ms single-thread spin kernels: a pair on one queue takes twice as long."""
    dev = torch.device(device)
    spin = getattr(torch.cuda, "_sleep", None)
    pool = [torch.cuda.Stream(device=dev) for _ in range(max(candidates, want))]
    if spin is None or want <= 1:
        return pool[:want]

    def timed(streams):
        best = float("inf")
        for _ in range(2):
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for s in streams:
                with torch.cuda.stream(s):
                    spin(cycles)
            torch.cuda.synchronize(dev)
            best = min(best, time.perf_counter() - t0)
        return best

    timed(pool)                                     # first use of a stream creates its queue binding
    single = timed(pool[:1])
    chosen = []
    for s in pool:
        if all(timed([rep, s]) < 1.5 * single for rep in chosen):
            chosen.append(s)
            if len(chosen) == want:
                break
    return chosen


def _cached_tensors():
    """Every tensor the host-side caches hold right now (strided / all-pixel index grids, centre tables, decoded
    visiting orders): what a graph captured just before may point at."""
    from . import pointnet_util
    keep = list(model_util._sel_cache.values()) + list(pointnet_util._hw_cache.values()) + \
        list(pointnet_util._centre_hw_cache.values()) + list(fused._DECODED.values()) + list(fused._HW.values())
    return keep


class PWCLONet:
    def __init__(self, device="cuda:0", seed=0, perm_source=None, feature_dtype=torch.float32):
        """feature_dtype=torch.float16: fp16 feature STORAGE in HBM between the fused kernels (BASELINE configs[2]);
        geometry, weights and arithmetic are unchanged (fused inference path only)."""
        self.device = torch.device(device)
        self.feature_dtype = feature_dtype
        self.store = tf_util.VariableStore(self.device, seed=seed)
        self.perms = perm_source if perm_source is not None else perm.PermSource(seed=seed)
        self._graph = None
        self._static_in = None
        self._static_out = None
        self._lanes = []
        self._captured_at = None          # (store.generation, perms.generation) the graphs were recorded under

    # -- eager ---------------------------------------------------------------
    def forward(self, xyz_f1_proj, xyz_f2_proj, is_training=False, bn_decay=None, pose_out=None):
        """get_model_from_projection under this net's variables and permutations."""
        with tf_util.default_store(self.store), perm.default_perm_source(self.perms), fused.storage(self.feature_dtype):
            if is_training:
                if self.feature_dtype != torch.float32:
                    raise NotImplementedError("training stores its features in fp32")
                return pwclo_model.get_model_from_projection(xyz_f1_proj, xyz_f2_proj, True, bn_decay, pose_out)
            with torch.no_grad():
                return pwclo_model.get_model_from_projection(xyz_f1_proj, xyz_f2_proj, False, bn_decay, pose_out)

    def forward_points(self, point_cloud, H_input, W_input, T_gt, T_trans, T_trans_inv, is_training=False,
                       bn_decay=None, aug_frame=None):
        """get_model with the reference's full signature (raw clouds in)."""
        with tf_util.default_store(self.store), perm.default_perm_source(self.perms), fused.storage(self.feature_dtype):
            ctx = torch.enable_grad() if is_training else torch.no_grad()
            with ctx:
                return pwclo_model.get_model(point_cloud, H_input, W_input, T_gt, T_trans, T_trans_inv, is_training,
                                             bn_decay, aug_frame)

    def check_range(self, xyz_f1_proj, xyz_f2_proj):
        """One eager forward on the CHECKED instances of the fused kernels (include/elo.h elo_range_check): the number of
        matrix-core operands -- gathered inputs and layer outputs -- at or beyond the fp16 range (|x| >= 65504, or NaN),
        where the hi/lo split saturates instead of representing the value.  0 for any sane checkpoint and scan; capture()
        runs it on its `sample` so that a captured graph (which replays the unchecked kernels) was vetted on real data."""
        from . import _lib
        prev = _lib.range_check(True)
        try:
            _lib.range_violations(xyz_f1_proj)                       # reset the counter
            self.forward(xyz_f1_proj, xyz_f2_proj)
            return _lib.range_violations(xyz_f1_proj)
        finally:
            _lib.range_check(bool(prev))

    # -- HIP graph -----------------------------------------------------------
    def capture(self, batch_size, H_input, W_input, warmup=3, lanes=1, num_points=None, point_stride=3, pose_ring=0, sample=None,
                fresh_orders=0, check_every=0):
        """Record the inference forward into `lanes` independent hipGraphs (torch.cuda.CUDAGraph on ROCm).
        With `num_points` the graph starts from RAW clouds: a lane owns a (B, 2*num_points, point_stride) cloud buffer
        and records the input stage (model_util.input_stage: 35 m crop + both projections, no augmentation) in front
        of the pyramid; feed it with `submit_points`.

        One frame pair keeps only a few of the 256 CUs busy per kernel, and frame pairs are independent,
        so several forwards can be in flight: lane i owns a graph and its static input / output buffers and
        replays on one of the streams `distinct_queue_streams` found (one per hardware queue; lanes beyond
        the number of queues share streams round-robin, so use a multiple of the queue count, 4: with 6 lanes two
        queues carry twice the work of the others, 5210 instead of 6030 pairs/s); the weights are shared.
        `lanes=1` is the plain single-stream replay.

        `sample`: a stacked (2B,H,W,3) pair of real range images: check_range() runs on it first and capture refuses weights
        / inputs whose operands leave the fp16 range of the hi/lo split (the replayed kernels do not check).

        `fresh_orders=R` (>= 2): every REPLAY walks its own window visiting orders, as every sess.run of the reference does
        (tf.random_shuffle inside each operator, utils/pointnet_util.py:45,104,193,270): the order tensors become slices of
        one flat buffer per lane, R versions are pre-drawn, and the LAST launch of the lane's graph (the l0 pose head)
        copies the next version in and decodes it for the replay that follows (perm.PermSource.enable_pool;
        elo_pose_head_args.next_orders): no launch of its own.  Replay n of a lane (n = 1, 2, ...) walks version
        (n - 1) % R: `perms.pooled_version(n - 1)`.  0: one fixed draw for the life of the graph.

        `check_every=N` (>= 1): the production guard of the fp16 hi/lo split.  Every lane records a SECOND graph of the same
        forward on the CHECKED kernel instances (elo_range_check: every matrix-core operand -- gathered inputs and every
        layer output -- is compared with the fp16 range on its way into a quad; the tile kernels, bit-identical results) and
        every N-th replay of a lane takes that graph; `range_violations()` / `collect()` read the device counter and raise.
        A stream of scans is therefore vetted continuously at 1/N of the checked kernels' extra cost (N = 1: every replay)
        instead of once on the capture `sample`; a single out-of-range scan between two checked replays can go unseen.

        `pose_ring=R` (>= 2): a lane's pose output is a ring of R rows blocks instead of one (B,7) block -- replay r of the
        lane writes slot r % R (the l0 pose-head kernel keeps the cursor on the device), so a stream of pairs is not
        followed by one copy-out launch per pair: `lane_poses(lane)` returns the rows written since `reset_poses(lane)`."""
        dev = self.device
        if sample is not None:                       # a representative (2B,H,W,3) pair: vet the operand ranges on it
            bad = self.check_range(sample[:batch_size], sample[batch_size:])
            if bad:
                raise RuntimeError("%d matrix-core operands of this forward lie at or beyond the fp16 range (|x| >= 65504): the "
                                   "fused kernels' hi/lo split would saturate them -- rescale the inputs / weights or run the "
                                   "fp32-MFMA build (ELO_DENSE_F32=1)" % bad)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        def zeros():          # both frames in one allocation: the Siamese pyramid then runs as one 2B batch
            return torch.zeros((2 * batch_size, H_input, W_input, 3), device=dev)
        with torch.cuda.stream(side):
            probe = zeros()
            for _ in range(warmup):                 # creates variables, folded weights, caches, hipBLASLt plans
                self.forward(probe[:batch_size], probe[batch_size:])
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        self._lanes = []
        if fresh_orders:
            self.perms.enable_pool(int(fresh_orders), lanes, dev)       # (the warm-up forwards above created every order tensor)
        streams = distinct_queue_streams(dev, lanes)
        if num_points is not None:                   # warm the input stage's allocations up as well
            with torch.cuda.stream(side):
                model_util.input_stage(torch.zeros((batch_size, 2 * num_points, point_stride), device=dev), None, None,
                                       H_input, W_input)
            torch.cuda.synchronize(dev)
        for i in range(lanes):
            both = zeros()
            lane = {"stream": streams[i % len(streams)], "pair": both, "in": (both[:batch_size], both[batch_size:]),
                    "graph": torch.cuda.CUDAGraph(), "replays": 0,
                    "pose": _ops.PoseRing(pose_ring, batch_size, dev) if pose_ring else torch.zeros((batch_size, 7), device=dev)}
            lane["order"] = torch.cuda.Event()        # submit()'s producer ordering: recorded on the caller's stream, waited on by the lane's
            with torch.cuda.stream(side):
                lane["order"].record()                # (materialises the hipEvent_t: elo_graph_submit gets the raw handle)
            if num_points is not None:
                lane["cloud"] = torch.zeros((batch_size, 2 * num_points, point_stride), device=dev)
            if fresh_orders:                          # this lane's order buffers; caches keyed on them filled before the capture
                self.perms.active_lane, self.perms.tail_armed = i, False
                with torch.cuda.stream(side):
                    self.forward(both[:batch_size], both[batch_size:])
                torch.cuda.synchronize(dev)
            self.perms.tail_armed = bool(fresh_orders)    # the recorded forward's last launch loads the NEXT replay's orders
            with graph_capture(lane["graph"]):
                if num_points is not None:
                    _pts, staged = model_util.input_stage(lane["cloud"], None, None, H_input, W_input)
                    lane["out"] = self.forward(staged[:batch_size], staged[batch_size:], pose_out=lane["pose"])
                else:
                    lane["out"] = self.forward(*lane["in"], pose_out=lane["pose"])
            if check_every:                           # the same forward on the checked kernel instances, same buffers
                from . import _lib
                lane["graph_checked"], lane["check_every"] = torch.cuda.CUDAGraph(), int(check_every)
                # the lane's OWN violation word: its address goes into the kernel arguments of the checked graph (ABI 26), so a
                # saturated operand is counted for the lane whose forward met it -- not for every lane, as one process-wide word did
                lane["range_counter"] = torch.zeros((1,), dtype=torch.int64, device=dev)
                prev = _lib.range_check(True)
                prev_counter = _lib.set_range_counter(lane["range_counter"].data_ptr())
                try:
                    self.perms.tail_armed = False     # (the warm-up below must not advance the lane's order cursor)
                    with torch.cuda.stream(side):     # the checked path's own allocations / caches, before its capture
                        self.forward(*lane["in"])
                    torch.cuda.synchronize(dev)
                    self.perms.tail_armed = bool(fresh_orders)
                    with graph_capture(lane["graph_checked"]):
                        if num_points is not None:
                            _pts, staged = model_util.input_stage(lane["cloud"], None, None, H_input, W_input)
                            lane["out_checked"] = self.forward(staged[:batch_size], staged[batch_size:], pose_out=lane["pose"])
                        else:
                            lane["out_checked"] = self.forward(*lane["in"], pose_out=lane["pose"])
                finally:
                    _lib.range_check(bool(prev))
                    _lib.set_range_counter(prev_counter)
                lane["range_counter"].zero_()         # (the warm-up forward counted too)
            # the graph holds raw device pointers into the module-level index / decoded-order caches; those caches evict
            # (clear()) when they grow: the lane keeps the tensors alive for as long as its graph exists
            lane["keep"] = _cached_tensors()
            lane["native"] = self._native_submit(lane, dev)
            self._lanes.append(lane)
        self.perms.active_lane, self.perms.tail_armed = 0, False
        torch.cuda.synchronize(dev)
        self._graph, self._static_in, self._static_out = (self._lanes[0]["graph"], self._lanes[0]["in"],
                                                          self._lanes[0]["out"])
        self._captured_at = (self.store.generation, self.perms.generation, tuning.digest())
        self._tuning_seen = tuning.version()
        self.captured_tuning = tuning.snapshot()           # the forms this graph has baked in (bench.py: config.tuning)
        return self

    @staticmethod
    def _native_submit(lane, dev):
        """What submit() hands to elo_graph_submit for this lane (tuning native_submit), or None: torch's own copy_ + replay().  Needs the
        graphs' raw exec handles (torch >= 2.8).  The lane's device index goes along: the native call makes it current for its duration when
        the submitting thread's current device is another one (one process per GPU is the design; a second thread need not set_device).
        Restriction: a graph launched this way skips CUDAGraph.replay()'s generator prologue -- the inference forward draws no torch random
        numbers (dropout is off, visiting orders are explicit inputs: perm.PermSource), so nothing is registered."""
        if not tuning.get("native_submit"):
            return None
        from . import _lib
        try:
            execs = [lane[k].raw_cuda_graph_exec() if k in lane else None for k in ("graph", "graph_checked")]
        except (AttributeError, RuntimeError):
            return None
        import ctypes
        pair = lane["pair"]
        return {"submit": _lib.lib().elo_graph_submit, "exec": ctypes.c_void_p(execs[0]), "exec_checked": ctypes.c_void_p(execs[1] or execs[0]),
                "stream": ctypes.c_void_p(lane["stream"].cuda_stream), "dst": ctypes.c_void_p(pair.data_ptr()),
                "nbytes": pair.numel() * pair.element_size(), "event": ctypes.c_void_p(lane["order"].cuda_event), "device": int(dev.index)}

    def _check_fresh(self):
        """A captured graph holds raw device pointers to the folded / packed inference weights and to the decoded
        visiting orders.  VariableStore.invalidate() (load_state_dict, tf_checkpoint.load_into, a training step) and
        PermSource.reshuffle() drop those tensors: replaying would read stale weights or recycled memory, silently."""
        if self._graph is None:
            raise RuntimeError("no captured graph: call capture() first")
        if self._captured_at[:2] != (self.store.generation, self.perms.generation):
            raise RuntimeError("the captured graph is stale: the variables or the visiting orders changed after capture() "
                               "(checkpoint load, training step or reshuffle) -- call capture() again")
        if self._tuning_seen != tuning.version():          # (re-hash only when something was changed: this sits on every submit)
            if self._captured_at[2] == tuning.digest():
                self._tuning_seen = tuning.version()
                return
            raise RuntimeError("the captured graph is stale: the tuning changed after capture() (a graph keeps the kernel forms of "
                               "its capture: tuning.py / elo_set_tuning) -- call capture() again; captured under "
                               "%s, now %s" % (self.captured_tuning, tuning.snapshot()))

    def load_inputs(self, xyz_f1_proj, xyz_f2_proj):
        self._static_in[0].copy_(xyz_f1_proj, non_blocking=True)
        self._static_in[1].copy_(xyz_f2_proj, non_blocking=True)

    def replay(self):
        self._check_fresh()
        return self._replay_lane(self._lanes[0])   # (lane 0's graph advances its device-side pose-ring cursor)

    def _replay_lane(self, lane):
        """Replay the lane's graph -- every `check_every`-th time the one recorded on the range-checked kernels."""
        n = lane.get("check_every", 0)
        lane["total"] = lane.get("total", 0) + 1
        checked = n and lane["total"] % n == 0
        (lane["graph_checked"] if checked else lane["graph"]).replay()
        lane["replays"] += 1
        return lane["out_checked"] if checked else lane["out"]

    def range_violations(self, lane_index=None):
        """Matrix-core operands at or beyond the fp16 range (|x| >= 65504, or NaN) seen by the checked replays since the last
        call (capture(..., check_every=N)): of one lane, or (None) of all lanes together.  Every lane has its own device word,
        written by its own checked graph only; reading it synchronises with the current stream -- call it once the lane's stream
        has been synchronised or waited on.  The count is remembered per lane until that lane's collect() raises it."""
        lanes = self._lanes if lane_index is None else [self._lanes[lane_index]]
        total = 0
        for lane in lanes:
            word = lane.get("range_counter")
            if word is None:
                continue
            bad = int(word.item())
            if bad:
                word.zero_()
                lane["tainted"] = lane.get("tainted", 0) + bad
                total += bad
        return total

    def collect(self, lane_index):
        """lane_poses(lane_index) for a lane whose work is DONE: synchronises the lane's stream, and raises if a checked
        replay saw an operand outside the fp16 range (the poses since the last collection are then not to be trusted)."""
        lane = self._lanes[lane_index]
        lane["stream"].synchronize()
        if lane.get("check_every"):
            self.range_violations(lane_index)        # this lane's own word
            bad = lane.pop("tainted", 0)
            if bad:
                raise RuntimeError("%d matrix-core operands at or beyond the fp16 range (|x| >= 65504 or NaN) since the last "
                                   "collection: the hi/lo split of the fused kernels saturated them -- rescale the inputs / "
                                   "weights or run the fp32-MFMA build (ELO_DENSE_F32=1)" % bad)
        return self.lane_poses(lane_index) if isinstance(lane["pose"], _ops.PoseRing) else self.lane_pose(lane_index)

    def lane_input(self, lane_index):
        """The lane's input buffer, (2B,H,W,3) = [frame 1 | frame 2]: a producer (a data loader, elo_input_stage, the previous
        stage of a pipeline) that writes its range images HERE -- on the lane's stream, or ordered before the submit -- needs no
        copy: submit(lane_index) then replays on what the buffer holds."""
        self._check_fresh()
        return self._lanes[lane_index]["pair"]

    def _order_lane(self, lane, ready, *inputs):
        """submit()'s producer ordering on the torch path: the lane's stream waits for `ready` (an event the caller recorded behind
        the producer) or, by default, for everything enqueued so far on the CURRENT stream; inputs are marked as in use on the lane's
        stream (record_stream: the caching allocator must not hand their memory out before the lane's copy has run)."""
        if ready is False:
            return
        stream = lane["stream"]
        if ready is None or ready is True:
            cur = torch.cuda.current_stream(self.device)
            if cur != stream and not cur.query():         # (an idle producer has nothing to wait for: no cross-queue barrier)
                lane["order"].record(cur)
                stream.wait_event(lane["order"])
        else:
            stream.wait_event(ready)
        for x in inputs:
            if x is not None and x.is_cuda:            # (a host tensor is copied by the lane's own H2D copy: nothing to mark)
                x.record_stream(stream)

    def submit(self, lane_index, xyz_f1_proj=None, xyz_f2_proj=None, ready=None):
        """Enqueue one forward on lane `lane_index` (its own stream); returns the lane's static outputs,
        valid once that stream has been synchronised (or waited on).  With `xyz_f2_proj` None the first
        argument is the stacked pair (2B,H,W,3) = [frame 1 | frame 2]: one copy instead of two; with both None the lane's
        input buffer was written in place (lane_input): no copy.

        ORDERING (the reference's sess.run(feed_dict=...) is synchronous, main.py:372-381; a lane is not): `ready=None` (default)
        -- the lane's stream first waits for everything enqueued so far on the caller's CURRENT stream, so a pair uploaded
        or computed there (or a lane_input() written in place there) is complete before the lane reads it, and the inputs are
        record_stream()-ed on the lane's stream; `ready=<torch.cuda.Event>` -- waits for that event instead (a producer on
        some other stream); `ready=False` -- no ordering and no record_stream: the caller owns both (inputs resident and
        synchronised -- bench.py's pool -- or produced on the lane's own stream)."""
        self._check_fresh()
        lane = self._lanes[lane_index]
        native = lane.get("native")
        if native is not None and xyz_f2_proj is None and (ready is None or isinstance(ready, bool)):
            # the host runtime's own submit (csrc/elo_host.cpp elo_graph_submit): the ordering, the copy and the graph launch as ONE
            # native call on the lane's stream -- 16 us of host time instead of 29 through torch (tools/submit_native_probe.py)
            pair = lane["pair"]
            if xyz_f1_proj is None:
                src, nbytes = None, 0
            else:
                if (xyz_f1_proj.shape != pair.shape or xyz_f1_proj.dtype != pair.dtype or xyz_f1_proj.device != pair.device
                        or not xyz_f1_proj.is_contiguous()):
                    native = None                     # (a conversion is needed: torch's copy_ below does it)
                else:
                    src, nbytes = xyz_f1_proj.data_ptr(), native["nbytes"]
            if native is not None:
                n = lane.get("check_every", 0)
                lane["total"] = lane.get("total", 0) + 1
                checked = n and lane["total"] % n == 0
                from . import _lib
                if ready is False:
                    producer = event = None
                else:
                    producer, event = torch.cuda.current_stream(self.device).cuda_stream, native["event"]
                    if xyz_f1_proj is not None:
                        xyz_f1_proj.record_stream(lane["stream"])
                _lib.check(native["submit"](native["exec_checked"] if checked else native["exec"], native["stream"], native["dst"], src, nbytes,
                                            producer, event, native["device"]))
                lane["replays"] += 1
                return lane["out_checked"] if checked else lane["out"]
        self._order_lane(lane, ready, xyz_f1_proj, xyz_f2_proj)
        with torch.cuda.stream(lane["stream"]):
            if xyz_f1_proj is None:
                pass
            elif xyz_f2_proj is None:
                lane["pair"].copy_(xyz_f1_proj, non_blocking=True)
            else:
                lane["in"][0].copy_(xyz_f1_proj, non_blocking=True)
                lane["in"][1].copy_(xyz_f2_proj, non_blocking=True)
            out = self._replay_lane(lane)
        return out

    def submit_points(self, lane_index, point_cloud, ready=None):
        """Enqueue one forward from raw clouds (B, 2N, stride) on a lane captured with `num_points`.  `ready`: as submit()."""
        self._check_fresh()
        lane = self._lanes[lane_index]
        self._order_lane(lane, ready, point_cloud)
        with torch.cuda.stream(lane["stream"]):
            lane["cloud"].copy_(point_cloud, non_blocking=True)
            out = self._replay_lane(lane)
        return out

    def lane_pose(self, lane_index):
        """The lane's (B,7) [l0_q_norm | l0_t] block, written by the l0 pose-head kernel of its last replay
        (capture(..., pose_ring=R): the slot of the last replay)."""
        lane = self._lanes[lane_index]
        pose = lane["pose"]
        if isinstance(pose, _ops.PoseRing):
            return pose.rows[(lane.get("base", 0) + lane["replays"] - 1) % pose.slots]
        return pose

    def reset_poses(self, lane_index):
        """Pose ring of the lane back to slot 0 (one small launch, enqueued on the lane's stream)."""
        lane = self._lanes[lane_index]
        with torch.cuda.stream(lane["stream"]):
            lane["pose"].reset()
        lane["replays"], lane["base"] = 0, 0

    def mark_poses(self, lane_index):
        """Start a new collection WITHOUT touching the device: lane_poses() then returns the rows of the replays from here
        on (the device-side cursor keeps running; the host remembers which slot the next replay writes).  No launch -- a
        stream of short collections (bench.py's 20-step repeats) paid one reset launch per lane and collection before."""
        lane = self._lanes[lane_index]
        ring = lane["pose"]
        lane["base"] = (lane.get("base", 0) + lane["replays"]) % ring.slots
        lane["replays"] = 0

    def lane_poses(self, lane_index):
        """(n,B,7): the rows of the lane's replays since reset_poses (n <= R, oldest first); a view of the ring, valid
        once the lane's stream has been synchronised or waited on."""
        lane = self._lanes[lane_index]
        ring, n, base = lane["pose"], lane["replays"], lane.get("base", 0)
        if n > ring.slots:
            raise RuntimeError("%d replays since reset_poses() on a ring of %d slots: rows were overwritten" % (n, ring.slots))
        if base + n <= ring.slots:
            return ring.rows[base:base + n]
        return torch.cat([ring.rows[base:], ring.rows[:base + n - ring.slots]], 0)        # (wrapped: a copy, on the current stream)

    def lane_stream(self, lane_index):
        return self._lanes[lane_index]["stream"]

    def __call__(self, xyz_f1_proj, xyz_f2_proj):
        if self._graph is None:
            return self.forward(xyz_f1_proj, xyz_f2_proj)
        self.load_inputs(xyz_f1_proj, xyz_f2_proj)
        return self.replay()
