"""Host side of the fused inference kernels (csrc/elo_fused.hip, include/elo.h "Fused inference
kernels"): weight packing into MFMA B-fragment order and torch-tensor front-ends.

A packed layer is cached in the VariableStore next to the folded weights and dropped by
VariableStore.invalidate(); packing costs a few small torch ops once per parameter update."""
import ctypes
import os

import torch

from . import _lib as L
from . import tf_util, tuning


def fp32_mfma():
    """True when libelo_hip.so was built with -DELO_DENSE_F32 (the fp32 MFMA path kept for comparison): the packed
    weights then stay fp32.  Asked of the library itself so that host packing and kernels cannot disagree."""
    return bool(L.lib().elo_dense_f32())


PRODUCTS_SPLIT, PRODUCTS_HALF = 0, 1               # include/elo.h ELO_PRODUCTS_*
_products = PRODUCTS_SPLIT


def products_mode():
    return _products


class products:
    """`with fused.products("half"):` -- run the dense layers of the fused kernels as ONE fp16 product per 16-k block
    (operands rounded to nearest fp16, fp32 accumulation) instead of the fp32-class three-product split: fp16
    arithmetic for BASELINE configs[2].  The mode is read when a layer is packed and when a kernel is launched (a
    captured graph keeps the mode it was captured under)."""

    def __init__(self, mode):
        self.mode = {"split": PRODUCTS_SPLIT, "half": PRODUCTS_HALF, PRODUCTS_SPLIT: PRODUCTS_SPLIT,
                     PRODUCTS_HALF: PRODUCTS_HALF}[mode]
        if self.mode == PRODUCTS_HALF and fp32_mfma():
            raise RuntimeError("libelo_hip.so was built with -DELO_DENSE_F32: no fp16 products")

    def __enter__(self):
        global _products
        self.prev, _products = _products, self.mode
        return self

    def __exit__(self, *exc):
        global _products
        _products = self.prev


_storage = torch.float32


def storage_dtype():
    """Storage type of the feature tensors the fused path creates from scratch (the all-zero input features); every
    kernel front-end below then gives its outputs the dtype of its feature inputs."""
    return _storage


class storage:
    """`with fused.storage(torch.float16):` -- fp16 FEATURE STORAGE in HBM (BASELINE configs[2]; SURVEY.md 8(d): s = 2):
    every feature tensor between two fused kernels is fp16; geometry, indices, masks and all arithmetic stay as they are."""

    def __init__(self, dtype):
        if dtype not in (torch.float32, torch.float16):
            raise TypeError("feature storage is torch.float32 or torch.float16")
        self.dtype = dtype

    def __enter__(self):
        global _storage
        self.prev, _storage = _storage, self.dtype
        return self

    def __exit__(self, *exc):
        global _storage
        _storage = self.prev


def _features(*ts):
    """Contiguous feature tensors of ONE storage dtype (fp32 or fp16); returns (tensors, dtype, ELO_F32 / ELO_F16)."""
    live = [t for t in ts if t is not None]
    dt = live[0].dtype
    if dt not in (torch.float32, torch.float16) or any(t.dtype != dt for t in live):
        raise TypeError("the feature tensors of a call are all float32 or all float16 (got %s)" % [str(t.dtype) for t in live])
    return [None if t is None else t.contiguous() for t in ts], dt, L.dtype_code(live[0])


class PackedDense:
    """One inference layer y = act(x @ W + b) with BN folded, packed in MFMA fragment order (include/elo.h, csrc/elo_fused.hip
    WFrag).  Each weight is split into fp16 hi + lo (w = hi + lo to 2^-22 relative) for the three-product scheme.  Per
    column block the K axis is laid out as PAIRS of 32 k -- a lane's eight halves are W[32p + 4kq + j] (j = 0..3) followed
    by W[32p + 16 + 4kq + j]: 1 KiB of hi8 for all lanes, then 1 KiB of lo8 -- plus, for an odd number of 16-k blocks, a
    16-k TAIL ([hi4 | lo4] per lane).  With `half` only round-to-nearest halves are kept (single-product mode); a
    library built with -DELO_DENSE_F32 takes plain fp32 fragments (16-k blocks of four floats per lane)."""

    def __init__(self, W, b, relu=True, row_order=None, half=False):
        if row_order is not None:                         # the kernel's LDS column order differs from the concat order
            W = W[row_order]
        K, N = W.shape
        if not fp32_mfma():
            # the hi/lo halves are fp16: a folded weight at or beyond the fp16 range would become inf (and lo = w - inf)
            # at pack time.  Real checkpoints fold BN gains of up to ~31x (eps 1e-3) into weights of O(1): far away, but not
            # by construction -- so it is checked where it would happen (one host sync per packed layer, at pack time only)
            peak = float(W.abs().max()) if W.numel() else 0.0
            if not peak < 65504.0:
                raise ValueError("a folded weight of magnitude %g does not fit the fp16 hi/lo split of the fused kernels "
                                 "(|w| < 65504): rescale the layer or run the fp32-MFMA build (ELO_DENSE_F32=1)" % peak)
        Kp, Np = (K + 15) // 16 * 16, (N + 15) // 16 * 16
        Wp = torch.zeros((Kp, Np), dtype=torch.float32, device=W.device)
        Wp[:K, :N] = W
        CB, KS, NP = Np // 16, Kp // 16, Kp // 32
        # element (cb, ks, lane = kq*16 + n, s) = Wp[ks*16 + 4*kq + s][cb*16 + n], s = 0..3
        frag = Wp.reshape(KS, 4, 4, CB, 16).permute(3, 0, 1, 4, 2).reshape(CB, KS, 64, 4).contiguous()
        self.products = PRODUCTS_HALF if half else PRODUCTS_SPLIT

        def paired(h4):          # (CB, KS, 64, 4) halves -> per column block [pairs of 8 halves per lane ..., tail of 4]
            parts = []
            if NP:
                parts.append(torch.cat([h4[:, 0:2 * NP:2], h4[:, 1:2 * NP:2]], -1))        # (CB, NP, 64, 8)
            return parts, (h4[:, KS - 1] if KS % 2 else None)                                 # tail (CB, 64, 4)

        if fp32_mfma():
            self.w = frag
        elif half:
            pairs, tail = paired(frag.to(torch.float16))
            flat = [p.reshape(CB, -1) for p in pairs] + ([tail.reshape(CB, -1)] if tail is not None else [])
            self.w = torch.cat(flat, 1).contiguous()                                          # (CB, KS*256) halves
        else:
            hi = frag.to(torch.float16)
            lo = (frag - hi.to(torch.float32)).to(torch.float16)
            (ph, th), (pl, tl) = paired(hi), paired(lo)
            flat = []
            if ph:
                flat.append(torch.stack([ph[0], pl[0]], 2).reshape(CB, -1))                   # per pair: [hi8 x 64 | lo8 x 64]
            if th is not None:
                flat.append(torch.cat([th, tl], -1).reshape(CB, -1))                          # per lane [hi4 | lo4]
            self.w = torch.cat(flat, 1).contiguous()                                          # (CB, KS*512) halves
        self.b = torch.zeros((Np,), dtype=torch.float32, device=W.device)
        self.b[:N] = b
        self.K, self.N, self.relu = K, N, relu
        self.plain = W.contiguous() if max(K, N) <= 32 else None      # narrow layers: also kept row-major (K,N)

    def struct(self):
        return L.Dense(self.w.data_ptr(), self.b.data_ptr(), self.K, self.N, 1 if self.relu else 0,
                       self.plain.data_ptr() if self.plain is not None else None, self.products)


def packed_layer(scope, cin, cout, bn=True, relu=True, row_order=None, tf_kernel_dims=(1, 1)):
    """get-or-create the layer's variables under the active scope, fold, pack, cache."""
    store = tf_util.get_store()
    name, W, b, bn_vars = tf_util.dense_variables(scope, cin, cout, tf_kernel_dims, bn)
    half = _products == PRODUCTS_HALF
    key = ("packed", name, None if row_order is None else tuple(row_order), relu, half)
    hit = store._folded.get(key)
    if hit is None:
        Wf, bf = store.folded(name, W, b, bn_vars)
        order = None if row_order is None else torch.as_tensor(row_order, device=Wf.device)
        hit = PackedDense(Wf, bf, relu, order, half)
        store._folded[key] = hit
    return hit


def setconv_row_order(C):
    """Weight rows of a set-conv's first layer in the kernel's column order [features (C) | xyz difference (3)]
    (reference concat: [xyz_diff, features], utils/pointnet_util.py:213)."""
    return list(range(3, 3 + C)) + [0, 1, 2]


def cv0_row_order(C):
    """CV_0's rows in the kernel's order [feat1 (C) | feat2 grouped (C) | geometry (10)] (reference: [geometry, feat1,
    feat2], utils/pointnet_util.py:62-66)."""
    return list(range(10, 10 + 2 * C)) + list(range(10))


def stage2_row_order(w_before, n_out, w_after):
    """Second-stage rows of a two-stage row-wise MLP in the kernel's order [out | before | after] (reference concat:
    [before, out, after], utils/pointnet_util.py:161-166)."""
    return (list(range(w_before, w_before + n_out)) + list(range(w_before)) +
            list(range(w_before + n_out, w_before + n_out + w_after)))


def _f32c(t):
    if t.dtype != torch.float32:
        raise TypeError("geometry tensors (xyz, masks) are float32")
    return t.contiguous()


def _chain(layers):
    arr = (L.Dense * 3)()
    for i, p in enumerate(layers):
        arr[i] = p.struct()
    return arr


class Grouping:
    """In-kernel grouping spec (include/elo.h elo_group_spec): the fused kernel runs random-k / select-k itself.
    want_indices=True also returns the (B,N,K,3) indices and (B,N,K) mask the stand-alone op would produce."""

    def __init__(self, random_hw, kernel_size, distance, stride_h=1, stride_w=1, want_indices=False):
        self.random_hw = random_hw.contiguous()
        self.kernel_h, self.kernel_w = int(kernel_size[0]), int(kernel_size[1])
        self.distance, self.stride_h, self.stride_w = float(distance), int(stride_h), int(stride_w)
        self.want_indices = want_indices
        self.idx = self.mask = None

    def struct(self, B, N, K, device):
        if self.random_hw.dtype != torch.int32 or self.random_hw.numel() != self.kernel_h * self.kernel_w:
            raise ValueError("FusedConv expects (kernel_size_h * kernel_size_w) random_hw shape.")
        if self.want_indices:
            self.idx = torch.empty((B, N, K, 3), dtype=torch.int32, device=device)
            self.mask = torch.empty((B, N, K), dtype=torch.float32, device=device)
        return L.GroupSpec(self.random_hw.data_ptr(), self.kernel_h, self.kernel_w, self.distance, self.stride_h,
                           self.stride_w, self.idx.data_ptr() if self.want_indices else None,
                           self.mask.data_ptr() if self.want_indices else None,
                           _decoded_order(self.random_hw, self.kernel_h, self.kernel_w))


_DECODED = {}


def _decoded_order(random_hw, kH, kW):
    """The visiting order as packed (dh, dw) window offsets (elo_group_spec.decoded_hw), computed once per content of
    the order tensor (keyed by storage and in-place version counter: PermSource.reshuffle() invalidates it) instead of
    two integer divisions per slot in every tile.  Never computed while a graph is being captured (the few torch
    kernels would be recorded into it): the capture's warm-up forwards have filled the cache by then."""
    from . import perm
    pooled = perm.pooled_decoded(random_hw)              # fresh orders per replay: decoded on the device by the refresh launch
    if pooled is not None:
        return pooled.data_ptr()
    key = (random_hw.data_ptr(), kH, kW)
    hit = _DECODED.get(key)
    if hit is not None and hit[0] is random_hw and hit[1] == random_hw._version:
        return hit[2].data_ptr()
    if random_hw.is_cuda and torch.cuda.is_current_stream_capturing():
        return None
    p = random_hw.to(torch.int64)
    dec = ((((p // kW) - kH // 2) << 16) | (((p % kW) - kW // 2) & 0xffff)).to(torch.int32).contiguous()
    if len(_DECODED) >= 512:
        _DECODED.clear()
    _DECODED[key] = (random_hw, random_hw._version, dec)     # holds the order tensor: its address cannot be reused meanwhile
    return dec.data_ptr()


_NO_GROUP = L.GroupSpec(None, 0, 0, 0.0, 0, 0, None, None, None)


def _setconv_args(src_xyz, src_feat, idx, mask, layers, centre_xyz=None, xyz1_grid=None, centre_hw=None, group=None,
                  K=None):
    L.require_gpu(src_xyz, src_feat, idx, mask, centre_xyz, xyz1_grid, centre_hw)
    src_xyz = _f32c(src_xyz)
    (src_feat,), dt, code = _features(src_feat)
    _, H2, W2, C = src_feat.shape
    dev = src_xyz.device
    ptr = lambda x: x.data_ptr() if x is not None else None
    H = W = 0
    if xyz1_grid is not None:
        xyz1_grid = _f32c(xyz1_grid)
        H, W = xyz1_grid.shape[1:3]
    if group is None:
        idx, mask = idx.contiguous(), _f32c(mask)
        B, n, K, _ = idx.shape
    else:
        B = src_xyz.shape[0]
        n = centre_hw.shape[1] if centre_hw is not None else H * W
    new_xyz = None
    if centre_hw is not None:
        centre_hw = centre_hw.contiguous()
        new_xyz = torch.empty((B, n, 3), dtype=torch.float32, device=dev)
    elif centre_xyz is not None:
        centre_xyz = _f32c(centre_xyz)
    out = torch.empty((B, n, layers[-1].N), dtype=dt, device=dev)
    a = L.SetconvArgs(B, n, K, H, W, H2, W2, C, ptr(xyz1_grid), ptr(centre_hw), ptr(centre_xyz), src_xyz.data_ptr(),
                      src_feat.data_ptr(), ptr(idx), ptr(mask), len(layers), _chain(layers), out.data_ptr(), ptr(new_xyz),
                      group.struct(B, n, K, dev) if group is not None else _NO_GROUP, code)
    keep = (src_xyz, src_feat, idx, mask, centre_xyz, xyz1_grid, centre_hw, layers, group)   # alive until the launch
    return a, out, new_xyz, keep


def setconv(src_xyz, src_feat, idx, mask, layers, centre_xyz=None, xyz1_grid=None, centre_hw=None, group=None, K=None):
    """group_concat -> MLP chain -> masked max over K in one launch.  Either (idx, mask) from a grouping op, or
    group=Grouping(...) + K to group in-kernel (needs xyz1_grid; centre_hw None = every pixel of xyz1_grid).
    layers[0] is packed with `row_order=setconv_row_order(C)` (the kernel's columns are [features | xyz difference]).
    Returns (out (B,n,Cout) in src_feat's dtype, new_xyz (B,n,3) or None)."""
    job = _setconv_args(src_xyz, src_feat, idx, mask, layers, centre_xyz, xyz1_grid, centre_hw, group, K)
    L.call("elo_setconv_fused", job[0], job[1])
    return job[1], job[2]


def setconv_pair(job_a, job_b):
    """Two set-conv jobs of identical shape (dicts of setconv() keyword arguments) in ONE launch."""
    ja, jb = _setconv_args(**job_a), _setconv_args(**job_b)
    L.call2("elo_setconv_fused2", ja[0], jb[0], ja[1])
    return (ja[1], ja[2]), (jb[1], jb[2])


def _mlp_args(sources, layers):
    L.require_gpu(*sources)
    batch_hint = sources[0].shape[0] if sources[0].dim() >= 3 else 0      # (B, N, C) inputs: the launcher's regime switch
    srcs, dt, code = _features(*sources)
    srcs = [s.reshape(-1, s.shape[-1]) for s in srcs]
    rows = srcs[0].shape[0]
    out = torch.empty((rows, layers[-1].N), dtype=dt, device=srcs[0].device)
    a = L.MlpArgs()
    a.rows, a.n_sources, a.n_layers, a.layers, a.out = rows, len(srcs), len(layers), _chain(layers), out.data_ptr()
    a.feat_dtype, a.batch_hint = code, batch_hint
    for i, s in enumerate(srcs):
        a.src[i], a.src_width[i] = s.data_ptr(), s.shape[1]
    return a, out, (srcs, layers)


def _ride_clear(a, clear, dt):
    """elo_mlp_args.clear_*: the launch's workgroups also clear `clear` (an _ops.ProjectionBuffers)."""
    if clear is None:
        return
    if clear.out_feat is not None and clear.out_feat.dtype != dt:
        raise TypeError("the projection buffers and the MLP's features must share one storage dtype")
    B, _N, H, W, C = clear.shape
    a.clear_scratch, a.clear_xyz = clear.scratch.data_ptr(), clear.out_xyz.data_ptr()
    a.clear_feat = clear.out_feat.data_ptr() if clear.out_feat is not None else None
    a.clear_cells, a.clear_C, a.clear_images = B * H * W, C, B
    clear.cleared = True


def _ride_sv(a, b, sv, dt):
    """elo_mlp_args.sv_*: the launch also reduces its final rows to softmax_valid's partial sums (sv: an _ops.SvPartials).
    Returns True when the launch takes the ride (sv.parts is then the slice count), False when it cannot (chain-kernel regime,
    shape): the pose head then runs its own partial-sums launch."""
    if sv is None:
        return False
    B, N, _ = sv.xyz.shape
    if a.rows != B * N:
        raise ValueError("the SvPartials cloud (%d x %d points) does not match the MLP's %d rows" % (B, N, a.rows))
    a.sv_npoints = N
    parts = L.lib().elo_mlp_sv_parts(ctypes.byref(a), ctypes.byref(b) if b is not None else None)
    if parts <= 0:
        a.sv_npoints = 0
        return False
    a.sv_scratch, a.sv_xyz = sv.scratch.data_ptr(), sv.xyz.data_ptr()
    if b is None:
        feat = sv.feature
        if feat is None or feat.shape != (B, N, 64) or feat.dtype != dt:
            raise ValueError("a single launch takes its (B, N, 64) features of the MLP's storage dtype from SvPartials.feature")
        sv.feature = feat = feat.contiguous()
        a.sv_feature = feat.data_ptr()
    sv.parts = parts
    return True


def mlp(sources, layers, clear=None, sv=None):
    """Row-wise MLP over concat(sources, -1) without building the concat.  sources: (..., C_i) tensors.
    `clear`: ProjectionBuffers of a later projection, cleared on the side (for a pose head that takes the partial sums of `sv`).
    `sv`: an _ops.SvPartials (with .feature): the output are softmax_valid's logits and the launch also computes its partial
    sums -- when it can (sv.parts > 0 afterwards); `clear` then rides only if it does."""
    a, out, _keep = _mlp_args(sources, layers)
    if _ride_sv(a, None, sv, out.dtype) or sv is None:
        _ride_clear(a, clear, out.dtype)
    L.call("elo_mlp_fused", a, out)
    return out.reshape(sources[0].shape[:-1] + (layers[-1].N,))


def mlp_pair(sources_a, layers_a, sources_b, layers_b):
    """Two row-wise MLPs of identical shape in ONE launch."""
    a, out_a, _ka = _mlp_args(sources_a, layers_a)
    b, out_b, _kb = _mlp_args(sources_b, layers_b)
    L.call2("elo_mlp_fused2", a, b, out_a)
    lead = sources_a[0].shape[:-1]
    return out_a.reshape(lead + (layers_a[-1].N,)), out_b.reshape(lead + (layers_b[-1].N,))


def _mlp2_args(sources, layers, before, after, layers2):
    a, out, keep = _mlp_args(sources, layers)
    rows = out.shape[0]
    before, after = _features(before, after, out)[0][:2]          # same storage dtype as the first stage
    flat = lambda t: None if t is None else t.reshape(rows, t.shape[-1])
    before, after = flat(before), flat(after)
    out2 = torch.empty((rows, layers2[-1].N), dtype=out.dtype, device=out.device)
    a.n_layers2, a.layers2, a.out2 = len(layers2), _chain(layers2), out2.data_ptr()
    a.before, a.w_before = (before.data_ptr(), before.shape[1]) if before is not None else (None, 0)
    a.after, a.w_after = (after.data_ptr(), after.shape[1]) if after is not None else (None, 0)
    return a, out, out2, (keep, before, after, layers2)


def mlp2_pair(job_a, job_b, clear=None, sv=None):
    """Two jobs of identical shape, each TWO chained row-wise MLPs, in ONE launch:
    out = layers(concat(sources)); out2 = layers2(concat(before, out, after)).  job: dict(sources, layers, before,
    after, layers2); layers2[0] is packed with `row_order=stage2_row_order(w_before, N, w_after)` (the kernel's columns
    are [out | before | after]).  Returns ((out_a, out2_a), (out_b, out2_b)).
    `sv`: an _ops.SvPartials: out2_a are softmax_valid's logits, out2_b its features, and the launch also computes the partial
    sums when it can (see mlp)."""
    a, out_a, out2_a, _ka = _mlp2_args(**job_a)
    b, out_b, out2_b, _kb = _mlp2_args(**job_b)
    if _ride_sv(a, b, sv, out_a.dtype) or sv is None:
        _ride_clear(a, clear, out_a.dtype)                        # (clear: as for mlp)
    L.call2("elo_mlp_fused2", a, b, out_a)
    lead = job_a["sources"][0].shape[:-1]
    shape = lambda t: t.reshape(lead + (t.shape[-1],))
    return (shape(out_a), shape(out2_a)), (shape(out_b), shape(out2_b))


THROUGHPUT_BATCH = 4    # from this batch on a forward keeps the GPU busy by itself and the chain kernels win at every level


def _prepass_rows(stage=1, batch=1):
    """Rows (centre points x K) per launch from which a cost-volume stage runs its grouping as a PRE-PASS (a stand-alone
    grouping launch writes idx / mask) and the REGISTER-RESIDENT kernel (cv1_rr_kernel / cv2_rr_kernel: one W stream per
    128 rows through an LDS ring, activations in registers) starts from them, instead of the tile kernel with in-kernel
    grouping.  Two regimes, measured with 8 lanes and fp16 features (pairs/s, tile-kernel thresholds against chain-kernel
    thresholds): batch 1 10 320 / 9 760 and batch 2 14 520 / 14 310 -- a forward is a latency chain, the tile kernels (with
    the set-upconv riders of their level, and no extra launch) are faster, so only the largest launches take the chain
    (stage 1 from 20 000 rows -- 24 576 until round 5: with four forwards in flight l0 of a 64 x 1800 pair, 21 600 rows, is
    faster on the chain kernel next to an unmerged set-upconv chain launch, 11.2 -> 11.7 k pairs/s, profiles/r05_batch1_regimes.txt
    -- stage 2 from 65 536); batch 4 19 520 / 21 330 and batch 8 24 040 / 25 450 -- the GPU is full, a
    kernel costs its CU-time, and every level takes the chain (from 8192 rows; the same switch sits in elo_setconv_fused2 and
    elo_mlp_fused2).  tuning `cv_prepass` (ELO_CV_PREPASS): 0 = never, 1 = always, N = from N rows on (both stages, any batch)."""
    e = tuning.get("cv_prepass")
    if e is None:
        return 8192 if batch >= THROUGHPUT_BATCH else (20000 if stage == 1 else 65536)
    return (1 << 60) if e == 0 else 0 if e == 1 else int(e)


_HW = {}


def _all_pixels(B, H, W, device):
    """(B, H*W, 2) int32 (h, w) of every pixel, row-major: the centre list of the cost volume's select-k (cached)."""
    key = (B, H, W, str(device))
    hit = _HW.get(key)
    if hit is None:
        hh = torch.arange(H, dtype=torch.int32, device=device).view(1, H, 1, 1).expand(B, H, W, 1)
        ww = torch.arange(W, dtype=torch.int32, device=device).view(1, 1, W, 1).expand(B, H, W, 1)
        hit = torch.cat([hh, ww], -1).reshape(B, H * W, 2).contiguous()
        if hit.is_cuda and torch.cuda.is_current_stream_capturing():
            return hit
        if len(_HW) >= 64:
            _HW.clear()
        _HW[key] = hit
    return hit




def group_prepass(kind, xyz1_grid, xyz2_grid, group, K):
    """A cost volume's grouping (every pixel of xyz1_grid a centre; kind "select" = stage 1's select-k, "random" = stage
    2's random-k) as its own launch: (idx (B,N,K,3) int32, mask (B,N,K)).  Forms: random-k takes the LDS-tiled
    elo_fused_conv_random_k_dense when its tile fits; select-k the LDS-tiled elo_fused_conv_select_k_dense on large grids
    (>= 1024 tiles of 64 centres: 64x1800 is 2.5x faster there) and the wave-per-centre kernel on the pyramid's small
    levels (its fixed cost is a third of the tiled one's)."""
    B, H, W, _ = xyz1_grid.shape
    _, H2, W2, _ = xyz2_grid.shape
    N, kH, kW = H * W, group.kernel_h, group.kernel_w
    idx = torch.empty((B, N, K, 3), dtype=torch.int32, device=xyz1_grid.device)
    mask = torch.empty((B, N, K), dtype=torch.float32, device=xyz1_grid.device)
    if kind == "select":
        from .fused_conv import _select_dense_fits                # the launcher's own bounds (K <= 7, <= 512 slots, 64 KB of LDS)
        dense = -(-W // 64) * H * B >= tuning.get("select_dense_tiles") and _select_dense_fits(kH, kW, K, 0, group.stride_h, group.stride_w)
        entry = "elo_fused_conv_select_k_dense" if dense else "elo_fused_conv_select_k"
    else:
        RH, RW = 1 // group.stride_h + kH, 63 // group.stride_w + kW           # (fused_conv._dense_fits)
        dense = 4 * ((kH * kW + 7) & ~7) + 16 * RH * RW + 4 * (129 * K + 256) <= 64 * 1024
        entry = "elo_fused_conv_random_k_dense" if dense else "elo_fused_conv_random_k"
    hw = None if dense else _all_pixels(B, H, W, xyz1_grid.device)
    a = L.GroupArgs(B, H, W, H2, W2, N, kH, kW, K, 0, group.distance, group.stride_h, group.stride_w,
                    xyz1_grid.data_ptr(), xyz2_grid.data_ptr(), None if dense else hw.data_ptr(), group.random_hw.data_ptr(),
                    idx.data_ptr(), None, None, mask.data_ptr())
    L.call(entry, a, idx)
    return idx, mask


def _rr_path(group, B, N, K, C, stage=1):
    """True when a cost-volume stage with in-kernel grouping should take the pre-pass + register-resident kernel."""
    return (group is not None and not group.want_indices and K <= 32 and B * N * K >= _prepass_rows(stage, B)
            and group.stride_h == 1 and group.stride_w == 1 and not fp32_mfma()
            and C in (16, 32, 64))


_RECORD = None          # a list while `recording()` is active: the cost-volume calls of a forward, arguments cloned


class recording:
    """`with fused.recording() as calls:` -- keep (stage, args, kwargs) of every cv_stage1 / cv_stage2 call made inside,
    with tensor arguments CLONED (a forward recycles its intermediates): bench.py times the cost-volume launches on the
    tensors a real forward feeds them instead of on synthesised ones."""

    def __enter__(self):
        global _RECORD
        self.prev, _RECORD = _RECORD, []
        return _RECORD

    def __exit__(self, *exc):
        global _RECORD
        _RECORD = self.prev


def _record(stage, args, kwargs):
    keep = lambda v: v.detach().clone() if torch.is_tensor(v) else v
    kw = {k: keep(v) for k, v in kwargs.items() if k != "side"}
    if kwargs.get("group") is not None:
        g = kwargs["group"]
        kw["group"] = Grouping(g.random_hw.clone(), [g.kernel_h, g.kernel_w], g.distance, g.stride_h, g.stride_w)
    _RECORD.append((stage, tuple(keep(v) for v in args), kw, bool(kwargs.get("side"))))


def cv_stage1(xyz1, feat1, xyz2_proj, feat2_proj, idx, mask, cv0, cv1, cv2, cv_xyz, sum_cv0, sum_cv1, group=None, K=None,
              side=None, side_chain=False):
    """Cost-volume stage 1 in one launch.  `side`: one or two set-conv jobs (dicts of setconv() keyword arguments, same
    shape) that only share inputs with the cost volume -- run INSIDE this launch (elo_cv_stage1_setconv_fused); the call
    then returns (out, [(out_a, new_xyz_a), ...]).
    `side_chain` (with two side jobs): the side jobs ride only if stage 1 AND they take the register-resident chain form
    (select-k pre-pass, then ONE launch of cv1_rr + set-conv chain workgroups: elo_cv_stage1_setconv_chain); otherwise the
    cost volume runs alone -- in whatever form its size asks for -- and the call returns (out, None): the caller launches
    the set-convs itself."""
    L.require_gpu(xyz1, feat1, xyz2_proj, feat2_proj, idx, mask)
    if _RECORD is not None:
        _record(1, (xyz1, feat1, xyz2_proj, feat2_proj, idx, mask, cv0, cv1, cv2, cv_xyz, sum_cv0, sum_cv1), dict(group=group, K=K, side=side))
    xyz1, xyz2_proj = _f32c(xyz1), _f32c(xyz2_proj)
    (feat1, feat2_proj), dt, code = _features(feat1, feat2_proj)
    _, H2, W2, C = feat2_proj.shape
    B, N = xyz1.shape[0], xyz1.shape[1]
    ptr = lambda x: x.data_ptr() if x is not None else None
    chain_jobs = None
    if side and side_chain:
        rr = len(side) == 2 and N == H2 * W2 and _rr_path(group, B, N, K, C)
        if rr:
            chain_jobs = [_setconv_args(**job) for job in side]          # (args, out, new_xyz, keep-alive)
            probe = L.Cv1Args(B, N, K, H2, W2, C, None, None, None, None, None, None, cv0.struct(), cv1.struct(), cv2.struct(),
                              cv_xyz.struct(), sum_cv0.struct(), sum_cv1.struct(), None, _NO_GROUP, code)
            if L.lib().elo_cv_stage1_setconv_chain_form(ctypes.byref(probe), ctypes.byref(chain_jobs[0][0]), ctypes.byref(chain_jobs[1][0])) != 1:
                chain_jobs = None
        if chain_jobs is None:
            return cv_stage1_alone(xyz1, feat1, xyz2_proj, feat2_proj, idx, mask, cv0, cv1, cv2, cv_xyz, sum_cv0, sum_cv1, group, K), None
    if (not side or chain_jobs) and N == H2 * W2 and _rr_path(group, B, N, K, C):
        idx, mask = group_prepass("select", xyz1.reshape(B, H2, W2, 3), xyz2_proj, group, K)
        group = None
    if group is None:
        idx, mask = idx.contiguous(), _f32c(mask)
        K = idx.shape[2]
    out = torch.empty((B, N, 64), dtype=dt, device=xyz1.device)
    a = L.Cv1Args(B, N, K, H2, W2, C, xyz1.data_ptr(), feat1.data_ptr(), xyz2_proj.data_ptr(), feat2_proj.data_ptr(),
                  ptr(idx), ptr(mask), cv0.struct(), cv1.struct(), cv2.struct(), cv_xyz.struct(),
                  sum_cv0.struct(), sum_cv1.struct(), out.data_ptr(),
                  group.struct(B, N, K, xyz1.device) if group is not None else _NO_GROUP, code)
    if chain_jobs:
        L.call3("elo_cv_stage1_setconv_chain", a, chain_jobs[0][0], chain_jobs[1][0], out)
        return out, [(j[1], j[2]) for j in chain_jobs]
    if side:
        jobs = [_setconv_args(**job) for job in side]                  # (args, out, new_xyz, keep-alive)
        L.call3("elo_cv_stage1_setconv_fused", a, jobs[0][0], jobs[1][0] if len(jobs) > 1 else None, out)
        return out, [(j[1], j[2]) for j in jobs]
    L.call("elo_cv_stage1_fused", a, out)
    return out


def cv_stage1_alone(xyz1, feat1, xyz2_proj, feat2_proj, idx, mask, cv0, cv1, cv2, cv_xyz, sum_cv0, sum_cv1, group, K):
    """cv_stage1 without side jobs and without a second recording of the call (cv_stage1(side_chain=True) falls back on it)."""
    global _RECORD
    keep, _RECORD = _RECORD, None
    try:
        return cv_stage1(xyz1, feat1, xyz2_proj, feat2_proj, idx, mask, cv0, cv1, cv2, cv_xyz, sum_cv0, sum_cv1, group=group, K=K)
    finally:
        _RECORD = keep


def cv_stage2(xyz1_proj, feat1_proj, cost_proj, idx, mask, xyz_enc, sum_cost0, sum_cost1, group=None, K=None):
    L.require_gpu(xyz1_proj, feat1_proj, cost_proj, idx, mask)
    if _RECORD is not None:
        _record(2, (xyz1_proj, feat1_proj, cost_proj, idx, mask, xyz_enc, sum_cost0, sum_cost1), dict(group=group, K=K))
    xyz1_proj = _f32c(xyz1_proj)
    (feat1_proj, cost_proj), dt, code = _features(feat1_proj, cost_proj)
    B, H, W, C = feat1_proj.shape
    N = H * W
    ptr = lambda x: x.data_ptr() if x is not None else None
    if _rr_path(group, B, N, K, C, stage=2):
        idx, mask = group_prepass("random", xyz1_proj, xyz1_proj, group, K)
        group = None
    if group is None:
        idx, mask = idx.contiguous(), _f32c(mask)
        K = idx.shape[2]
    out = torch.empty((B, N, 64), dtype=dt, device=xyz1_proj.device)
    a = L.Cv2Args(B, N, K, H, W, C, xyz1_proj.data_ptr(), feat1_proj.data_ptr(), cost_proj.data_ptr(), ptr(idx),
                  ptr(mask), xyz_enc.struct(), sum_cost0.struct(), sum_cost1.struct(), out.data_ptr(),
                  group.struct(B, N, K, xyz1_proj.device) if group is not None else _NO_GROUP, code)
    L.call("elo_cv_stage2_fused", a, out)
    return out
