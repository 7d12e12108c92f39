/*
 * include/elo.h -- C ABI of libelo_hip.so, the MI355X (gfx950) implementation of
 * EfficientLO-Net's projection-aware 3D feature hot path.
 *
 * Plain pointers and sizes only (no torch / TF types).  Every pointer is a
 * DEVICE pointer on the current HIP device unless stated otherwise; `stream`
 * is a hipStream_t passed as void* (NULL = the null stream).  Nothing here
 * allocates, synchronises or keeps global state, so every entry point is
 * re-entrant and safe to call while a hipGraph is being captured on `stream`.
 * All entry points return ELO_OK or a negative elo_status; elo_last_error()
 * gives the message for the calling thread.
 *
 * Each entry point names the reference interface it replaces
 * (paths relative to the upstream repository IRMVLab/EfficientLO-Net).
 */
#ifndef ELO_H_
#define ELO_H_

#ifdef __cplusplus
extern "C" {
#endif

typedef void *elo_stream_t; /* hipStream_t */

typedef enum elo_status {
    ELO_OK = 0,
    ELO_ERR_ARG = -1,    /* null pointer / non-positive size / inconsistent shapes      */
    ELO_ERR_LIMIT = -2,  /* outside the supported envelope (e.g. kernel window > 5000)  */
    ELO_ERR_LAUNCH = -3  /* hipGetLastError() after the launch                          */
} elo_status;

int elo_abi_version(void);          /* bumps when a struct below changes layout */
int elo_dense_f32(void);            /* 1: built with -DELO_DENSE_F32 (elo_dense.w_packed holds fp32 weights), 0: fp16 hi|lo */
const char *elo_last_error(void);   /* thread-local, never NULL                 */
/* Range check of the fp16-split operands (a debugging switch, process-wide; also ELO_RANGE_CHECK=1 in the environment):
 * while enabled, the fused kernels run instances that COUNT every activation / gathered feature with |x| >= 65504 or
 * NaN on its way into a matrix-core operand (where the fp16 split saturates instead of raising).
 * elo_range_check(1 / 0) switches it and returns the previous setting (-1: only query);
 * elo_range_violations() takes the count since the last call out of the device counter with ONE atomic exchange (a
 * violation recorded meanwhile by a checked launch on another stream is kept for the next call), synchronises `stream`
 * and returns it.  That counter is process-wide; a caller with several streams in flight gives every fused argument block its
 * own word instead (ABI 26: `range_counter` in elo_setconv_args / elo_mlp_args / elo_cv1_args / elo_cv2_args -- the checked instances
 * add there when it is not NULL): efficientlo-net_amd/model.py bakes a lane's word into the lane's checked graph, so a violation is
 * that lane's alone. */
int elo_range_check(int enable);
int elo_range_violations(unsigned long long *count, elo_stream_t stream);

/* Host runtime: one step of a captured forward (a "lane": a hipGraphExec_t with fixed input / output buffers) as one call --
 * if bytes != 0 a device-to-device hipMemcpyAsync(dst <- src) of the lane's input, then hipGraphLaunch, both on `stream`.
 * graph_exec: the hipGraphExec_t (torch: CUDAGraph.raw_cuda_graph_exec()).
 * ORDERING (ABI 23; the contract replaced is the reference's synchronous sess.run(feed_dict=...), main.py:372-381): with
 * order_event != NULL (a hipEvent_t the lane owns) the call records it on `producer` -- the stream whose work so far produced
 * src, or wrote the lane's input buffer in place (0 = the null stream) -- and makes `stream` wait for it before the copy and
 * the graph, so a caller may submit right behind the kernels that fill src (a producer found idle by hipStreamQuery has nothing
 * to wait for: no event is recorded then).  order_event == NULL: no ordering -- the caller
 * guarantees src (or the in-place input) is complete before `stream` reaches this step (inputs resident and synchronised,
 * or produced on `stream` itself).
 * LIFETIME: the caller keeps src alive until `stream` has run the copy, as with any asynchronous copy (torch: record_stream).
 * device >= 0: the lane's device; made current for the call when the calling thread's current device differs (and put back). */
int elo_graph_submit(void *graph_exec, elo_stream_t stream, void *dst, const void *src, unsigned long bytes,
                     elo_stream_t producer, void *order_event, int device);

/* The library's tuning: every choice of KERNEL FORM that is not a function of the arguments alone, as ONE value.  The library
 * itself reads no environment variable; the host fills this once (efficientlo-net_amd/_lib.py maps the ELO_* variables named
 * below onto it when it loads the library) and may change it between launches.  A form baked into a captured hipGraph stays
 * what it was at capture time: the host hashes the tuning into its capture generation and refuses to replay a graph under
 * another tuning (efficientlo-net_amd/model.py).  Every form of an entry point computes the same function (bit for bit, or to
 * fp32 summation order where a comment says so): these are speed choices.  The elo_debug_* hooks further down are
 * single-field shorthands for elo_set_tuning. */
typedef struct elo_tuning {
    int chain_forms;            /* 1: the register-resident ("chain") kernels may be taken, 0: tile kernels only   [ELO_CV1_RR]            */
    int narrow_mfma;            /* narrow set-conv layers: 0 VALU kernel, 1 matrix cores for 19 -> 16 -> 16 -> 32  [ELO_SETCONV_NARROW_MFMA] */
    int range_check;            /* 1: the fused kernels run their operand-range-checked instances (elo_range_check) [ELO_RANGE_CHECK]       */
    int select_dense_waves;     /* elo_fused_conv_select_k_dense: 4 / 8 / 16 waves per tile, 0 = by grid size     [ELO_SELECT_DENSE_WAVES]  */
    int random_dense_rows;      /* elo_fused_conv_random_k_dense: 2 / 4 rows per tile, 0 = by grid size            [ELO_DENSE_ROWS]          */
    long setconv_chain_rows;    /* rows per launch from which elo_setconv_fused2 takes the chain form; -1: 20 000 (below batch
                                   ELO_THROUGHPUT_BATCH: 100 000 until round 5)                                    [ELO_SETCONV_RR_ROWS]     */
    long mlp_chain_rows;        /* ... elo_mlp_fused2; -1: 2048 / 8192                                            [ELO_MLP_RR_ROWS]         */
    long small_tile_units;      /* 16-row tiles while 32-row tiles would give fewer workgroups than this (64)      [ELO_SMALL_TILE_UNITS]    */
    int pool_wave;              /* elo_masked_softmax_pool with C = 64, K <= 32: 1 (default) one WAVE per point, one or two light nontemporal
                                   loads per lane and tensor (HBM-cold streaming: tools/micro/hbm_probe.hip), 0 the quarter-wave form  [ELO_POOL_WAVE] */
} elo_tuning;
int elo_get_tuning(elo_tuning *out);             /* what the launchers read now: elo_set_tuning's value with pending elo_debug_* overrides on top */
int elo_get_tuning_base(elo_tuning *out);        /* what elo_set_tuning installed (no elo_debug_* override): the value a read-modify-write starts from */
int elo_set_tuning(const elo_tuning *in);      /* ELO_ERR_ARG on a field outside its domain (nothing is changed then) */

/* ------------------------------------------------------------------------- *
 * Neighbour grouping on the H x W range image.
 *
 * Replaces the launchers
 *   void FusedConvRandomKLauncher(int batch_size, int H, int W, int npoints,
 *        int kernel_size_H, int kernel_size_W, int K, int flag_copy,
 *        float distance, int stride_h, int stride_w, const float *xyz1,
 *        const float *xyz2, const int *idx_n2, const int *random_hw,
 *        int *selected_bhw_idx, float *valid_idx, float *valid_in_dis_idx,
 *        float *selected_mask, int small_h, int small_w)
 *     -- tf_ops/2d_conv_random_k/fused_conv_g.cu:162, declared fused_conv.cpp:73
 *   void FusedConvSelectKLauncher(... same 21 arguments ...)
 *     -- tf_ops/2d_conv_select_k/fused_conv_g.cu:215
 * and the four cudaMemset zero-fills of the op glue (fused_conv.cpp:154-166):
 * the kernels write every element of every output, so the caller passes
 * uninitialised buffers.
 *
 * Shapes (row-major, contiguous):
 *   xyz1 (batch,H,W,3) f32        centres' grid
 *   xyz2 (batch,H2,W2,3) f32      queried grid, H2=ceil(H/stride_h), W2=ceil(W/stride_w)
 *   idx_n2 (batch,npoints,2) i32  (h,w) of each centre in xyz1
 *   random_hw (kernel_h*kernel_w) i32   visiting order, a permutation of 0..KT-1
 *   selected_bhw_idx (batch,npoints,K,3) i32   OUT
 *   valid_idx, valid_in_dis_idx (batch,npoints,KT,1) f32   OUT, either may be NULL
 *        (no caller in the model reads them: pointnet_util.py:49,106,197,272)
 *   selected_mask (batch,npoints,K,1) f32   OUT
 * Preconditions mirrored from the op (fused_conv.cpp:78-123) and from the
 * kernel's own bounds: sizes > 0, flag_copy in {0,1}, KT <= 5000
 * (fused_conv_g.cu:42-43), kernel_w/2 <= W2 (single cylindrical wrap, :89-97),
 * H2 < 32768, W2 < 65536.  random_hw being a permutation is NOT checked.
 * ------------------------------------------------------------------------- */
typedef struct elo_group_args {
    int batch, H, W;            /* xyz1 grid                               */
    int H2, W2;                 /* xyz2 grid ("small_h", "small_w")        */
    int npoints;
    int kernel_h, kernel_w;
    int K;
    int flag_copy;
    float distance;
    int stride_h, stride_w;
    const float *xyz1;
    const float *xyz2;
    const int *idx_n2;
    const int *random_hw;
    int *selected_bhw_idx;
    float *valid_idx;           /* nullable */
    float *valid_in_dis_idx;    /* nullable */
    float *selected_mask;
} elo_group_args;

int elo_fused_conv_random_k(const elo_group_args *a, elo_stream_t stream);
int elo_fused_conv_select_k(const elo_group_args *a, elo_stream_t stream);
/* elo_fused_conv_random_k for the call shape "EVERY pixel of xyz1 is a centre, in row-major order" -- idx_n2 =
 * get_hw_idx(...) (utils/pointnet_util.py:23-30), npoints == H*W: the cost-volume stage 2 and set-upconv calls
 * (:106-108, :272-274) and BASELINE configs[0].  a->idx_n2 is IGNORED (may be NULL); same outputs bit for bit.
 * The window union of a tile of 4 x 64 centres is staged once in LDS and every centre walks its window from there.
 * ELO_ERR_LIMIT when the window / K do not fit the 64 KB tile: call the general entry point. */
int elo_fused_conv_random_k_dense(const elo_group_args *a, elo_stream_t stream);
/* elo_fused_conv_select_k for the same call shape (every pixel a centre, row-major; a->idx_n2 ignored) with K <= 7,
 * flag_copy == 0 and a window of at most 512 slots -- the select-k of the refinement cost volumes
 * (utils/pointnet_util.py:49-51: K = 6 of 5x15 / 7x25 / 11x41).  The window union of 64 consecutive centres of a grid row
 * is staged once in LDS; a lane per centre walks it (no visiting order needed: the K smallest distances in increasing
 * order are unique unless two of them are EQUAL, and exactly then the centre is redone by the wave-per-centre form in
 * the reference's visiting order).  Same outputs bit for bit.  ELO_ERR_LIMIT outside those bounds. */
int elo_fused_conv_select_k_dense(const elo_group_args *a, elo_stream_t stream);
/* debugging hook: force 4, 8 or 16 waves per tile in elo_fused_conv_select_k_dense (0 = chosen by grid size; also
 * ELO_SELECT_DENSE_WAVES); returns the previous setting */
int elo_debug_select_dense_waves(int waves);

/* ------------------------------------------------------------------------- *
 * Feature path: fused gather / encode / pool kernels.  These replace chains of
 * stock TF ops (tf.gather_nd, tf.tile, tf.concat, tf.where, tf.nn.softmax,
 * tf.reduce_max/sum, tf.scatter_nd, ...) in utils/pointnet_util.py and
 * model_util.py; each comment names the lines it covers.  "idx" is always the
 * (batch,npoints,K,3) int32 (b,h,w) tensor produced by the grouping ops and
 * "mask" its (batch,npoints,K) float 0/1 mask.  Features are fp32, row-major,
 * channels last.  A masked slot gathers idx (0,0,0) and is multiplied by 0,
 * exactly like `tf.gather_nd(...) * mask` (Appendix A.4 of SURVEY.md).
 * ------------------------------------------------------------------------- */

/* set-conv / set-upconv input:  out[b,n,k,:] = [ src_xyz[idx]*m - centre_xyz[b,n] , src_feat[idx]*m ]
 * utils/pointnet_util.py:203-213 (down_conv) and :277-284 (up_conv). */
typedef struct elo_group_concat_args {
    int batch, npoints, K;
    int H2, W2, C;                /* source grid and its feature channels */
    const float *centre_xyz;      /* (batch,npoints,3) */
    const float *src_xyz;         /* (batch,H2,W2,3)   */
    const float *src_feat;        /* (batch,H2,W2,C)   */
    const int *idx;
    const float *mask;
    float *out;                   /* (batch,npoints,K,3+C) */
} elo_group_concat_args;
int elo_group_concat(const elo_group_concat_args *a, elo_stream_t stream);

/* out[b,n,c] = max_k x[b,n,k,c] * mask[b,n,k]     utils/pointnet_util.py:224-230, :295-298 */
typedef struct elo_masked_maxpool_args {
    int batch, npoints, K, C;
    const float *x;               /* (batch,npoints,K,C) */
    const float *mask;
    float *out;                   /* (batch,npoints,C)   */
} elo_masked_maxpool_args;
int elo_masked_maxpool(const elo_masked_maxpool_args *a, elo_stream_t stream);

/* Storage type of the FEATURE tensors of the three cost-volume kernels below (fields typed `void *`): fp32, or fp16
 * storage with fp32 arithmetic (BASELINE configs[2]; SURVEY.md section 8(d): s = 2).  Geometry inputs (xyz), indices
 * and masks are always fp32 / int32. */
enum { ELO_F32 = 0, ELO_F16 = 1 };

/* Cost volume, stage 1 (point -> patch of frame 2), utils/pointnet_util.py:54-66:
 *   q = xyz2[idx]*m, diff = q - p, euc = sqrt(sum(diff^2) + 1e-20)
 *   out[b,n,k,:] = [ p(3), q(3), diff(3), euc(1), feat1[b,n](C), feat2[idx]*m (C) ]      (10+2C channels) */
typedef struct elo_cv_encode1_args {
    int batch, npoints, K;
    int H2, W2, C;
    const float *xyz1;            /* (batch,npoints,3)  warped frame-1 points      */
    const void *feat1;            /* (batch,npoints,C)        dtype                */
    const float *xyz2;            /* (batch,H2,W2,3)                               */
    const void *feat2;            /* (batch,H2,W2,C)          dtype                */
    const int *idx;
    const float *mask;
    void *out;                    /* (batch,npoints,K,10+2C)  dtype                */
    int dtype;                    /* ELO_F32 / ELO_F16 */
} elo_cv_encode1_args;
int elo_cv_encode1(const elo_cv_encode1_args *a, elo_stream_t stream);

/* Cost volume, stage 2 (patch -> patch inside frame 1), utils/pointnet_util.py:110-129:
 *   g = xyz1_grid[idx]*m, diff = g - p, euc as above
 *   xyz_cat[b,n,k,:] = [ p, g, diff, euc ]                       (10 channels)
 *   rest[b,n,k,:]    = [ feat1[b,n] (C), cost[idx]*m (Cc) ]      (C+Cc channels)
 * (the reference concatenates conv(xyz_cat) in front of `rest`; the caller
 *  does that product as a split GEMM, see pointnet_util.cost_volume) */
typedef struct elo_cv_encode2_args {
    int batch, npoints, K;
    int H, W, C, Cc;              /* xyz1 grid (npoints == H*W), feat1 and cost channels */
    const float *xyz1;            /* (batch,H,W,3)  */
    const void *feat1;            /* (batch,H,W,C)   dtype */
    const void *cost;             /* (batch,H,W,Cc)  dtype: stage-1 output */
    const int *idx;
    const float *mask;
    void *xyz_cat;                /* (batch,npoints,K,10)    dtype */
    void *rest;                   /* (batch,npoints,K,C+Cc)  dtype */
    int dtype;                    /* ELO_F32 / ELO_F16 */
} elo_cv_encode2_args;
int elo_cv_encode2(const elo_cv_encode2_args *a, elo_stream_t stream);

/* out[b,n,c] = sum_k softmax_k( mask==1 ? logits : -1e10 )[k,c] * values[b,n,k,c]
 * utils/pointnet_util.py:92-98 and :137-146.  `values` rows may be a channel
 * slice of a wider tensor: element (row,c) is values[row*values_stride + c]. */
typedef struct elo_softmax_pool_args {
    int batch, npoints, K, C;
    const void *logits;           /* (batch,npoints,K,C)  dtype */
    const void *values;           /*                      dtype */
    int values_stride;            /* elements between consecutive (b,n,k) rows, >= C */
    const float *mask;
    void *out;                    /* (batch,npoints,C)    dtype; the softmax itself runs in fp32 */
    int dtype;                    /* ELO_F32 / ELO_F16 */
} elo_softmax_pool_args;
int elo_masked_softmax_pool(const elo_softmax_pool_args *a, elo_stream_t stream);

/* Fresh visiting orders per replay of a captured forward (tf.random_shuffle inside every operator on every sess.run:
 * utils/pointnet_util.py:45,104,193,270).  All order tensors of a forward are slices of `flat` (their decoded (dh, dw)
 * forms, elo_group_spec.decoded_hw, slices of `decoded`); `pool` holds `versions` pre-drawn contents of `flat`.  One
 * launch copies version (*cursor % versions) into flat, decodes it and advances the device-side cursor -- captured at the
 * head of a hipGraph it gives every replay its own orders at unchanged addresses.
 *   table (n_entries,4) i32: (offset, KT, kernel_h, kernel_w) per order tensor; entry_of (total) i32: slot -> entry. */
typedef struct elo_perm_refresh_args {
    const int *pool;            /* (versions, total) */
    int versions, total;
    int *cursor;                /* (1) device counter */
    int *flat, *decoded;        /* (total) each */
    const int *entry_of;        /* (total) */
    const int *table;           /* (n_entries, 4) */
    int n_entries;
} elo_perm_refresh_args;
int elo_perm_refresh(const elo_perm_refresh_args *a, elo_stream_t stream);

/* model_util.py:319-343 softmax_valid: per batch element, softmax over the
 * VALID points (xyz != (0,0,0)) per channel, out[b,0,c] = sum_n softmax*feature.
 * Two launches: `parts` blocks per batch element reduce slices of the point axis
 * with an online softmax, a second kernel merges the partials.
 * scratch: 3 * batch * ELO_SV_MAX_PARTS * C floats, laid out (3, batch, ELO_SV_MAX_PARTS, C): [maximum | denominator |
 * weighted sum] of slice `part` of batch element b (the partial launch writes at most 64 slices; a row-wise MLP launch that
 * computes the partial sums itself -- elo_mlp_args.sv_* -- one slice per row tile, up to ELO_SV_MAX_PARTS). */
#define ELO_SV_MAX_PARTS 512
typedef struct elo_softmax_valid_args {
    int batch, npoints, C;
    const float *feature;         /* (batch,npoints,C) */
    const float *weight;          /* (batch,npoints,C) */
    const float *xyz;             /* (batch,npoints,3): a point is valid unless all three are exactly 0 */
    float *out;                   /* (batch,1,C); all-invalid batch element -> 0 */
    float *scratch;
    float *stats;                 /* (batch,2,C) OUT or NULL: [maximum | denominator] of the masked softmax per channel (what
                                     elo_softmax_valid_backward needs besides `out`; denominator 0 = no valid point) */
} elo_softmax_valid_args;
int elo_softmax_valid(const elo_softmax_valid_args *a, elo_stream_t stream);

/* Pose head of one pyramid level, inference form (pwclo_model.py:194-208 at l3,
 * :262-280 / :338-356 / :406-425 at the refinement levels), fused:
 *   f      = softmax_valid(feature, weight, xyz)                       (B,C)
 *   big    = f @ W_big + b_big                                         (B,hidden)   conv1d, no activation
 *   q_det  = normalise(big @ W_q + b_q),  t_det = big @ W_t + b_t      normalise: q / (sqrt(sum q^2 + 1e-10) + 1e-10)
 *   coarse == NULL :  q = q_det, t = t_det                             (l3)
 *   else           :  q = q_det (x) q_coarse,
 *                     t = (q_det (x) [0,t_coarse] (x) q_det^-1)[1:] + t_det
 *   q_norm = normalise(q)                                              (:427-430)
 * Weights are row-major (in,out).  Dropout (training only) is not part of this kernel.
 * scratch: as elo_softmax_valid. */
typedef struct elo_pose_head_args {
    int batch, npoints, C, hidden;
    const void *feature, *weight; /* (batch,npoints,C) feat_dtype */
    const float *xyz;
    const float *W_big, *b_big;   /* (C,hidden), (hidden) */
    const float *W_q, *b_q;       /* (hidden,4), (4)      */
    const float *W_t, *b_t;       /* (hidden,3), (3)      */
    const float *q_coarse;        /* (batch,4) or NULL    */
    const float *t_coarse;        /* (batch,3) or NULL    */
    float *q, *t, *q_norm;        /* (batch,4), (batch,3), (batch,4) OUT */
    float *scratch;
    float *pose7;                 /* (batch,7) [q_norm | t] OUT, or NULL: the pose as one row a caller can log */
    /* Optional side job for the workgroups of the first launch: clear the buffers of the elo_warp_project call
     * that will consume this pose (its scratch min-range words and its two outputs), so that call can set
     * `prepared` and skip its own init launch.  clear_scratch == NULL: no side job. */
    unsigned *clear_scratch;      /* the warp call's `scratch`: its first clear_cells + 4*batch words are set to 0x7f7f7f7f */
    float *clear_xyz;             /* its out_xyz  (clear_cells*3 floats  <- 0) */
    void *clear_feat;             /* its out_feat (clear_cells*clear_C elements of feat_dtype <- 0), NULL when clear_C == 0 */
    long clear_cells;             /* batch*H*W of that call */
    int clear_C;
    int feat_dtype;               /* ELO_F32 / ELO_F16: storage of feature, weight and of clear_feat (C, clear_C even for fp16) */
    /* pose7 as a RING, for a launch that is replayed from a captured graph (fixed pointers): with pose7_slots > 1 and
     * pose7_cursor != NULL, pose7 is (pose7_slots,batch,7); batch element b's row goes to slot pose7_cursor[b] % pose7_slots
     * and pose7_cursor[b] is incremented (by the one thread that writes the row).  A stream of frame pairs then needs no
     * copy-out per pair: the caller drains the ring every pose7_slots replays (and may reset the cursors to 0). */
    int pose7_slots;
    unsigned *pose7_cursor;       /* (batch) or NULL */
    /* Optional side job for the LAST launch of a captured forward (the l0 pose head): load the next pooled set of window
     * visiting orders (elo_perm_refresh semantics) once this pose is written -- the NEXT replay of the graph then walks
     * fresh orders without a launch of its own.  next_orders.pool == NULL: no side job. */
    elo_perm_refresh_args next_orders;
    /* ready_parts > 0: the partial sums of softmax_valid are ALREADY in `scratch`, ready_parts slices per batch element, written
     * by the launch that produced `feature` / `weight` (elo_mlp_fused / elo_mlp_fused2 with sv_scratch == scratch;
     * elo_mlp_sv_parts gives the count): no partial-sums launch here -- one launch less per pyramid level.  The clear_* buffers
     * are then not cleared by this call (elo_mlp_args.clear_* did it). */
    int ready_parts;
} elo_pose_head_args;
int elo_pose_head(const elo_pose_head_args *a, elo_stream_t stream);

/* Pose warp + spherical re-projection.
 *   warp : p' = ((q (x) [0,p]) (x) q^-1)[1:] + t, zeroed where p == (0,0,0)
 *          (pwclo_model.py:213-227; model_util.py:17-69), skipped when q == NULL
 *   project : model_util.py:181-292 ProjectPC2SphericalRing -- per point
 *          r = |p'|, col = int((pi - atan2(y,x)) / az_res), row = H - int(asin(z/r)/vert_res + vert_off)
 *          (NaN -> 0, the GPU float->int convention), both clipped; the point(s)
 *          with the minimum r of a cell are SUMMED into it (tf.scatter_nd adds).
 * az_res / vert_res / vert_off are computed by the caller exactly as
 * model_util.py:189-200 does (python double -> float32).
 * scratch: (batch*H*W) + 4*batch + 2*(batch*npoints) 32-bit device words
 *          [min range per cell | 4 zero-point flags per image | cell of point | range bits of point]. */
typedef struct elo_warp_project_args {
    int batch, npoints, C;        /* C may be 0 (no features) */
    int H, W;
    float az_res, vert_res, vert_off;
    const float *xyz;             /* (batch,npoints,3) */
    const void *feat;             /* (batch,npoints,C) feat_dtype, or NULL */
    const float *q;               /* (batch,4) or NULL = no warp */
    const float *t;               /* (batch,3) */
    float *warped;                /* (batch,npoints,3) OUT (nullable when q == NULL) */
    float *out_xyz;               /* (batch,H,W,3) OUT */
    void *out_feat;               /* (batch,H,W,C) feat_dtype OUT, or NULL */
    unsigned *scratch;
    int prepared;                 /* 1: scratch / out_xyz / out_feat were cleared by elo_pose_head (clear_*): no init launch */
    int feat_dtype;               /* ELO_F32 / ELO_F16 storage of feat / out_feat (fp16: C even; duplicates are summed with
                                     packed fp16 atomics, i.e. rounded per addition) */
} elo_warp_project_args;
int elo_warp_project(const elo_warp_project_args *a, elo_stream_t stream);

/* Raw-cloud input stage: PreProcess (model_util.py:346-445; the point part) + the input ProjectPC2SphericalRing of
 * BOTH frames (pwclo_model.py:54-67) in three launches (clear, per-point pass, scatter).  Per point p of frame f:
 *   valid = any(p != 0);  p4 = [p, 1], zeroed (all four) where sqrt(x^2 + y^2) > crop_xy  (model_util.py:380-383);
 *   p4 <- T_trans[b] . p4  when aug_frame[b] == f  (:392-394, :408-410);  out = p4[:3] * valid  (:421-422)
 * then the projection of elo_warp_project (no warp).  Outputs are STACKED over frames, frame 1 of every batch element
 * first: points (2*batch, npoints, 3), out_xyz (2*batch, H, W, 3) -- the layout the Siamese pyramid runs as one batch.
 * q_gt / t_gt of PreProcess (a 4x4 product and an Euler round trip per batch element) stay with the caller.
 * scratch: (2*batch*H*W) + 4*(2*batch) + 2*(2*batch*npoints) 32-bit device words. */
typedef struct elo_input_stage_args {
    int batch, npoints;           /* points per frame */
    int point_stride;             /* floats per point in `cloud` (>= 3: x, y, z first; main.py feeds 6) */
    int H, W;
    float az_res, vert_res, vert_off;   /* as in elo_warp_project_args */
    float crop_xy;                /* 35 (model_util.py:380) */
    const float *cloud;           /* (batch, 2*npoints, point_stride): frame 1's points, then frame 2's (pwclo_model.py:56-57) */
    const float *T_trans;         /* (batch,4,4) row-major, or NULL = no augmentation */
    const int *aug_frame;         /* (batch) 1 or 2: the frame T_trans applies to (NULL with T_trans == NULL) */
    float *points;                /* (2*batch, npoints, 3) OUT */
    float *out_xyz;               /* (2*batch, H, W, 3) OUT */
    unsigned *scratch;
} elo_input_stage_args;
int elo_input_stage(const elo_input_stage_args *a, elo_stream_t stream);

/* elo_pose_head followed by elo_warp_project of the NEXT level's cloud by the pose it just computed
 * (pwclo_model.py:211-236 after :194-208 / :262-280), in three launches instead of five: the projection's buffers
 * are cleared by the pose head's first launch (a->clear_* must name w's scratch / out_xyz / out_feat), every workgroup
 * of the second launch recomputes the head (block 0 stores it) and warps + bins its 256 points with the (q, t) it holds,
 * the third launch is the projection's scatter.  w->q / w->t are ignored (the pose is a->q, a->t); w->warped is required. */
int elo_pose_head_warp(const elo_pose_head_args *a, const elo_warp_project_args *w, elo_stream_t stream);

/* ------------------------------------------------------------------------- *
 * Backward passes of the feature kernels above, for TRAINING (csrc/elo_backward.hip).
 * The reference trains through TensorFlow's autodiff of its stock ops (main.py:171-176): gather_nd -> scatter-add of
 * the incoming gradient (utils/pointnet_util.py:54-55, :110-111, :203-204, :277-278; masks are stop_gradient, indices
 * integer), reduce_max -> the maximal entries (even split among exact ties), softmax, scatter_nd -> gather
 * (model_util.py:264-273).  Each entry point is the adjoint of ONE forward entry point: same shapes, `grad_x` has the
 * shape of `x`.  fp32.  Outputs marked ACC receive atomic adds and must be ZERO on entry; the others are written in full.
 * Any gradient output may be NULL (not wanted).  Masked slots add nothing anywhere.
 * ------------------------------------------------------------------------- */
typedef struct elo_group_concat_bwd_args {
    int batch, npoints, K;
    int H2, W2, C;
    const float *grad_out;        /* (batch,npoints,K,3+C) */
    const int *idx;
    const float *mask;
    float *grad_centre;           /* (batch,npoints,3)  = -sum_k grad_out[..., :3]        */
    float *grad_src_xyz;          /* (batch,H2,W2,3)  ACC                                  */
    float *grad_src_feat;         /* (batch,H2,W2,C)  ACC                                  */
} elo_group_concat_bwd_args;
int elo_group_concat_backward(const elo_group_concat_bwd_args *a, elo_stream_t stream);

typedef struct elo_masked_maxpool_bwd_args {
    int batch, npoints, K, C;
    const float *x;               /* the forward input (batch,npoints,K,C) */
    const float *mask;
    const float *grad_out;        /* (batch,npoints,C) */
    float *grad_x;                /* (batch,npoints,K,C) */
} elo_masked_maxpool_bwd_args;
int elo_masked_maxpool_backward(const elo_masked_maxpool_bwd_args *a, elo_stream_t stream);

typedef struct elo_cv_encode1_bwd_args {
    int batch, npoints, K;
    int H2, W2, C;
    const float *xyz1;            /* forward inputs: the geometry code's norm needs them */
    const float *xyz2;
    const int *idx;
    const float *mask;
    const float *grad_out;        /* (batch,npoints,K,10+2C) */
    float *grad_xyz1;             /* (batch,npoints,3) */
    float *grad_feat1;            /* (batch,npoints,C) */
    float *grad_xyz2;             /* (batch,H2,W2,3)  ACC */
    float *grad_feat2;            /* (batch,H2,W2,C)  ACC */
} elo_cv_encode1_bwd_args;
int elo_cv_encode1_backward(const elo_cv_encode1_bwd_args *a, elo_stream_t stream);

typedef struct elo_cv_encode2_bwd_args {
    int batch, npoints, K;
    int H, W, C, Cc;
    const float *xyz1;            /* (batch,H,W,3) */
    const int *idx;
    const float *mask;
    const float *grad_xyz_cat;    /* (batch,npoints,K,10)   */
    const float *grad_rest;       /* (batch,npoints,K,C+Cc) */
    float *grad_xyz1;             /* (batch,H,W,3)   ACC (centre and neighbours live in the same grid) */
    float *grad_feat1;            /* (batch,H,W,C)   */
    float *grad_cost;             /* (batch,H,W,Cc)  ACC */
} elo_cv_encode2_bwd_args;
int elo_cv_encode2_backward(const elo_cv_encode2_bwd_args *a, elo_stream_t stream);

typedef struct elo_softmax_pool_bwd_args {
    int batch, npoints, K, C;
    const float *logits;          /* forward inputs */
    const float *values;
    int values_stride;
    const float *mask;
    const float *grad_out;        /* (batch,npoints,C) */
    float *grad_logits;           /* (batch,npoints,K,C); 0 at masked slots (tf.where feeds a constant there) */
    float *grad_values;           /* (batch,npoints,K,C) contiguous, whatever values_stride was */
} elo_softmax_pool_bwd_args;
int elo_masked_softmax_pool_backward(const elo_softmax_pool_bwd_args *a, elo_stream_t stream);

typedef struct elo_softmax_valid_bwd_args {
    int batch, npoints, C;
    const float *feature, *weight, *xyz;     /* forward inputs */
    const float *grad_out;        /* (batch,1,C) */
    float *grad_feature;          /* (batch,npoints,C) */
    float *grad_weight;           /* (batch,npoints,C) */
    const float *out, *stats;     /* the forward's out (batch,1,C) and stats (batch,2,C): with them the adjoint is one
                                     element-wise pass; both NULL: the kernel recomputes them (one block per batch element
                                     and 64 channels, three serial passes over the points -- 300 us at 3600 points) */
} elo_softmax_valid_bwd_args;
int elo_softmax_valid_backward(const elo_softmax_valid_bwd_args *a, elo_stream_t stream);

/* Adjoint of elo_warp_project.  `scratch` is the forward call's scratch as it left it (per-cell minimum range, zero-point
 * flags, per-point cell and range bits: who won each cell); the forward's projection constants come along because the
 * cells a zero point can fall in are recomputed from them.  Gradients never reach the cell indices (model_util.py:255-273). */
typedef struct elo_warp_project_bwd_args {
    int batch, npoints, C;
    int H, W;
    float az_res, vert_res, vert_off;
    const float *xyz;             /* forward input (batch,npoints,3) */
    const float *q, *t;           /* forward inputs, or NULL = the forward did not warp */
    const unsigned *scratch;
    const float *grad_out_xyz;    /* (batch,H,W,3) or NULL */
    const float *grad_out_feat;   /* (batch,H,W,C) or NULL */
    const float *grad_warped;     /* (batch,npoints,3) or NULL: gradient on the forward's `warped` output */
    float *grad_xyz;              /* (batch,npoints,3) */
    float *grad_feat;             /* (batch,npoints,C) */
    float *grad_q;                /* (batch,4)  ACC */
    float *grad_t;                /* (batch,3)  ACC */
} elo_warp_project_bwd_args;
int elo_warp_project_backward(const elo_warp_project_bwd_args *a, elo_stream_t stream);

/* ---------------------------------------------------------------------------
 * Training layer: the ROW REDUCTIONS of conv2d -> batch norm (batch statistics) -> ReLU
 * (utils/tf_util.py:120-185 conv2d, :512-563 batch_norm_template) on a (rows, C) fp32 matrix, rows = B*N*K.
 * The dense products themselves (z = x W + b, dx = dz W^T) stay library GEMMs; these entry points are the passes over
 * the rows around them.  C: a power of two in 4..256 (every batch-normalised width of the model); all tensors fp32,
 * 16-byte aligned.  Reductions are per-block partial sums in caller-provided scratch, combined in a fixed order (fp64
 * for the batch-norm sums): no atomics, bit-reproducible.
 * ------------------------------------------------------------------------- */
#define ELO_BN_MAX_PARTS 512      /* partial rows of the batch-norm reductions: scratch = 2 * C * ELO_BN_MAX_PARTS floats */
typedef struct elo_bn_stats_args {
    long rows; int C;
    const float *z;               /* (rows,C) */
    float *scratch;               /* (ELO_BN_MAX_PARTS, 2, C) */
    float eps, momentum;          /* 1e-3; 1 - bn_decay */
    float *mean, *invstd;         /* (C) OUT: batch mean, 1/sqrt(biased batch variance + eps) */
    float *running_mean, *running_var;   /* (C) IN/OUT or both NULL: r <- (1-momentum) r + momentum * (mean | unbiased var) */
    int groups;                   /* > 1: the rows are `groups` equal contiguous blocks with their OWN batch statistics (the two frames of a
                                   * Siamese batch in one launch: utils/tf_util.py's layer called twice with shared variables): mean, invstd
                                   * are (groups,C), the moving averages take the groups' moments one after the other; 0 / 1: one group */
} elo_bn_stats_args;
int elo_bn_stats(const elo_bn_stats_args *a, elo_stream_t stream);
long elo_bn_scratch_floats(int C, int groups);  /* the scratch of elo_bn_stats / elo_bn_backward AS THIS BUILD sizes it (a host that mirrors the #define
                                             * and a stale library disagree silently: ask) */

typedef struct elo_bn_apply_args {
    long rows; int C;
    const float *z, *mean, *invstd, *gamma, *beta;
    int relu;                     /* 1: y = max(., 0) */
    float *y;                     /* (rows,C) OUT (may alias z) */
    int groups;                   /* as elo_bn_stats_args: mean, invstd (groups,C) */
} elo_bn_apply_args;
int elo_bn_apply(const elo_bn_apply_args *a, elo_stream_t stream);

/* g = dy * [gamma*xhat + beta > 0] (relu = 1) or dy;  sums <- [sum g | sum g*xhat] = [d beta | d gamma];
 * dz = gamma * invstd * (g - sum g / rows - xhat * sum g*xhat / rows)        (three launches) */
typedef struct elo_bn_backward_args {
    long rows; int C;
    const float *dy, *z, *mean, *invstd, *gamma, *beta;
    int relu;
    float *scratch;               /* (ELO_BN_MAX_PARTS, 2, C) */
    float *sums;                  /* (2*C) OUT [d beta | d gamma] */
    float *dz;                    /* (rows,C) OUT (may alias dy); NULL: the sums only (two launches) -- dz is then formed by elo_dense_rows */
    int groups;                   /* as elo_bn_stats_args: mean, invstd (groups,C), sums (groups,2*C) -- d beta / d gamma are their sums over the groups */
} elo_bn_backward_args;
int elo_bn_backward(const elo_bn_backward_args *a, elo_stream_t stream);

/* dW = x^T g (Cin,Cout row-major), db = column sums of g (or NULL): fp32 operands on v_mfma_f32_16x16x4_f32, fp32
 * accumulation per row slice, slices summed in order.  Any Cin, Cout.
 * scratch: elo_weight_grad_slices(rows, Cin, Cout) * (Cin*Cout + Cout) floats. */
typedef struct elo_weight_grad_args {
    long rows; int Cin, Cout;
    const float *x;               /* (rows,Cin)  */
    const float *g;               /* (rows,Cout) */
    float *dW;                    /* (Cin,Cout) OUT */
    float *db;                    /* (Cout) OUT or NULL */
    float *scratch;
} elo_weight_grad_args;
int elo_weight_grad_slices(long rows, int Cin, int Cout);
int elo_dense_weight_grad(const elo_weight_grad_args *a, elo_stream_t stream);

/* out = x W (+ bias) over the rows of a training layer (utils/tf_util.py:120-185: conv2d 1x1 and its adjoint dx = dz W^T with
 * transposed = 1), fp32 on v_mfma_f32_16x16x4_f32: one pass over x, W staged once per workgroup in LDS.  Any Cin; Cout <= 192
 * and ceil(Cin/16) * ceil(Cout/16) <= 160 (elo_dense_rows_supported).  scratch != NULL: ALSO the batch-norm moments of out, from the
 * accumulators (what elo_bn_stats computes with one more pass over out): mean, invstd and the moving averages as there. */
#define ELO_DENSE_MAX_PARTS 2048  /* scratch = 2 * Cout * ELO_DENSE_MAX_PARTS floats */
typedef struct elo_dense_rows_args {
    long rows; int Cin, Cout;
    const float *x;               /* (rows,Cin), 16-byte aligned */
    const float *W;               /* (Cin,Cout) row-major; transposed = 1: (Cout,Cin) row-major */
    int transposed;
    const float *bias;            /* (Cout) or NULL */
    float *out;                   /* (rows,Cout) OUT, 16-byte aligned */
    float *scratch;               /* NULL: no moments */
    float eps, momentum;
    float *mean, *invstd;         /* (Cout) OUT when scratch != NULL */
    float *running_mean, *running_var;   /* (Cout) IN/OUT or both NULL */
    /* bn_z != NULL (dx = dz W^T of a batch-normalised layer; Cin = that layer's width, % 4 == 0; no moments): x holds dy and the operand
     * is dz = gamma invstd (g - sum_g / rows - xhat sum_gxhat / rows), formed on the load and ALSO written to bn_dz (rows,Cin) --
     * elo_bn_backward's third launch inside this pass.  bn_sums: its (2*Cin) [sum g | sum g xhat]. */
    const float *bn_z, *bn_mean, *bn_invstd, *bn_gamma, *bn_beta, *bn_sums;
    int bn_relu;
    float *bn_dz;
    int groups;                   /* as elo_bn_stats_args, for the moments (mean, invstd (groups,Cout)) and for the bn_* operand
                                   * (bn_mean, bn_invstd (groups,Cin), bn_sums (groups,2*Cin)); the plain product ignores it */
} elo_dense_rows_args;
int elo_dense_rows_supported(long rows, int Cin, int Cout);
long elo_dense_rows_scratch_floats(int Cout, int groups);
int elo_dense_rows(const elo_dense_rows_args *a, elo_stream_t stream);

/* Adam (torch.optim.Adam's arithmetic; the reference trains with tf.train.AdamOptimizer, main.py:171-176) over ONE flat
 * fp32 parameter buffer in one launch: every variable, its gradient and its two moments are views of four buffers of n
 * floats.  hyper (device, 4 floats, written by the host before the step): [lr / (1 - beta1^t), 1 / sqrt(1 - beta2^t), eps, -]. */
typedef struct elo_adam_flat_args {
    long n;
    float *param;                 /* (n) IN/OUT */
    const float *grad;            /* (n)        */
    float *exp_avg;               /* (n) IN/OUT */
    float *exp_avg_sq;            /* (n) IN/OUT */
    const float *hyper;           /* (4) device */
    float beta1, beta2;
    float one_minus_beta1, one_minus_beta2;   /* (float)(1.0 - beta) from the caller's doubles */
} elo_adam_flat_args;
int elo_adam_flat(const elo_adam_flat_args *a, elo_stream_t stream);

/* The per-pair pose algebra of a TRAINING step in one launch each way (inference has it inside pose_head_kernel):
 * q_det = normalise(q_raw); coarse level (q_coarse == NULL): q = q_det, t = t_det; refinement level: q = q_det (x) q_coarse,
 * t = (q_det (x) [0, t_coarse] (x) q_det^-1)[1:] + t_det; q_norm = normalise(q)   (pwclo_model.py:197-208, :262-280;
 * model_util.py:17-69).  Forward: grad_q == NULL, writes q, t, q_norm.  Backward: grad_q / grad_t / grad_q_norm given (all
 * three), writes grad_q_raw, grad_t_det and (refinement level) grad_q_coarse, grad_t_coarse.  All tensors (batch,4) / (batch,3) fp32. */
typedef struct elo_pose_compose_args {
    int batch;
    const float *q_raw, *t_det, *q_coarse, *t_coarse;
    float *q, *t, *q_norm;
    const float *grad_q, *grad_t, *grad_q_norm;
    float *grad_q_raw, *grad_t_det, *grad_q_coarse, *grad_t_coarse;
} elo_pose_compose_args;
int elo_pose_compose(const elo_pose_compose_args *a, elo_stream_t stream);

/* get_loss (pwclo_model.py:437-481): per level L_q = mean_b |q_gt - normalise(q)|_2 (+1e-10 under the root), L_x = mean
 * sqrt((t - t_gt)^2 + 1e-10), level = L_x e^-w_x + w_x + L_q e^-w_q + w_q; loss = 0.2 l0 + 0.4 l1 + 0.8 l2 + 1.6 l3.
 * q[lv] / t[lv]: level lv = 0..3 (batch,4) / (batch,3).  Forward: grad_out == NULL, writes *loss.  Backward: grad_out (the
 * incoming scalar gradient, device) given, writes grad_q[lv], grad_t[lv], *grad_w_x, *grad_w_q. */
typedef struct elo_pose_loss_args {
    int batch;
    const float *q[4], *t[4];
    const float *q_gt, *t_gt;     /* (batch,4), (batch,3) */
    const float *w_x, *w_q;       /* scalars (device) */
    float *loss;
    const float *grad_out;
    float *grad_q[4], *grad_t[4];
    float *grad_w_x, *grad_w_q;
} elo_pose_loss_args;
int elo_pose_loss(const elo_pose_loss_args *a, elo_stream_t stream);

/* ------------------------------------------------------------------------- *
 * Fused inference kernels: gather/encode -> chain of 1x1 convolutions (BN and
 * bias folded, ReLU) -> pooling, in ONE launch with the activations of a
 * 16/32-row tile resident in LDS and the contractions on the matrix cores.
 * They compute what the unfused kernels above + the hipBLASLt GEMMs compute,
 * for the launch-bound small-batch regime.
 *
 * A layer is  y = act(x[K] @ W[K,N] + b[N]).  `w_packed` is W zero-padded to (Kp = ceil16(K), Np = ceil16(N)) and stored
 * in MFMA fragment order, every weight SPLIT into fp16 hi + lo (hi = fp16(w), lo = fp16(w - hi)): the kernels compute
 * hi*hi + hi*lo + lo*hi on the fp16 matrix cores with fp32 accumulation (2^-20 relative per product; fp16 range:
 * |x| < 65504 saturates -- see ELO_RANGE_CHECK below).  With KS = Kp / 16 blocks of 16 k and lane = 16*kq + n, the
 * fragment element (cb, ks, lane, s) is  W[ks*16 + 4*kq + s][cb*16 + n],  s = 0..3.  Per column block cb (contiguous,
 * KS * 1 KiB) the K axis is laid out as
 *     KS / 2 PAIRS of 32 k  (v_mfma_f32_16x16x32_f16), pair p = blocks 2p and 2p+1:
 *         1 KiB of hi8: per lane the eight halves  hi(cb, 2p, lane, 0..3), hi(cb, 2p+1, lane, 0..3)   (16 bytes),
 *         1 KiB of lo8: the same eight elements' lo halves,
 *     then, for odd KS, one TAIL of 16 k  (v_mfma_f32_16x16x16_f16), block KS-1:
 *         1 KiB: per lane  [hi0 hi1 hi2 hi3 | lo0 lo1 lo2 lo3]  (16 bytes),
 * so that every wave-wide 16-byte load is a contiguous 1 KiB.  ELO_PRODUCTS_HALF keeps only the round-to-nearest
 * halves: 1 KiB per pair (8 halves per lane), 512 bytes per tail.  Activations are split the same way ONCE, by the
 * layer (or the gather) that produces them, and live in the LDS tile as [hi x4 | lo x4] per four consecutive columns.
 * (A library built with -DELO_DENSE_F32 expects KS blocks of four fp32 weights per lane instead -- element order as
 * above, no pairs -- and uses v_mfma_f32_16x16x4_f32.)
 * `bias` has Np fp32 entries (zero padded).  Packing is done once per parameter update by the host
 * (efficientlo-net_amd/fused.py).
 *
 * Feature storage.  Every FEATURE tensor these kernels read or write in HBM (fields typed `void *`) is fp32 or fp16
 * according to the call's `feat_dtype` (ELO_F32 / ELO_F16; BASELINE configs[2]: s = 2 in SURVEY.md section 8(d)).
 * Geometry (xyz, centres, new_xyz), indices and masks are always fp32 / int32; all arithmetic is as described above
 * whatever the storage (an fp16 input is an exact operand: hi = x, lo = 0).  fp16 rows are read 16 bytes at a time:
 * channel counts must be multiples of 8 (fp32: of 4) for the vector paths; other widths take an element-wise path
 * where one exists (set-conv, row-wise MLP) and are an ELO_ERR_LIMIT otherwise (cost volume).
 * Masks are 0/1 by contract: `x * mask` is implemented as a select.
 *
 * Operands beyond the fp16 range saturate: elo_range_check() above counts them in a debugging run.
 * ------------------------------------------------------------------------- */
enum { ELO_PRODUCTS_SPLIT = 0,      /* fp32-class: hi/lo split operands, three fp16 MFMA products (default)     */
       ELO_PRODUCTS_HALF = 1 };     /* fp16 arithmetic: operands rounded to nearest fp16, ONE product, fp32 accumulate;
                                       `w_packed` then holds the halves only (pairs / tail as above, half the bytes).
                                       All layers of one launch must use the same mode.  (BASELINE configs[2].)   */

typedef struct elo_dense {
    const float *w_packed;
    const float *bias;
    int K, N;
    int relu;
    const float *w_plain;         /* optional: the same folded W as plain row-major (K,N); lets narrow chains
                                     (all widths <= 32) run on the wave-per-point VALU kernel (fp32 in either mode) */
    int products;                 /* ELO_PRODUCTS_SPLIT / ELO_PRODUCTS_HALF */
} elo_dense;

#define ELO_MAX_CHAIN 3

/* Optional in-kernel neighbour grouping for the fused kernels: with random_hw != NULL the kernel
 * runs the grouping itself (same semantics as elo_fused_conv_random_k / _select_k with flag_copy = 0,
 * one wave per centre) and the idx / mask INPUTS of the argument block are ignored (may be NULL).
 * idx_out / mask_out, when given, receive the (batch,npoints,K,3) indices and (batch,npoints,K) mask
 * exactly as the stand-alone ops would write them (used by the parity tests). */
typedef struct elo_group_spec {
    const int *random_hw;         /* (kernel_h*kernel_w) visiting order, or NULL = use idx/mask inputs */
    int kernel_h, kernel_w;
    float distance;
    int stride_h, stride_w;
    int *idx_out;                 /* nullable */
    float *mask_out;              /* nullable */
    const int *decoded_hw;        /* nullable: random_hw already decoded, entry i = (dh << 16) | (dw & 0xffff) with
                                   * dh = random_hw[i] / kernel_w - kernel_h/2, dw = random_hw[i] % kernel_w - kernel_w/2;
                                   * spares every tile two integer divisions per window slot */
} elo_group_spec;

/* set-conv / set-upconv stage 1:  group_concat -> chain -> masked max over K.
 * utils/pointnet_util.py:197-230 (down_conv) and :272-298 (up_conv). K <= 32.
 * centre_hw != NULL: centre = xyz1_grid[b, centre_hw[b,n,0], centre_hw[b,n,1]] and is also
 * written to new_xyz (the `new_xyz_proj` output of down_conv, :206); else centre = centre_xyz[b,n].
 * layers[0] expects its input rows ordered [features (C), xyz difference (3)] (the reference concatenates
 * [xyz_diff, features], :213: the host permutes the weight rows when packing; `w_plain` likewise). */
typedef struct elo_setconv_args {
    int batch, npoints, K;
    int H, W;                     /* xyz1 grid (only with centre_hw)        */
    int H2, W2, C;                /* gathered grid and its feature channels */
    const float *xyz1_grid;       /* (batch,H,W,3) or NULL                  */
    const int *centre_hw;         /* (batch,npoints,2) or NULL              */
    const float *centre_xyz;      /* (batch,npoints,3) or NULL              */
    const float *src_xyz;         /* (batch,H2,W2,3)                        */
    const void *src_feat;         /* (batch,H2,W2,C)   feat_dtype           */
    const int *idx;
    const float *mask;
    int n_layers;
    elo_dense layers[ELO_MAX_CHAIN];
    void *out;                    /* (batch,npoints,layers[last].N) feat_dtype */
    float *new_xyz;               /* (batch,npoints,3) or NULL              */
    elo_group_spec group;         /* random-k; needs xyz1_grid; centre_hw == NULL: centre n is pixel (n / W, n % W) */
    int feat_dtype;               /* ELO_F32 / ELO_F16 */
    unsigned long long *range_counter;   /* the CHECKED instances (elo_range_check) add their count of out-of-range operands here (a lane's own
                                          * device word: a violation is then that lane's, not every lane's); NULL: the process-wide counter of
                                          * elo_range_violations() */
} elo_setconv_args;
int elo_setconv_fused(const elo_setconv_args *a, elo_stream_t stream);
/* two independent jobs of identical shape in ONE launch (b may be NULL): the embedding and the embedding-mask
 * set-upconv of a refinement level (pwclo_model.py:247,250) share everything but weights and one input */
int elo_setconv_fused2(const elo_setconv_args *a, const elo_setconv_args *b, elo_stream_t stream);

/* Row-wise MLP over the concatenation of up to three row-aligned sources:
 * flow_predictor (utils/pointnet_util.py:153-175) and set-upconv stage 2 (:303-311). */
typedef struct elo_mlp_args {
    long rows;
    int n_sources;
    const void *src[3];           /* (rows, src_width[i]) feat_dtype */
    int src_width[3];
    int n_layers;
    elo_dense layers[ELO_MAX_CHAIN];
    void *out;                    /* (rows, layers[last].N) feat_dtype */
    /* Optional SECOND row-wise MLP in the same launch (n_layers2 > 0), fed by the first one's output: its input rows
     * are  [ out | before (w_before) | after (w_after) ]  -- flow_predictor's concat [points_f1, upsampled_feat,
     * cost_volume] (utils/pointnet_util.py:161-166) with the set-upconv stage 2 (:303-311) as the first MLP and its
     * output moved to the front (the host permutes the rows of layers2[0] when packing; layers[last].N % 4 == 0).
     * `out` is still written; the second output goes to out2. */
    int n_layers2;
    elo_dense layers2[ELO_MAX_CHAIN];
    const void *before;           /* (rows, w_before) feat_dtype, or NULL with w_before == 0 */
    int w_before;
    const void *after;            /* (rows, w_after)  feat_dtype, or NULL with w_after == 0  */
    int w_after;
    void *out2;                   /* (rows, layers2[last].N) feat_dtype */
    int feat_dtype;               /* ELO_F32 / ELO_F16 */
    /* Optional side job (job 0 of a paired launch only): the launch's workgroups also clear the buffers of an
     * elo_warp_project / elo_pose_head_warp call further down the stream -- elo_pose_head_args.clear_* semantics, for a
     * pose head run with ready_parts (which has no partial-sums launch of its own left to do it).  clear_scratch == NULL: no side job. */
    unsigned *clear_scratch;      /* first clear_cells + 4*clear_images words <- 0x7f7f7f7f */
    float *clear_xyz;             /* clear_cells*3 floats <- 0 */
    void *clear_feat;             /* clear_cells*clear_C elements of feat_dtype <- 0, NULL when clear_C == 0 */
    long clear_cells;             /* images*H*W of the projection */
    int clear_C;
    int clear_images;
    /* Batch size the rows came from (rows = batch x points), 0 = unknown: from ELO_THROUGHPUT_BATCH on the launcher takes the
     * register-resident kernel for every launch of 2048 rows or more (the GPU is full: a kernel costs its CU-time), below
     * that only from 8192 rows (a forward is a latency chain there and the tile kernel is faster). */
    int batch_hint;
    /* Optional (sv_scratch != NULL, the first job of a launch): the launch ALSO computes the first half of softmax_valid
     * (model_util.py:319-343) over its final output -- per row tile and channel the (maximum, denominator, weighted sum) triple
     * of the masked softmax over the tile's points, in elo_softmax_valid's scratch layout -- so the pose head that follows needs
     * no partial-sums launch (elo_pose_head_args.ready_parts = elo_mlp_sv_parts(a, b)).  The final output (out2, or out of a
     * one-stage MLP) must be 64 wide.
     *   paired launch (elo_mlp_fused2):  job a's final output are the LOGITS (`weight`), job b's the FEATURES; sv_feature NULL;
     *   single launch (elo_mlp_fused):   the final output are the logits, the features are sv_feature.
     * Rows are (batch, sv_npoints): a row tile never straddles two batch elements.  Taken by the tile kernel only: call
     * elo_mlp_sv_parts first -- 0 means this launch would run as a register-resident chain (or has another shape) and the
     * sv_* fields must stay NULL. */
    float *sv_scratch;            /* 3 * batch * ELO_SV_MAX_PARTS * 64 floats */
    const float *sv_xyz;          /* (batch, sv_npoints, 3): a row is a valid point unless all three are exactly 0 */
    const void *sv_feature;       /* single launch: (rows, 64) feat_dtype; paired launch: NULL */
    int sv_npoints;               /* rows per batch element (rows % sv_npoints == 0) */
    unsigned long long *range_counter;   /* the CHECKED instances (elo_range_check) add their count of out-of-range operands here (a lane's own
                                          * device word: a violation is then that lane's, not every lane's); NULL: the process-wide counter of
                                          * elo_range_violations() */
} elo_mlp_args;
#define ELO_THROUGHPUT_BATCH 4
int elo_mlp_fused(const elo_mlp_args *a, elo_stream_t stream);
int elo_mlp_fused2(const elo_mlp_args *a, const elo_mlp_args *b, elo_stream_t stream);   /* paired launch, as above */
/* Slices per batch element the launch (a, b) -- b NULL: a single launch -- would write with sv_scratch set (the sv_* fields
 * of `a` need not be filled yet, sv_npoints must be); 0: it cannot (chain-kernel regime, shape, more than ELO_SV_MAX_PARTS tiles). */
int elo_mlp_sv_parts(const elo_mlp_args *a, const elo_mlp_args *b);

/* Attentive cost volume, stage 1 (utils/pointnet_util.py:54-100) in one launch:
 * encode -> CV_0..2 -> CV_xyz -> sum_CV_0..1 -> masked softmax over K -> weighted sum.
 * cv0 expects its input rows ordered [feat1 (C), feat2 grouped (C), geometry (10)] (reference: [geometry, feat1,
 * feat2], :62-66) and sum_cv0 [x (64), xyz-encoding (64)] (reference: [encoding, x], :84): the host permutes the
 * weight rows when packing.  K <= 32, C % 4 == 0 (fp16 storage: C % 8 == 0). */
typedef struct elo_cv1_args {
    int batch, npoints, K;
    int H2, W2, C;
    const float *xyz1;            /* (batch,npoints,3) */
    const void *feat1;            /* (batch,npoints,C)  feat_dtype */
    const float *xyz2;            /* (batch,H2,W2,3)   */
    const void *feat2;            /* (batch,H2,W2,C)    feat_dtype */
    const int *idx;
    const float *mask;
    elo_dense cv0, cv1, cv2, cv_xyz, sum_cv0, sum_cv1;     /* N: 128,64,64,64,128,64 */
    void *out;                    /* (batch,npoints,64) feat_dtype */
    elo_group_spec group;         /* select-k of xyz2 around every pixel of xyz1 (npoints == H2*W2, stride 1) */
    int feat_dtype;               /* ELO_F32 / ELO_F16 */
    unsigned long long *range_counter;   /* the CHECKED instances (elo_range_check) add their count of out-of-range operands here (a lane's own
                                          * device word: a violation is then that lane's, not every lane's); NULL: the process-wide counter of
                                          * elo_range_violations() */
} elo_cv1_args;
int elo_cv_stage1_fused(const elo_cv1_args *a, elo_stream_t stream);
/* debugging hooks of the register-resident ("chain") kernel forms -- cv1_rr_kernel, cv2_rr_kernel, setconv_rr_kernel,
 * mlp2_rr_kernel: the forms elo_cv_stage1_fused / elo_cv_stage2_fused (pre-grouped calls), elo_setconv_fused2 and
 * elo_mlp_fused2 take from their row thresholds on.  Both forms of an entry point give the same bits.
 * elo_debug_cv1_rr(0) keeps ALL FOUR entry points on their tile kernels, 1 allows the chain forms (the default, also
 * ELO_CV1_RR), -1 = back to the environment's choice; returns the previous setting.  (The name is round 3's, when the
 * switch covered cost-volume stage 1 only.)
 * elo_debug_rr_rows(setconv_rows, mlp_rows): rows per launch from which elo_setconv_fused2 / elo_mlp_fused2 take the chain
 * form; -1 = the built-in regimes (or ELO_SETCONV_RR_ROWS / ELO_MLP_RR_ROWS, read once per process).
 * elo_debug_rr_launches(counts4, reset): launches of [cv1_rr, cv2_rr, setconv_rr, mlp2_rr] since the last reset. */
int elo_debug_cv1_rr(int on);
int elo_debug_rr_rows(long setconv_rows, long mlp_rows);
int elo_debug_rr_launches(unsigned long long *counts4, int reset);
int elo_debug_sv_ride_launches(unsigned long long *count, int reset);     /* ... and of mlp_sv_kernel (elo_mlp_args.sv_*) */
int elo_debug_chain_pair_launches(unsigned long long *count, int reset);  /* ... and of cv1_setconv_rr_kernel (elo_cv_stage1_setconv_chain) */
/* the two narrow set-conv layers of the pyramid (6 -> 8 -> 8 -> 16 and 19 -> 16 -> 16 -> 32, K = 32; elo_setconv_fused with
 * elo_dense.w_plain given): 1 = setconv_narrow_kernel, the MLP on the matrix cores, for the 19-channel layer (the default;
 * also ELO_SETCONV_NARROW_MFMA; the 6-channel layer stays on the VALU kernel: slower on the matrix cores, measured), 0 =
 * setconv_small_kernel, the fp32 VALU form, for both (the only one in the fp32-MFMA build), -1 = back to the environment's
 * choice; returns the previous setting.  Results agree to fp32-class rounding. */
int elo_debug_narrow_mfma(int on);
/* elo_debug_narrow_launches(counts2, reset): launches of [setconv_narrow_kernel, setconv_small_kernel] since the last reset
 * (the oracle test of the matrix-core form asserts through it which kernel produced the tensor) */
int elo_debug_narrow_launches(unsigned long long *counts2, int reset);
/* elo_cv_stage1_fused AND one or two set-conv jobs (elo_setconv_fused / elo_setconv_fused2 semantics, tile-kernel form;
 * jb may be NULL) in ONE launch: the first workgroups of the grid run cost-volume tiles, the rest set-conv tiles.
 * For branches that only share inputs -- the cost volume and the two set-upconvs of a refinement level
 * (pwclo_model.py:242-250), the initial cost volume and the layer-3 set-conv of the pyramid (:138, :170) -- so that a
 * small-batch forward pays one launch boundary and keeps both branches in flight together.  Same results bit for bit. */
int elo_cv_stage1_setconv_fused(const elo_cv1_args *a, const elo_setconv_args *ja, const elo_setconv_args *jb, elo_stream_t stream);
/* The same move on the register-resident kernels (round 5): the cost volume PRE-GROUPED (idx / mask of the select-k pre-pass), two
 * set-conv jobs of the set-upconv shape (64 + 3 -> 128 -> 64, in-kernel random-k): one launch of 512-thread chain workgroups, the
 * first ones on the cost volume, then job a, then job b -- the bits of elo_cv_stage1_fused + elo_setconv_fused2.
 * elo_cv_stage1_setconv_chain_form: 1 when (a's C and layers, the jobs' shape / size / products mode) take it; `a` need not carry
 * idx / mask or a grouping spec yet. */
int elo_cv_stage1_setconv_chain_form(const elo_cv1_args *a, const elo_setconv_args *ja, const elo_setconv_args *jb);
int elo_cv_stage1_setconv_chain(const elo_cv1_args *a, const elo_setconv_args *ja, const elo_setconv_args *jb, elo_stream_t stream);

/* Attentive cost volume, stage 2 (utils/pointnet_util.py:104-146) in one launch.
 * sum_cost0 expects input rows ordered [cost[idx]*m (64), xyz-encoding (64), feat1 (C)]
 * (reference order [encoding, feat1, grouped], :129). K <= 32, npoints == H*W, C % 4 == 0 (fp16: % 8), C <= 64. */
typedef struct elo_cv2_args {
    int batch, npoints, K;
    int H, W, C;
    const float *xyz1;            /* (batch,H,W,3)  */
    const void *feat1;            /* (batch,H,W,C)  feat_dtype */
    const void *cost;             /* (batch,H,W,64) feat_dtype */
    const int *idx;
    const float *mask;
    elo_dense xyz_enc, sum_cost0, sum_cost1;               /* N: 64,128,64 */
    void *out;                    /* (batch,npoints,64) feat_dtype */
    elo_group_spec group;         /* random-k of xyz1 around every pixel of xyz1 (stride 1) */
    int feat_dtype;               /* ELO_F32 / ELO_F16 */
    unsigned long long *range_counter;   /* the CHECKED instances (elo_range_check) add their count of out-of-range operands here (a lane's own
                                          * device word: a violation is then that lane's, not every lane's); NULL: the process-wide counter of
                                          * elo_range_violations() */
} elo_cv2_args;
int elo_cv_stage2_fused(const elo_cv2_args *a, elo_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* ELO_H_ */
