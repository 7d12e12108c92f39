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
const char *elo_last_error(void);   /* thread-local, never NULL                 */

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

#ifdef __cplusplus
}
#endif
#endif /* ELO_H_ */
