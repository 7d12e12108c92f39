// elo_project_device.h -- the cell a point falls in (ProjectPC2SphericalRing, model_util.py:234-245), shared by the
// forward projection (elo_features.hip) and its adjoint (elo_backward.hip): both must name the same cells.
#pragma once
#include "elo_common.h"

namespace elo {

// `at` = atan2f(y, x)
__device__ __forceinline__ int cell_of_point(float at, float z, float r, int H, int W, float az_res, float vert_res,
                                             float vert_off)
{
    const float PI_F = 3.14159265358979323846f;
    // float -> int of a NaN is 0 (the GPU convention, SURVEY a-10) -- spelled out, so that the compiler's constant
    // folder (scatter_min_kernel evaluates this for the literal zero point) and v_cvt_i32_f32 agree
    auto to_int = [](float v) { return v != v ? 0 : (int)v; };
    int col = to_int((PI_F - at) / az_res);                                  // model_util.py:234-235
    const float beta = asinf(z / r);
    int row = H - to_int(beta / vert_res + vert_off);                        // NaN -> 0, :237-242
    row = row < 0 ? 0 : row > H - 1 ? H - 1 : row;
    col = col < 0 ? 0 : col > W - 1 ? W - 1 : col;
    return row * W + col;
}

// which of a zero point's three possible cells: by atan2f of its signed zeros (0: +-0, 1: pi, 2: -pi)
__device__ __forceinline__ int zero_kind(float at) { return at > 1.0f ? 1 : at < -1.0f ? 2 : 0; }

// the three cells a zero point (r = 0) can fall in, by the signs of its zeros through atan2f (zero_kind order)
__device__ __forceinline__ void zero_cells(int H, int W, float az_res, float vert_res, float vert_off, int (&zc)[3])
{
    zc[0] = cell_of_point(0.0f, 0.0f, 0.0f, H, W, az_res, vert_res, vert_off);
    zc[1] = cell_of_point(atan2f(0.0f, -0.0f), 0.0f, 0.0f, H, W, az_res, vert_res, vert_off);
    zc[2] = cell_of_point(atan2f(-0.0f, -0.0f), 0.0f, 0.0f, H, W, az_res, vert_res, vert_off);
}

// Side job of a launch in front of a projection (elo_pose_head_args.clear_* on the softmax partial-sums kernel,
// elo_mlp_args.clear_* on the row-wise MLP): its workgroups also clear the projection's buffers, which saves that
// call's init launch (three per forward).
constexpr int ZFLAGS = 4;                          // projection scratch, per image: one flag per cell a zero point can fall in (3 used)
struct ProjectionClear { unsigned *minr; float *xyz; unsigned *feat; long cells; int C; int images; };   // minr: cells + images words; C: 32-bit words of features per cell

// n 32-bit words at p <- v by the whole launch (thread `me` of `nthreads`): 16-byte stores where p is 16-byte aligned (every
// tensor a ProjectionBuffers allocates is), the last n % 4 words one by one
__device__ __forceinline__ void fill_words(unsigned *p, unsigned n, unsigned v, unsigned me, unsigned nthreads)
{
    if (((uintptr_t)p & 15) == 0) {
        const unsigned q = n >> 2;
        uint4 *p4 = reinterpret_cast<uint4 *>(p);
        for (unsigned i = me; i < q; i += nthreads) p4[i] = uint4{v, v, v, v};
        if (me < (n & 3u)) p[(q << 2) + me] = v;
    } else {
        for (unsigned i = me; i < n; i += nthreads) p[i] = v;
    }
}

// (round 5: one word per store and a three-way branch per word made this side job ~2 us of the 8 us launches it rides on -- 60 k
//  words over the 512 threads of a two-workgroup partial-sums launch; 16-byte stores, one region after the other)
__device__ __forceinline__ void clear_projection(const ProjectionClear &c)
{
    if (!c.minr) return;
    const unsigned nthreads = gridDim.x * gridDim.y * gridDim.z * blockDim.x;
    const unsigned me = ((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * blockDim.x + threadIdx.x;
    const unsigned cells = (unsigned)c.cells;
    fill_words(c.minr, cells + (unsigned)c.images * ZFLAGS, 0x7f7f7f7fu, me, nthreads);      // min range per cell + the zero-point flags behind them
    fill_words(reinterpret_cast<unsigned *>(c.xyz), cells * 3u, 0u, me, nthreads);
    if (c.C) fill_words(c.feat, cells * (unsigned)c.C, 0u, me, nthreads);
}

}  // namespace elo
