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

__device__ __forceinline__ void clear_projection(const ProjectionClear &c)
{
    if (!c.minr) return;
    const long n_xyz = c.cells * 3, total = c.cells + n_xyz + c.cells * c.C;
    const long nthreads = (long)gridDim.x * gridDim.y * gridDim.z * blockDim.x;
    const long me = (((long)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * blockDim.x + threadIdx.x;
    for (long i = me; i < (long)c.images * ZFLAGS; i += nthreads) c.minr[c.cells + i] = 0x7f7f7f7fu;   // the zero-point flags
    for (long i = me; i < total; i += nthreads) {
        if (i < c.cells) c.minr[i] = 0x7f7f7f7fu;
        else if (i < c.cells + n_xyz) c.xyz[i - c.cells] = 0.0f;
        else c.feat[i - c.cells - n_xyz] = 0u;
    }
}

}  // namespace elo
