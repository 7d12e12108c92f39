// elo_train.hip -- the row reductions of a TRAINING 1x1 convolution + batch norm (+ ReLU), gfx950.
//
// A training layer of the reference is conv2d -> batch norm with BATCH statistics -> ReLU (utils/tf_util.py:120-185,
// :512-563) on a (rows, C) matrix with rows = B*N*K up to ~10^6 and C = 8 ... 256.  Its dense products are small;
// what costs time are the reductions OVER THE ROWS: the batch moments, the two sums batch norm's backward needs, and the
// weight gradient X^T dZ -- in the library kernels torch dispatches to for these shapes 4.8 + 5.1 + ~14 ms of a 49 ms
// training step at batch 8 (tools/train_kernel_stats.py).  All of them stream (rows, C) once; the kernels below do that
// at HBM speed:
//   elo_bn_stats            sum, sum of squares per channel (per-block fp32 partials, combined in fp64 in a fixed order),
//                           then mean / 1/sqrt(var + eps) and the moving-average update of F.batch_norm(training=True)
//   elo_bn_apply            y = act(gamma * (z - mean) * invstd + beta)
//   elo_bn_backward_reduce  g = dy * [pre-activation > 0];  sum g, sum g * xhat per channel  (= d beta, d gamma)
//   elo_bn_backward_apply   dz = gamma * invstd * (g - sum_g / M - xhat * sum_gx / M)
//   elo_dense_weight_grad   dW = X^T dZ on v_mfma_f32_16x16x4_f32 (fp32 operands, fp32 accumulate), db = column sums of dZ
// C must be a power of two in 4 .. 256 for the four batch-norm kernels (every batch-normalised width of the model).
// No atomics in THIS file: every reduction is per-block partials + a fixed-order combine, so these kernels give the same
// bits on every run (a whole training step does not: the scatter adjoints of elo_backward.hip -- group_concat, cv_encode1/2,
// warp_project -- add with float atomics in run-dependent order), and nothing serialises on one L2 address (the first
// version added fp64 / fp32 atomics per block: 1800 blocks x 16 same-address atomics made bn_stats 46 us a call and the
// weight gradient 154 us, no faster than the library kernels they replaced).
#include <hip/hip_runtime.h>

#include <initializer_list>
#include "elo_common.h"

namespace elo {
namespace {

constexpr int TB = 256;                 // threads per block
constexpr int ROW_UNROLL = 8;           // 16-byte loads in flight per thread and tensor

__device__ __forceinline__ float4 ld4(const float *p, long i) { return reinterpret_cast<const float4 *>(p)[i]; }

// block-wide reduction of per-thread float4 partials that belong to column group (tid % q): the TB / q threads of a
// column group add up through LDS, the first q threads return the totals
__device__ __forceinline__ float4 reduce_groups(float4 v, int q, float4 *lds)
{
    const int tid = threadIdx.x;
    __syncthreads();
    lds[tid] = v;
    __syncthreads();
    float4 t{0.f, 0.f, 0.f, 0.f};
    if (tid < q)
        for (int i = tid; i < TB; i += q) { const float4 o = lds[i]; t.x += o.x; t.y += o.y; t.z += o.z; t.w += o.w; }
    return t;
}

// partial sums of block b: part[b][0..C) and part[b][C..2C)
__device__ __forceinline__ void put4(float *part, int C, int which, int cg, const float4 v)
{
    reinterpret_cast<float4 *>(part + ((size_t)blockIdx.x * 2 + which) * C)[cg] = v;
}

// channel c's two totals over `parts` partial rows, in fp64, by one wave (fixed order): valid in every lane
__device__ __forceinline__ void combine(const float *part, int parts, int C, int c, double &t0, double &t1)
{
    const int lane = threadIdx.x & 63;
    double a = 0.0, b = 0.0;
    for (int i0 = lane; i0 < parts; i0 += 64 * 8) {        // 16 loads in flight (one dependent trip per 64 partial rows before:
        float va[8], vb[8];                                //  a 2048-part reduction was 32 round trips behind each other)
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = i0 + 64 * u, ic = i < parts ? i : lane;
            va[u] = part[((size_t)ic * 2) * C + c]; vb[u] = part[((size_t)ic * 2 + 1) * C + c];
            if (i >= parts) { va[u] = 0.f; vb[u] = 0.f; }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) { a += (double)va[u]; b += (double)vb[u]; }
    }
    for (int d = 32; d >= 1; d >>= 1) { a += __shfl_xor(a, d, 64); b += __shfl_xor(b, d, 64); }
    t0 = a; t1 = b;
}

// rows [r0, r1) of this block, a multiple of the rows one block iteration covers
__device__ __forceinline__ void block_rows(long M, int rpb, long &r0, long &r1)
{
    const long iters = (M + rpb - 1) / rpb, per = (iters + gridDim.x - 1) / gridDim.x;
    r0 = (long)blockIdx.x * per * rpb;
    r1 = r0 + per * rpb < M ? r0 + per * rpb : M;
}

// (blockIdx.y: the GROUP of rows -- M rows each, one set of moments per group: the two frames of a Siamese batch are one launch)
__global__ __launch_bounds__(TB) void bn_stats_kernel(const float *__restrict__ z, long M, int C, float *__restrict__ part)
{
    __shared__ float4 lds[TB];
    z += (size_t)blockIdx.y * M * C;
    part += (size_t)blockIdx.y * gridDim.x * 2 * C;
    const int q = C >> 2, rpb = TB / q, cg = threadIdx.x % q, rr = threadIdx.x / q;
    long r0, r1;
    block_rows(M, rpb, r0, r1);
    float4 s{0.f, 0.f, 0.f, 0.f}, ss{0.f, 0.f, 0.f, 0.f};
    for (long r = r0 + rr; r < r1; r += (long)rpb * ROW_UNROLL) {
        float4 v[ROW_UNROLL];
#pragma unroll
        for (int u = 0; u < ROW_UNROLL; ++u) {
            const long ru = r + (long)u * rpb;
            v[u] = ld4(z, (ru < r1 ? ru : r) * q + cg);
            if (ru >= r1) v[u] = float4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int u = 0; u < ROW_UNROLL; ++u) {
            s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w;
            ss.x += v[u].x * v[u].x; ss.y += v[u].y * v[u].y; ss.z += v[u].z * v[u].z; ss.w += v[u].w * v[u].w;
        }
    }
    const float4 ts = reduce_groups(s, q, lds), tss = reduce_groups(ss, q, lds);
    if (threadIdx.x < q) { put4(part, C, 0, cg, ts); put4(part, C, 1, cg, tss); }
}

// one wave per channel: moments from the partial sums (fp64), F.batch_norm's moving averages (unbiased variance)
__global__ __launch_bounds__(64) void bn_finalize_kernel(const float *__restrict__ part, int parts, long M, int C, float eps,
                                                        float momentum, float *__restrict__ mean, float *__restrict__ invstd,
                                                        float *__restrict__ running_mean, float *__restrict__ running_var, int groups)
{
    const int c = blockIdx.x;
    double rm = 0.0, rv = 0.0;
    if (running_mean && threadIdx.x == 0) { rm = (double)running_mean[c]; rv = (double)running_var[c]; }
    for (int g = 0; g < groups; ++g) {                 // the moving averages take the groups' moments one after the other, in order
        double s, ss;
        combine(part + (size_t)g * parts * 2 * C, parts, C, c, s, ss);
        if (threadIdx.x != 0) continue;
        const double m = s / (double)M;
        double var = ss / (double)M - m * m;
        var = var > 0.0 ? var : 0.0;
        mean[(size_t)g * C + c] = (float)m;
        invstd[(size_t)g * C + c] = (float)(1.0 / sqrt(var + (double)eps));
        if (running_mean) {
            const double unbiased = M > 1 ? var * (double)M / (double)(M - 1) : var;
            // (rounded to fp32 after every group: exactly what one call per group leaves behind)
            rm = (double)(float)((1.0 - (double)momentum) * rm + (double)momentum * m);
            rv = (double)(float)((1.0 - (double)momentum) * rv + (double)momentum * unbiased);
        }
    }
    if (running_mean && threadIdx.x == 0) { running_mean[c] = (float)rm; running_var[c] = (float)rv; }
}

// one wave per channel: sums[c] = sum g, sums[C + c] = sum g * xhat  (= d beta, d gamma)
__global__ __launch_bounds__(64) void bn_bwd_combine_kernel(const float *__restrict__ part, int parts, int C, float *__restrict__ sums)
{
    const int c = blockIdx.x;
    part += (size_t)blockIdx.y * parts * 2 * C;
    sums += (size_t)blockIdx.y * 2 * C;
    double a, b;
    combine(part, parts, C, c, a, b);
    if (threadIdx.x == 0) { sums[c] = (float)a; sums[C + c] = (float)b; }
}

// (C is a power of two <= 256 (check_bn), so q = C / 4 divides the block and the grid stride: a thread keeps ONE channel quad for
//  its whole life.  The first form took `i % q` -- a 64-bit modulo, ~150 instructions -- and re-read the four per-channel
//  constants for every 16-byte element, one element in flight per thread: tools/isa_by_line.py.  Same arithmetic.)
constexpr int BN_APPLY_UNROLL = 4;                  // elements in flight per thread

__global__ __launch_bounds__(TB) void bn_apply_kernel(const float *__restrict__ z, long n4, int C, const float *__restrict__ mean,
                                                      const float *__restrict__ invstd, const float *__restrict__ gamma,
                                                      const float *__restrict__ beta, int relu, float *__restrict__ y)
{
    const int cg = threadIdx.x & ((C >> 2) - 1);
    z += (size_t)blockIdx.y * n4 * 4; y += (size_t)blockIdx.y * n4 * 4;
    mean += (size_t)blockIdx.y * C; invstd += (size_t)blockIdx.y * C;
    const float4 m = ld4(mean, cg), s = ld4(invstd, cg), g = ld4(gamma, cg), b = ld4(beta, cg);
    const long stride = (long)gridDim.x * TB;
    auto one = [&](const float4 v) {
        float4 o{(v.x - m.x) * s.x * g.x + b.x, (v.y - m.y) * s.y * g.y + b.y, (v.z - m.z) * s.z * g.z + b.z,
                 (v.w - m.w) * s.w * g.w + b.w};
        if (relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
        return o;
    };
    long i = (long)blockIdx.x * TB + threadIdx.x;
    for (; i + (BN_APPLY_UNROLL - 1) * stride < n4; i += BN_APPLY_UNROLL * stride) {
        float4 v[BN_APPLY_UNROLL];
#pragma unroll
        for (int u = 0; u < BN_APPLY_UNROLL; ++u) v[u] = ld4(z, i + u * stride);
#pragma unroll
        for (int u = 0; u < BN_APPLY_UNROLL; ++u) reinterpret_cast<float4 *>(y)[i + u * stride] = one(v[u]);
    }
    for (; i < n4; i += stride) reinterpret_cast<float4 *>(y)[i] = one(ld4(z, i));
}

struct BnCol { float4 m, s, g, b; };

__device__ __forceinline__ void bn_back(const float4 dy, const float4 z, const BnCol &p, int relu, float4 &g, float4 &xh)
{
    xh = float4{(z.x - p.m.x) * p.s.x, (z.y - p.m.y) * p.s.y, (z.z - p.m.z) * p.s.z, (z.w - p.m.w) * p.s.w};
    g = dy;
    if (relu) {
        g.x = xh.x * p.g.x + p.b.x > 0.f ? dy.x : 0.f; g.y = xh.y * p.g.y + p.b.y > 0.f ? dy.y : 0.f;
        g.z = xh.z * p.g.z + p.b.z > 0.f ? dy.z : 0.f; g.w = xh.w * p.g.w + p.b.w > 0.f ? dy.w : 0.f;
    }
}

__global__ __launch_bounds__(TB) void bn_bwd_reduce_kernel(const float *__restrict__ dy, const float *__restrict__ z, long M, int C,
                                                           const float *__restrict__ mean, const float *__restrict__ invstd,
                                                           const float *__restrict__ gamma, const float *__restrict__ beta,
                                                           int relu, float *__restrict__ part)
{
    __shared__ float4 lds[TB];
    const int q = C >> 2, rpb = TB / q, cg = threadIdx.x % q, rr = threadIdx.x / q;
    dy += (size_t)blockIdx.y * M * C; z += (size_t)blockIdx.y * M * C;
    mean += (size_t)blockIdx.y * C; invstd += (size_t)blockIdx.y * C;
    part += (size_t)blockIdx.y * gridDim.x * 2 * C;
    const BnCol p{ld4(mean, cg), ld4(invstd, cg), ld4(gamma, cg), ld4(beta, cg)};
    long r0, r1;
    block_rows(M, rpb, r0, r1);
    float4 s1{0.f, 0.f, 0.f, 0.f}, s2{0.f, 0.f, 0.f, 0.f};
    constexpr int U = ROW_UNROLL / 2;
    for (long r = r0 + rr; r < r1; r += (long)rpb * U) {
        float4 a[U], b[U];
        bool live[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long ru = r + (long)u * rpb;
            live[u] = ru < r1;
            a[u] = ld4(dy, (live[u] ? ru : r) * q + cg);
            b[u] = ld4(z, (live[u] ? ru : r) * q + cg);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float4 g, xh;
            bn_back(a[u], b[u], p, relu, g, xh);
            if (!live[u]) g = float4{0.f, 0.f, 0.f, 0.f};
            s1.x += g.x; s1.y += g.y; s1.z += g.z; s1.w += g.w;
            s2.x += g.x * xh.x; s2.y += g.y * xh.y; s2.z += g.z * xh.z; s2.w += g.w * xh.w;
        }
    }
    const float4 t1 = reduce_groups(s1, q, lds), t2 = reduce_groups(s2, q, lds);
    if (threadIdx.x < q) { put4(part, C, 0, cg, t1); put4(part, C, 1, cg, t2); }
}

__global__ __launch_bounds__(TB) void bn_bwd_apply_kernel(const float *__restrict__ dy, const float *__restrict__ z, long n4, long M,
                                                          int C, const float *__restrict__ mean, const float *__restrict__ invstd,
                                                          const float *__restrict__ gamma, const float *__restrict__ beta, int relu,
                                                          const float *__restrict__ sums, float *__restrict__ dz)
{
    const int cg = threadIdx.x & ((C >> 2) - 1);     // (one channel quad per thread: see bn_apply_kernel)
    dy += (size_t)blockIdx.y * n4 * 4; z += (size_t)blockIdx.y * n4 * 4; dz += (size_t)blockIdx.y * n4 * 4;
    mean += (size_t)blockIdx.y * C; invstd += (size_t)blockIdx.y * C; sums += (size_t)blockIdx.y * 2 * C;
    const long stride = (long)gridDim.x * TB;
    const float inv_m = 1.0f / (float)M;
    const BnCol p{ld4(mean, cg), ld4(invstd, cg), ld4(gamma, cg), ld4(beta, cg)};
    const float4 a1 = ld4(sums, cg), a2 = ld4(sums + C, cg);
    const float4 k1{a1.x * inv_m, a1.y * inv_m, a1.z * inv_m, a1.w * inv_m}, k2{a2.x * inv_m, a2.y * inv_m, a2.z * inv_m, a2.w * inv_m};
    auto one = [&](const float4 dyv, const float4 zv) {
        float4 g, xh;
        bn_back(dyv, zv, p, relu, g, xh);
        return float4{p.g.x * p.s.x * (g.x - k1.x - xh.x * k2.x), p.g.y * p.s.y * (g.y - k1.y - xh.y * k2.y),
                      p.g.z * p.s.z * (g.z - k1.z - xh.z * k2.z), p.g.w * p.s.w * (g.w - k1.w - xh.w * k2.w)};
    };
    long i = (long)blockIdx.x * TB + threadIdx.x;
    for (; i + (BN_APPLY_UNROLL - 1) * stride < n4; i += BN_APPLY_UNROLL * stride) {
        float4 a[BN_APPLY_UNROLL], b[BN_APPLY_UNROLL];
#pragma unroll
        for (int u = 0; u < BN_APPLY_UNROLL; ++u) { a[u] = ld4(dy, i + u * stride); b[u] = ld4(z, i + u * stride); }
#pragma unroll
        for (int u = 0; u < BN_APPLY_UNROLL; ++u) reinterpret_cast<float4 *>(dz)[i + u * stride] = one(a[u], b[u]);
    }
    for (; i < n4; i += stride) reinterpret_cast<float4 *>(dz)[i] = one(ld4(dy, i), ld4(z, i));
}

// ---- dW = X^T G ------------------------------------------------------------------------------------------------------
// v_mfma_f32_16x16x4_f32: A[i][k] in lane (i = l % 16, k = l / 16), B[k][j] in lane (j = l % 16, k = l / 16),
// D[4 * (l / 16) + r][l % 16] in acc[r].  With A = X^T (i: input channel, k: one of 4 rows) and B = G (j: output channel)
// a step consumes 4 rows.  dW is cut into blocks of (16 CI x 16 CO); a WAVE owns one block over one SLICE of the rows
// (blockIdx.x), and the (up to four) waves of a workgroup own different blocks over THE SAME slice: they stream the same
// rows of X and G at the same time, so those come from HBM once and from the CU's L1 afterwards (the first form gave
// each workgroup one block: X was read once per column block and G once per row block -- 354 MB instead of 177 MB for a
// 128 -> 128 layer, at 1.8 TB/s the whole 200 us).  No reduction inside the workgroup: a wave stores its block of the
// slice's partial dW; slices_combine_kernel sums the slices in a fixed order.  Loads are 4 bytes per lane (4 rows x 16
// consecutive channels per instruction), U row groups in flight.
// Round 6, two changes:
//  * 16-BYTE LOADS.  A lane used to load ONE channel of a row for each of its CI (CO) tiles; now (XV / GV: the block lies inside
//    the matrix and its width allows aligned vectors) it loads CI (CO) CONSECUTIVE channels of its row in one instruction and uses
//    component a as the operand of tile a -- tile a then holds the channels ci0 + CI * i + a instead of ci0 + 16 a + i: a permutation
//    of dW's rows (columns) inside the block that only the final store has to know.  A quarter of the load instructions, whole
//    256-byte row segments per quarter-wave.
//  * THE NEXT TRIP'S LOADS ARE IN FLIGHT DURING THIS TRIP'S MFMAs (two register sets, the trip loop unrolled twice; loads are
//    unconditional -- clamped to valid rows -- because the compiler cannot count loads issued behind a branch and waits for all of
//    them; dead rows are weighted out where a ragged trip is USED).  Before, every trip of 16 rows began with a full HBM round trip.
//    (A third register set -- two trips ahead -- measured the same or slower: tools/wgrad_micro.py.)
template <int W> struct VecOf;
template <> struct VecOf<1> { typedef float T; };
template <> struct VecOf<2> { typedef float2 T; };
template <> struct VecOf<4> { typedef float4 T; };
template <> struct VecOf<3> { typedef float T; };      // (never loaded as a vector)

template <int CI, int CO, bool XV, bool GV>
__global__ __launch_bounds__(TB, 2) void weight_grad_kernel(const float *__restrict__ x, const float *__restrict__ g, long M, int Cin,
                                                         int Cout, int gy, int gz, float *__restrict__ part, float *__restrict__ bpart)
{
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), i16 = lane & 15, k4 = lane >> 4;
    const int tb = blockIdx.y * (blockDim.x >> 6) + wave;            // this wave's block of dW
    if (tb >= gy * gz) return;
    const int ci0 = (tb / gz) * CI * 16, co0 = (tb % gz) * CO * 16;
    const long groups = (M + 3) / 4, per = (groups + gridDim.x - 1) / gridDim.x;
    const long g0 = (long)blockIdx.x * per, g1 = g0 + per < groups ? g0 + per : groups;
    f32x4 acc[CI][CO];
#pragma unroll
    for (int a = 0; a < CI; ++a)
#pragma unroll
        for (int b = 0; b < CO; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bsum[CO];
#pragma unroll
    for (int b = 0; b < CO; ++b) bsum[b] = 0.f;
    // column offsets of this lane's operands (scalar form: one per tile, clamped into the matrix -- a clamped column feeds only rows
    // / columns of the block that are never stored)
    int xc[CI], gc[CO];
#pragma unroll
    for (int a = 0; a < CI; ++a) { const int c = XV ? ci0 + CI * i16 + a : ci0 + a * 16 + i16; xc[a] = c < Cin ? c : 0; }
#pragma unroll
    for (int b = 0; b < CO; ++b) { const int c = GV ? co0 + CO * i16 + b : co0 + b * 16 + i16; gc[b] = c < Cout ? c : 0; }
    constexpr int U = CI * CO >= 8 ? 4 : 8;                        // row groups per trip (narrow layers are load-latency bound)
    auto load = [&](float (&xv)[U][CI], float (&gv)[U][CO], long gr) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            // (32-bit element offsets -- the launcher bounds rows * width by 2^31 --: a min, a multiply-add and the scaled-offset
            //  addressing mode instead of ~12 64-bit VALU instructions per load, ~100 per trip beside its 64 MFMAs)
            const unsigned row = (unsigned)(gr + u) * 4u + (unsigned)k4, rc = row < (unsigned)M ? row : (unsigned)M - 1u;
            if (XV) {
                const typename VecOf<CI>::T v = *reinterpret_cast<const typename VecOf<CI>::T *>(x + (rc * (unsigned)Cin + (unsigned)xc[0]));
                const float *f = reinterpret_cast<const float *>(&v);
#pragma unroll
                for (int a = 0; a < CI; ++a) xv[u][a] = f[a];
            } else {
#pragma unroll
                for (int a = 0; a < CI; ++a) xv[u][a] = x[rc * (unsigned)Cin + (unsigned)xc[a]];
            }
            if (GV) {
                const typename VecOf<CO>::T v = *reinterpret_cast<const typename VecOf<CO>::T *>(g + (rc * (unsigned)Cout + (unsigned)gc[0]));
                const float *f = reinterpret_cast<const float *>(&v);
#pragma unroll
                for (int b = 0; b < CO; ++b) gv[u][b] = f[b];
            } else {
#pragma unroll
                for (int b = 0; b < CO; ++b) gv[u][b] = g[rc * (unsigned)Cout + (unsigned)gc[b]];
            }
        }
    };
    auto use = [&](float (&xv)[U][CI], float (&gv)[U][CO], long gr) {
        if (gr + U > g1 || (gr + U) * 4 > M) {                    // a ragged trip: rows past the slice or the matrix count for nothing
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const float w = gr + u < g1 && (gr + u) * 4 + k4 < M ? 1.f : 0.f;
#pragma unroll
                for (int b = 0; b < CO; ++b) gv[u][b] *= w;
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int a = 0; a < CI; ++a)
#pragma unroll
                for (int b = 0; b < CO; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(xv[u][a], gv[u][b], acc[a][b], 0, 0, 0);
#pragma unroll
            for (int b = 0; b < CO; ++b) bsum[b] += gv[u][b];
        }
    };
    float xa[U][CI], ga[U][CO], xb[U][CI], gb[U][CO];
    load(xa, ga, g0);
#pragma unroll 1
    for (long gr = g0; gr < g1; gr += 2 * U) {
        load(xb, gb, gr + U);
        __builtin_amdgcn_sched_barrier(0);         // the loads go out before the MFMAs that hide them
        use(xa, ga, gr);
        __builtin_amdgcn_sched_barrier(0);
        load(xa, ga, gr + 2 * U);
        __builtin_amdgcn_sched_barrier(0);
        if (gr + U < g1) use(xb, gb, gr + U);
        __builtin_amdgcn_sched_barrier(0);
    }
    float *dst = part + (size_t)blockIdx.x * Cin * Cout;
#pragma unroll
    for (int a = 0; a < CI; ++a)
#pragma unroll
        for (int b = 0; b < CO; ++b) {
            const int co = GV ? co0 + CO * i16 + b : co0 + b * 16 + i16;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = 4 * k4 + r, ci = XV ? ci0 + CI * i + a : ci0 + a * 16 + i;
                if (ci < Cin && co < Cout) dst[(size_t)ci * Cout + co] = acc[a][b][r];
            }
        }
    if (bpart && ci0 == 0) {
#pragma unroll
        for (int b = 0; b < CO; ++b) {
            float v = bsum[b];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            const int co = GV ? co0 + CO * i16 + b : co0 + b * 16 + i16;
            if (k4 == 0 && co < Cout) bpart[(size_t)blockIdx.x * Cout + co] = v;
        }
    }
}

// out[i] = sum over slices of part[slice][i] (n = Cin*Cout for dW, Cout for db).  A block owns 16 outputs; its 16 thread
// rows take every 16th slice each (8 loads in flight), then add up through LDS in row order: up to 1024 slices of a
// narrow layer are 8 round trips instead of 1024 dependent ones.
// (one launch for both results: the first `blocks_w` blocks sum dW, the rest db)
__global__ __launch_bounds__(TB) void slices_combine_kernel(const float *__restrict__ part_w, int slices, long n_w, float *__restrict__ out_w,
                                                            unsigned blocks_w, const float *__restrict__ part_b, long n_b,
                                                            float *__restrict__ out_b)
{
    __shared__ float red[16][17];
    const bool second = blockIdx.x >= blocks_w;
    const float *__restrict__ part = second ? part_b : part_w;
    const long n = second ? n_b : n_w;
    float *__restrict__ out = second ? out_b : out_w;
    const int j = threadIdx.x & 15, sr = threadIdx.x >> 4;
    const long i = (long)(second ? blockIdx.x - blocks_w : blockIdx.x) * 16 + j;
    const long ic = i < n ? i : n - 1;
    float v = 0.f;
    for (int s = sr; s < slices; s += 16 * 8) {
        float t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int su = s + 16 * u; t[u] = part[(size_t)(su < slices ? su : s) * n + ic]; if (su >= slices) t[u] = 0.f; }
#pragma unroll
        for (int u = 0; u < 8; ++u) v += t[u];
    }
    red[sr][j] = v;
    __syncthreads();
    if (sr == 0 && i < n) {
        float t = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) t += red[r][j];
        out[i] = t;
    }
}

bool pow2_width(int C) { return C >= 4 && C <= 256 && (C & (C - 1)) == 0; }

int grid_for(long M, int C)
{
    const int rpb = TB / (C >> 2);
    const long iters = (M + rpb - 1) / rpb;
    const long want = (iters + ROW_UNROLL - 1) / ROW_UNROLL;        // at least one unrolled trip per block
    return (int)(want < 1 ? 1 : want > ELO_BN_MAX_PARTS ? ELO_BN_MAX_PARTS : want);
}

int check_bn(const char *who, long M, int C, std::initializer_list<const void *> ptrs)
{
    if (M <= 0) return fail(ELO_ERR_ARG, "%s: no rows", who);
    if (!pow2_width(C)) return fail(ELO_ERR_LIMIT, "%s: C = %d is not a power of two in 4..256", who, C);
    for (const void *p : ptrs)
        if (!p || ((uintptr_t)p & 15)) return fail(ELO_ERR_ARG, "%s: null or unaligned tensor pointer", who);
    return ELO_OK;
}

}  // namespace

void bn_finalize_launch(const float *part, int parts, long M, int C, float eps, float momentum, float *mean, float *invstd,
                        float *running_mean, float *running_var, int groups, hipStream_t s)      // for elo_train_dense.hip's fused moments
{
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(C), dim3(64), 0, s, part, parts, M, C, eps, momentum, mean, invstd, running_mean, running_var, groups);
}
}  // namespace elo

using namespace elo;

extern "C" long elo_bn_scratch_floats(int C, int groups) { return 2l * C * ELO_BN_MAX_PARTS * (groups > 1 ? groups : 1); }

// rows per group (groups <= 1: one group), or -1
static long group_rows(const char *who, long rows, int groups)
{
    const int g = groups > 1 ? groups : 1;
    if (g > 64 || rows % g) { fail(ELO_ERR_ARG, "%s: %ld rows do not split into %d groups", who, rows, groups); return -1; }
    return rows / g;
}

extern "C" int elo_bn_stats(const elo_bn_stats_args *a, elo_stream_t stream)
{
    const char *who = "elo_bn_stats";
    if (!a) return fail(ELO_ERR_ARG, "%s: null argument block", who);
    if (int rc = check_bn(who, a->rows, a->C, {a->z, a->scratch, a->mean, a->invstd})) return rc;
    if ((a->running_mean == nullptr) != (a->running_var == nullptr)) return fail(ELO_ERR_ARG, "%s: running_mean and running_var go together", who);
    hipStream_t s = (hipStream_t)stream;
    const int G = a->groups > 1 ? a->groups : 1;
    const long Mg = group_rows(who, a->rows, a->groups);
    if (Mg < 0) return ELO_ERR_ARG;
    const int parts = grid_for(Mg, a->C);
    hipLaunchKernelGGL(bn_stats_kernel, dim3(parts, G), dim3(TB), 0, s, a->z, Mg, a->C, a->scratch);
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(a->C), dim3(64), 0, s, a->scratch, parts, Mg, a->C, a->eps, a->momentum,
                       a->mean, a->invstd, a->running_mean, a->running_var, G);
    return check_launch(who);
}

extern "C" int elo_bn_apply(const elo_bn_apply_args *a, elo_stream_t stream)
{
    const char *who = "elo_bn_apply";
    if (!a) return fail(ELO_ERR_ARG, "%s: null argument block", who);
    if (int rc = check_bn(who, a->rows, a->C, {a->z, a->mean, a->invstd, a->gamma, a->beta, a->y})) return rc;
    const int G = a->groups > 1 ? a->groups : 1;
    const long Mg = group_rows(who, a->rows, a->groups);
    if (Mg < 0) return ELO_ERR_ARG;
    const long n4 = Mg * (a->C >> 2);
    const long blocks = (n4 + TB * 4 - 1) / (TB * 4);
    hipLaunchKernelGGL(bn_apply_kernel, dim3((unsigned)(blocks > 8192 ? 8192 : blocks), G), dim3(TB), 0, (hipStream_t)stream, a->z, n4, a->C,
                       a->mean, a->invstd, a->gamma, a->beta, a->relu, a->y);
    return check_launch(who);
}

extern "C" int elo_bn_backward(const elo_bn_backward_args *a, elo_stream_t stream)
{
    const char *who = "elo_bn_backward";
    if (!a) return fail(ELO_ERR_ARG, "%s: null argument block", who);
    if (int rc = check_bn(who, a->rows, a->C, {a->dy, a->z, a->mean, a->invstd, a->gamma, a->beta, a->scratch, a->sums, a->dz ? a->dz : a->sums})) return rc;
    hipStream_t s = (hipStream_t)stream;
    const int G = a->groups > 1 ? a->groups : 1;
    const long Mg = group_rows(who, a->rows, a->groups);
    if (Mg < 0) return ELO_ERR_ARG;
    const int parts = grid_for(Mg, a->C);
    hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(parts, G), dim3(TB), 0, s, a->dy, a->z, Mg, a->C, a->mean, a->invstd,
                       a->gamma, a->beta, a->relu, a->scratch);
    hipLaunchKernelGGL(bn_bwd_combine_kernel, dim3(a->C, G), dim3(64), 0, s, a->scratch, parts, a->C, a->sums);
    if (!a->dz) return check_launch(who);
    const long n4 = Mg * (a->C >> 2);
    const long blocks = (n4 + TB * 4 - 1) / (TB * 4);
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3((unsigned)(blocks > 8192 ? 8192 : blocks), G), dim3(TB), 0, s, a->dy, a->z, n4, Mg, a->C,
                       a->mean, a->invstd, a->gamma, a->beta, a->relu, a->sums, a->dz);
    return check_launch(who);
}

// row slices of the weight gradient: a slice is one workgroup's rows (its waves own different blocks of dW): enough
// slices to fill the GPU and hide the load latency of the narrow layers, few enough that the partial blocks stay
// <= 32 MB and every slice has >= 32 row groups
extern "C" int elo_weight_grad_slices(long rows, int Cin, int Cout)
{
    const long groups = (rows + 3) / 4;
    long s = groups / 32 + 1;
    const long cap = (32l << 20) / ((long)Cin * Cout * 4);
    s = s > 2048 ? 2048 : s;
    s = s > cap ? cap : s;
    return (int)(s < 1 ? 1 : s);
}

extern "C" int elo_dense_weight_grad(const elo_weight_grad_args *a, elo_stream_t stream)
{
    const char *who = "elo_dense_weight_grad";
    if (!a) return fail(ELO_ERR_ARG, "%s: null argument block", who);
    if (a->rows <= 0 || a->Cin <= 0 || a->Cout <= 0) return fail(ELO_ERR_ARG, "%s: bad sizes", who);
    if (!a->x || !a->g || !a->dW || !a->scratch) return fail(ELO_ERR_ARG, "%s: null tensor pointer", who);
    if (a->rows * (long)(a->Cin > a->Cout ? a->Cin : a->Cout) >= (1l << 31)) return fail(ELO_ERR_LIMIT, "%s: rows * width >= 2^31 (32-bit element offsets)", who);
    const int cit = (a->Cin + 15) / 16, cot = (a->Cout + 15) / 16;
    const int slices = elo_weight_grad_slices(a->rows, a->Cin, a->Cout);
    float *bpart = a->db ? a->scratch + (size_t)slices * a->Cin * a->Cout : nullptr;
    hipStream_t s = (hipStream_t)stream;
    // blocks of dW: all column tiles of a <= 64-wide output (1, 2 or 4); row tiles: the count in 1..4 that pads the input
    // width least, larger on a tie or when the padding stays <= 25 % (138 channels = 9 tiles: 3 x 3; 80 = 5 tiles: 2 x 3)
    const int CO = cot >= 4 ? 4 : cot >= 2 ? 2 : 1;
    int CI = 1;
    for (int c = 2; c <= 4; ++c) {
        const int padded = (cit + c - 1) / c * c;
        if (padded <= (cit + CI - 1) / CI * CI || padded * 4 <= cit * 5) CI = c;
    }
    const int gy = (cit + CI - 1) / CI, gz = (cot + CO - 1) / CO, nb = gy * gz, wpb = nb < TB / 64 ? nb : TB / 64;
    const dim3 grid((unsigned)slices, (unsigned)((nb + wpb - 1) / wpb));
    // vector loads: the tile count is a vector width (1, 2, 4), every block of the grid is whole, rows stay 16-byte aligned
    const bool xv = CI != 3 && a->Cin % (16 * CI) == 0 && ((uintptr_t)a->x & 15) == 0;
    const bool gv = a->Cout % (16 * CO) == 0 && ((uintptr_t)a->g & 15) == 0;
#define ELO_WG4(CI_, CO_, XV_, GV_)                                                                                           \
    hipLaunchKernelGGL((weight_grad_kernel<CI_, CO_, XV_, GV_>), grid, dim3(64 * wpb), 0, s, a->x, a->g, a->rows, a->Cin, a->Cout, gy, gz, \
                       a->scratch, bpart)
#define ELO_WG(CI_, CO_)                                                                                                      \
    if (CI == CI_ && CO == CO_) {                                                                                             \
        if (xv && gv) ELO_WG4(CI_, CO_, (CI_ != 3), true);                                                                    \
        else if (gv) ELO_WG4(CI_, CO_, false, true);                                                                          \
        else if (xv) ELO_WG4(CI_, CO_, (CI_ != 3), false);                                                                    \
        else ELO_WG4(CI_, CO_, false, false);                                                                                 \
    }
    ELO_WG(1, 1); ELO_WG(1, 2); ELO_WG(1, 4); ELO_WG(2, 1); ELO_WG(2, 2); ELO_WG(2, 4);
    ELO_WG(3, 1); ELO_WG(3, 2); ELO_WG(3, 4); ELO_WG(4, 1); ELO_WG(4, 2); ELO_WG(4, 4);
#undef ELO_WG4
#undef ELO_WG
    const long n = (long)a->Cin * a->Cout;
    const unsigned bw = (unsigned)((n + 15) / 16), bb = a->db ? (unsigned)((a->Cout + 15) / 16) : 0u;
    hipLaunchKernelGGL(slices_combine_kernel, dim3(bw + bb), dim3(TB), 0, s, a->scratch, slices, n, a->dW, bw, bpart, (long)a->Cout, a->db);
    return check_launch(who);
}

// ================================================================ Adam over ONE flat parameter buffer
// main.py:171-176 (tf.train.AdamOptimizer): all 382 variables of the model are views of one contiguous fp32 buffer (and so
// are their gradients -- distributed.FlatGradBucket -- and the two moment buffers), so an optimiser step is ONE elementwise
// launch over 899 134 floats instead of torch's ~46 multi-tensor launches.  The step's scalars (learning rate over the
// first-moment bias correction, the second-moment bias correction, epsilon) are computed by the host in double precision
// and handed over in a 4-float device buffer the captured graph reads (one async copy per step; it also replaces the
// learning-rate fill).  The arithmetic is torch.optim.Adam's (amsgrad off, no weight decay):
//     m = b1 m + (1 - b1) g;  v = b2 v + (1 - b2) g^2;  p -= (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
__global__ __launch_bounds__(256) void adam_flat_kernel(float *__restrict__ p, const float *__restrict__ g, float *__restrict__ m,
                                                        float *__restrict__ v, long n, const float *__restrict__ hyper, float b1, float b2,
                                                        float c1, float c2)      // c = 1 - b, rounded from the host's double (1.0f - 0.999f is 4.7e-5 off)
{
    const float step_size = hyper[0], inv_sqrt_bc2 = hyper[1], eps = hyper[2];
    const long i4 = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i4 >= n) return;
    if (i4 + 4 <= n) {
        const float4 gv = *reinterpret_cast<const float4 *>(g + i4);
        float4 mv = *reinterpret_cast<float4 *>(m + i4), vv = *reinterpret_cast<float4 *>(v + i4), pv = *reinterpret_cast<float4 *>(p + i4);
        auto upd = [&](float &pp, float gg, float &mm, float &vq) {
            mm = b1 * mm + c1 * gg;
            vq = b2 * vq + c2 * gg * gg;
            pp -= step_size * (mm / (sqrtf(vq) * inv_sqrt_bc2 + eps));
        };
        upd(pv.x, gv.x, mv.x, vv.x); upd(pv.y, gv.y, mv.y, vv.y); upd(pv.z, gv.z, mv.z, vv.z); upd(pv.w, gv.w, mv.w, vv.w);
        *reinterpret_cast<float4 *>(m + i4) = mv; *reinterpret_cast<float4 *>(v + i4) = vv; *reinterpret_cast<float4 *>(p + i4) = pv;
    } else {
        for (long i = i4; i < n; ++i) {
            const float gg = g[i];
            const float mm = b1 * m[i] + c1 * gg, vq = b2 * v[i] + c2 * gg * gg;
            m[i] = mm; v[i] = vq;
            p[i] -= step_size * (mm / (sqrtf(vq) * inv_sqrt_bc2 + eps));
        }
    }
}

extern "C" int elo_adam_flat(const elo_adam_flat_args *a, elo_stream_t stream)
{
    const char *who = "elo_adam_flat";
    if (!a) return fail(ELO_ERR_ARG, "%s: null argument block", who);
    if (a->n < 0) return fail(ELO_ERR_ARG, "%s: bad size", who);
    if (a->n == 0) return ELO_OK;
    if (!a->param || !a->grad || !a->exp_avg || !a->exp_avg_sq || !a->hyper) return fail(ELO_ERR_ARG, "%s: null tensor pointer", who);
    if (((uintptr_t)a->param | (uintptr_t)a->grad | (uintptr_t)a->exp_avg | (uintptr_t)a->exp_avg_sq) % 16)
        return fail(ELO_ERR_ARG, "%s: the flat buffers must be 16-byte aligned", who);
    const long quads = (a->n + 3) / 4;
    hipLaunchKernelGGL(adam_flat_kernel, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a->param, a->grad, a->exp_avg,
                       a->exp_avg_sq, a->n, a->hyper, a->beta1, a->beta2, a->one_minus_beta1, a->one_minus_beta2);
    return check_launch(who);
}

// ================================================================ the per-pair pose algebra of a training step
// pwclo_model.py:197-208, :262-280 (pose head -> normalise -> compose with the coarse pose) and :437-481 (get_loss) are a
// few dozen scalars per frame pair, which torch runs as ~60 eight-element kernels per level forward and twice that
// backward: ~800 of a training step's ~2000 launches (tools/train_op_counts.py).  Here: one thread per batch element, the
// whole chain in registers, one launch forward and one backward (the adjoints are written out by hand, operator by operator,
// and checked against torch.autograd in float64: tests/test_train_kernels_gpu.py).
namespace {
struct Q4 { float v[4]; };
__device__ __forceinline__ float ham_sign(int k, int i)
{
    // component k of a (x) b = sum_i sign[k][i] * a_i * b_(i ^ k)   (model_util._hamilton)
    const unsigned neg = 0x428eu;            // bit (4k + i) set: the term is negative  [k=0: i=1,2,3 | k=1: i=3 | k=2: i=1 | k=3: i=2]
    return (neg >> (4 * k + i)) & 1u ? -1.0f : 1.0f;
}
__device__ __forceinline__ Q4 ham(const Q4 &a, const Q4 &b)
{
    Q4 c;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float s = 0.0f;
#pragma unroll
        for (int i = 0; i < 4; ++i) s += ham_sign(k, i) * a.v[i] * b.v[i ^ k];
        c.v[k] = s;
    }
    return c;
}
// adjoint of c = a (x) b: ga += dL/da, gb += dL/db for the incoming gc
__device__ __forceinline__ void ham_bwd(const Q4 &a, const Q4 &b, const Q4 &gc, Q4 &ga, Q4 &gb)
{
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float w = ham_sign(k, i) * gc.v[k];
            ga.v[i] += w * b.v[i ^ k];
            gb.v[i ^ k] += w * a.v[i];
        }
}
// y = q / (sqrt(sum q^2 + 1e-10) + 1e-10)   (pwclo_model.py:206)
__device__ __forceinline__ Q4 normalise(const Q4 &q)
{
    const float n = sqrtf(q.v[0] * q.v[0] + q.v[1] * q.v[1] + q.v[2] * q.v[2] + q.v[3] * q.v[3] + 1e-10f), d = n + 1e-10f;
    return Q4{{q.v[0] / d, q.v[1] / d, q.v[2] / d, q.v[3] / d}};
}
__device__ __forceinline__ void normalise_bwd(const Q4 &q, const Q4 &gy, Q4 &gq)
{
    const float n = sqrtf(q.v[0] * q.v[0] + q.v[1] * q.v[1] + q.v[2] * q.v[2] + q.v[3] * q.v[3] + 1e-10f), d = n + 1e-10f;
    const float dot = gy.v[0] * q.v[0] + gy.v[1] * q.v[1] + gy.v[2] * q.v[2] + gy.v[3] * q.v[3];
#pragma unroll
    for (int i = 0; i < 4; ++i) gq.v[i] += gy.v[i] / d - dot / (d * d) * (q.v[i] / n);
}
// y = conj(q) / (sum q^2 + 1e-10)   (model_util.py:61-69)
__device__ __forceinline__ Q4 inverse(const Q4 &q)
{
    const float s = q.v[0] * q.v[0] + q.v[1] * q.v[1] + q.v[2] * q.v[2] + q.v[3] * q.v[3] + 1e-10f;
    return Q4{{q.v[0] / s, -q.v[1] / s, -q.v[2] / s, -q.v[3] / s}};
}
__device__ __forceinline__ void inverse_bwd(const Q4 &q, const Q4 &gy, Q4 &gq)
{
    const float s = q.v[0] * q.v[0] + q.v[1] * q.v[1] + q.v[2] * q.v[2] + q.v[3] * q.v[3] + 1e-10f;
    const float c[4] = {q.v[0], -q.v[1], -q.v[2], -q.v[3]};
    const float dot = gy.v[0] * c[0] + gy.v[1] * c[1] + gy.v[2] * c[2] + gy.v[3] * c[3];
    const float gs = -dot / (s * s);
#pragma unroll
    for (int i = 0; i < 4; ++i) gq.v[i] += (i == 0 ? 1.0f : -1.0f) * gy.v[i] / s + gs * 2.0f * q.v[i];
}

// forward of one batch element; `coarse`: the l3 head (no composition: q = normalise(q_raw), t = t_det)
struct PoseFwd { Q4 q_det, qi, p1, q, q_norm; float t[3]; };
__device__ __forceinline__ PoseFwd pose_forward(const Q4 &qr, const float (&td)[3], const Q4 &qc, const float (&tc)[3], bool coarse)
{
    PoseFwd f;
    f.q_det = normalise(qr);
    if (coarse) {
        f.q = f.q_det;
        f.t[0] = td[0]; f.t[1] = td[1]; f.t[2] = td[2];
    } else {
        f.qi = inverse(f.q_det);
        const Q4 P{{0.0f, tc[0], tc[1], tc[2]}};
        f.p1 = ham(f.q_det, P);                            // :275-277: t_coarse rotated by q_det
        const Q4 p2 = ham(f.p1, f.qi);
        f.q = ham(f.q_det, qc);                            // :279
        f.t[0] = p2.v[1] + td[0]; f.t[1] = p2.v[2] + td[1]; f.t[2] = p2.v[3] + td[2];     // :280
    }
    f.q_norm = normalise(f.q);
    return f;
}
}  // namespace

__global__ void pose_compose_kernel(const elo_pose_compose_args a)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= a.batch) return;
    const bool coarse = a.q_coarse == nullptr;
    Q4 qr, qc{{1.0f, 0.0f, 0.0f, 0.0f}};
    float td[3], tc[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int i = 0; i < 4; ++i) qr.v[i] = a.q_raw[b * 4 + i];
#pragma unroll
    for (int i = 0; i < 3; ++i) td[i] = a.t_det[b * 3 + i];
    if (!coarse) {
#pragma unroll
        for (int i = 0; i < 4; ++i) qc.v[i] = a.q_coarse[b * 4 + i];
#pragma unroll
        for (int i = 0; i < 3; ++i) tc[i] = a.t_coarse[b * 3 + i];
    }
    const PoseFwd f = pose_forward(qr, td, qc, tc, coarse);
    if (!a.grad_q) {                                           // ---- forward
#pragma unroll
        for (int i = 0; i < 4; ++i) { a.q[b * 4 + i] = f.q.v[i]; a.q_norm[b * 4 + i] = f.q_norm.v[i]; }
#pragma unroll
        for (int i = 0; i < 3; ++i) a.t[b * 3 + i] = f.t[i];
        return;
    }
    // ---- backward: incoming (grad_q, grad_t, grad_q_norm) -> (grad_q_raw, grad_t_det, grad_q_coarse, grad_t_coarse)
    Q4 gq, gqn, gq_det{{0, 0, 0, 0}}, gqc{{0, 0, 0, 0}}, gqr{{0, 0, 0, 0}};
    float gt[3];
#pragma unroll
    for (int i = 0; i < 4; ++i) { gq.v[i] = a.grad_q[b * 4 + i]; gqn.v[i] = a.grad_q_norm[b * 4 + i]; }
#pragma unroll
    for (int i = 0; i < 3; ++i) gt[i] = a.grad_t[b * 3 + i];
    normalise_bwd(f.q, gqn, gq);                               // q_norm = normalise(q)
    if (coarse) {
        gq_det = gq;                                           // q = q_det
    } else {
        Q4 gP{{0, 0, 0, 0}}, gp1{{0, 0, 0, 0}}, gqi{{0, 0, 0, 0}};
        ham_bwd(f.q_det, qc, gq, gq_det, gqc);                 // q = q_det (x) q_coarse
        const Q4 gp2{{0.0f, gt[0], gt[1], gt[2]}};             // t = p2[1:] + t_det
        ham_bwd(f.p1, f.qi, gp2, gp1, gqi);                    // p2 = p1 (x) qi
        const Q4 P{{0.0f, tc[0], tc[1], tc[2]}};
        ham_bwd(f.q_det, P, gp1, gq_det, gP);                  // p1 = q_det (x) P
        inverse_bwd(f.q_det, gqi, gq_det);                     // qi = inverse(q_det)
#pragma unroll
        for (int i = 0; i < 4; ++i) a.grad_q_coarse[b * 4 + i] = gqc.v[i];
#pragma unroll
        for (int i = 0; i < 3; ++i) a.grad_t_coarse[b * 3 + i] = gP.v[1 + i];
    }
    normalise_bwd(qr, gq_det, gqr);                            // q_det = normalise(q_raw)
#pragma unroll
    for (int i = 0; i < 4; ++i) a.grad_q_raw[b * 4 + i] = gqr.v[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) a.grad_t_det[b * 3 + i] = gt[i];
}

extern "C" int elo_pose_compose(const elo_pose_compose_args *a, elo_stream_t stream)
{
    const char *who = "elo_pose_compose";
    if (!a) return fail(ELO_ERR_ARG, "%s: null argument block", who);
    if (a->batch <= 0) return a->batch == 0 ? ELO_OK : fail(ELO_ERR_ARG, "%s: bad batch", who);
    if (!a->q_raw || !a->t_det || (!a->q_coarse != !a->t_coarse)) return fail(ELO_ERR_ARG, "%s: null tensor pointer", who);
    if (a->grad_q) {
        if (!a->grad_t || !a->grad_q_norm || !a->grad_q_raw || !a->grad_t_det || (a->q_coarse && (!a->grad_q_coarse || !a->grad_t_coarse)))
            return fail(ELO_ERR_ARG, "%s: backward needs every gradient pointer", who);
    } else if (!a->q || !a->t || !a->q_norm) return fail(ELO_ERR_ARG, "%s: null output pointer", who);
    hipLaunchKernelGGL(pose_compose_kernel, dim3((unsigned)((a->batch + 63) / 64)), dim3(64), 0, (hipStream_t)stream, *a);
    return check_launch(who);
}

// get_loss (pwclo_model.py:437-481) over the four levels in one workgroup; backward (grad_out given): the same kernel
// writes the gradients of the eight pose tensors and of (w_x, w_q), scaled by the incoming scalar gradient.
__global__ __launch_bounds__(64) void pose_loss_kernel(const elo_pose_loss_args a)
{
    const int b = threadIdx.x, B = a.batch;
    const float wx = *a.w_x, wq = *a.w_q, ex = expf(-wx), eq = expf(-wq);
    const float weight[4] = {0.2f, 0.4f, 0.8f, 1.6f};         // l0 .. l3  (:478-481)
    const float go = a.grad_out ? *a.grad_out : 0.0f;
    float loss = 0.0f, gwx = 0.0f, gwq = 0.0f;                 // (per thread: its batch elements' share)
    for (int lv = 0; lv < 4; ++lv) {
        float lq = 0.0f, lx = 0.0f;
        for (int e = b; e < B; e += 64) {
            Q4 q, gt4;
#pragma unroll
            for (int i = 0; i < 4; ++i) { q.v[i] = a.q[lv][e * 4 + i]; gt4.v[i] = a.q_gt[e * 4 + i]; }
            const Q4 qn = normalise(q);                        // :443 (the model's output is normalised once more)
            float d[4], s = 1e-10f;
#pragma unroll
            for (int i = 0; i < 4; ++i) { d[i] = gt4.v[i] - qn.v[i]; s += d[i] * d[i]; }
            const float nq = sqrtf(s);
            lq += nq;
            float dx[3], rx[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) { dx[i] = a.t[lv][e * 3 + i] - a.t_gt[e * 3 + i]; rx[i] = sqrtf(dx[i] * dx[i] + 1e-10f); lx += rx[i]; }
            if (a.grad_out) {
                // d level / d loss_q = exp(-w_q) / B; loss_q element = sqrt(sum d^2 + 1e-10), d = q_gt - normalise(q)
                const float cq = go * weight[lv] * eq / B;
                Q4 gqn, gq{{0, 0, 0, 0}};
#pragma unroll
                for (int i = 0; i < 4; ++i) gqn.v[i] = -cq * d[i] / nq;
                normalise_bwd(q, gqn, gq);
#pragma unroll
                for (int i = 0; i < 4; ++i) a.grad_q[lv][e * 4 + i] = gq.v[i];
                const float cx = go * weight[lv] * ex / (3.0f * B);
#pragma unroll
                for (int i = 0; i < 3; ++i) a.grad_t[lv][e * 3 + i] = cx * dx[i] / rx[i];
            }
        }
        // sums over the batch: a 64-lane butterfly (one wave)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { lq += __shfl_xor(lq, o); lx += __shfl_xor(lx, o); }
        const float loss_q = lq / B, loss_x = lx / (3.0f * B);
        loss += weight[lv] * (loss_x * ex + wx + loss_q * eq + wq);
        gwx += weight[lv] * (1.0f - loss_x * ex);
        gwq += weight[lv] * (1.0f - loss_q * eq);
    }
    if (b == 0) {
        if (a.grad_out) { *a.grad_w_x = go * gwx; *a.grad_w_q = go * gwq; }
        else *a.loss = loss;
    }
}

extern "C" int elo_pose_loss(const elo_pose_loss_args *a, elo_stream_t stream)
{
    const char *who = "elo_pose_loss";
    if (!a) return fail(ELO_ERR_ARG, "%s: null argument block", who);
    if (a->batch <= 0) return fail(ELO_ERR_ARG, "%s: bad batch", who);
    for (int lv = 0; lv < 4; ++lv)
        if (!a->q[lv] || !a->t[lv] || (a->grad_out && (!a->grad_q[lv] || !a->grad_t[lv]))) return fail(ELO_ERR_ARG, "%s: null tensor pointer", who);
    if (!a->q_gt || !a->t_gt || !a->w_x || !a->w_q || (a->grad_out ? (!a->grad_w_x || !a->grad_w_q) : !a->loss))
        return fail(ELO_ERR_ARG, "%s: null tensor pointer", who);
    hipLaunchKernelGGL(pose_loss_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, *a);
    return check_launch(who);
}
