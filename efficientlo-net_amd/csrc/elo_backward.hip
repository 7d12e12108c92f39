// elo_backward.hip -- backward passes of the per-operator feature kernels (elo_features.hip) for TRAINING.
//
// The reference trains through `train_op` (main.py:171-176): TensorFlow differentiates its stock ops, i.e.
//   tf.gather_nd(x, idx) * mask   ->  scatter-add of the incoming gradient into x at idx, masked slots dropped
//                                     (utils/pointnet_util.py:54-55, :110-111, :203-204, :277-278; the mask is wrapped in
//                                     stop_gradient and the indices are integers: no gradient for either)
//   tf.reduce_max                 ->  the gradient goes to the maximal entries (split evenly among exact ties)
//   tf.nn.softmax + reduce_sum    ->  the softmax Jacobian
//   tf.scatter_nd (projection)    ->  a gather of the gradient at the scattered rows (model_util.py:264-273); the cell
//                                     indices and the minimum-range mask carry no gradient
// Each entry point below is the hand-written adjoint of ONE forward kernel: every gradient of an operator comes from
// one launch instead of the ~20 elementwise / index_add launches autograd would chain.  fp32 only (training stores fp32).
// Outputs marked "accumulated" receive atomic adds and must be ZERO on entry (the caller allocates them with
// torch.zeros: a fill kernel, capturable in a hipGraph); the others are written in full.  A MASKED slot adds nothing
// anywhere: its nominal target is cell (0,0,0) of batch element 0 and as atomics tens of
// thousands of them would queue on one row (same-address atomics serialise at ~170 ns each).
#include "elo_common.h"
#include "elo_project_device.h"

namespace elo {
namespace {

__device__ __forceinline__ long cell_of(const int *idx, long row, int H, int W)
{
    const int *id = idx + row * 3;
    return ((long)id[0] * H + id[1]) * W + id[2];
}

// ---------------------------------------------------------------- group_concat: out = [src_xyz[idx]*m - centre, src_feat[idx]*m]
__global__ __launch_bounds__(ELO_BLOCK) void group_concat_bwd_kernel(const elo_group_concat_bwd_args a)
{
    const int CT = 3 + a.C;
    const long points = (long)a.batch * a.npoints, rows = points * a.K;
    const long stride = (long)gridDim.x * blockDim.x, me = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (a.grad_centre) {                                           // d/d centre = -1 on every slot, masked or not
        for (long e = me; e < points * 3; e += stride) {
            const long pt = e / 3;
            const int c = (int)(e - pt * 3);
            float s = 0.0f;
            for (int k = 0; k < a.K; ++k) s += a.grad_out[(pt * a.K + k) * CT + c];
            a.grad_centre[e] = -s;
        }
    }
    for (long e = me; e < rows * CT; e += stride) {
        const long row = point_batch(e, CT);
        const int c = (int)(e - row * CT);
        if (a.mask[row] == 0.0f) continue;
        const float g = a.grad_out[e];
        if (g == 0.0f) continue;
        const long cell = cell_of(a.idx, row, a.H2, a.W2);
        if (c < 3) { if (a.grad_src_xyz) atomicAdd(a.grad_src_xyz + cell * 3 + c, g); }
        else if (a.grad_src_feat) atomicAdd(a.grad_src_feat + cell * a.C + (c - 3), g);
    }
}

// ---------------------------------------------------------------- masked max pool: out = max_k x*m
__global__ __launch_bounds__(ELO_BLOCK) void masked_maxpool_bwd_kernel(const elo_masked_maxpool_bwd_args a)
{
    const long total = (long)a.batch * a.npoints * a.C;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const long pt = point_batch(e, a.C);
        const int c = (int)(e - pt * a.C);
        const float *x = a.x + pt * a.K * a.C + c;
        const float *m = a.mask + pt * a.K;
        float best = -INFINITY;
        for (int k = 0; k < a.K; ++k) best = fmaxf(best, x[(long)k * a.C] * m[k]);
        int ties = 0;
        for (int k = 0; k < a.K; ++k) ties += x[(long)k * a.C] * m[k] == best;
        const float g = a.grad_out[e] / (float)ties;                 // reduce_max: even split among exact ties
        float *o = a.grad_x + pt * a.K * a.C + c;
        for (int k = 0; k < a.K; ++k) o[(long)k * a.C] = x[(long)k * a.C] * m[k] == best ? g * m[k] : 0.0f;
    }
}

// The same, a thread per (point, four consecutive channels) with the K products held in registers: ONE pass over x with 16-byte loads
// (all K in flight) and 16-byte stores, instead of three strided 4-byte walks and a 4-byte store per element.  Same comparisons, same bits.
template <int K>
__global__ __launch_bounds__(ELO_BLOCK) void masked_maxpool_bwd_vec_kernel(const elo_masked_maxpool_bwd_args a, const long items)
{
    const long e = (long)blockIdx.x * ELO_BLOCK + threadIdx.x;
    if (e >= items) return;
    const int q = a.C >> 2;
    const long pt = point_batch(e, q);
    const int cq = (int)(e - pt * q);
    const float4 *x = reinterpret_cast<const float4 *>(a.x + pt * K * a.C) + cq;
    const float *m = a.mask + pt * K;
    float4 p[K];
    float w[K];
#pragma unroll
    for (int k = 0; k < K; ++k) { p[k] = x[(long)k * q]; w[k] = m[k]; }
    float4 best{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
    for (int k = 0; k < K; ++k) {
        p[k] = float4{p[k].x * w[k], p[k].y * w[k], p[k].z * w[k], p[k].w * w[k]};
        best = float4{fmaxf(best.x, p[k].x), fmaxf(best.y, p[k].y), fmaxf(best.z, p[k].z), fmaxf(best.w, p[k].w)};
    }
    int tx = 0, ty = 0, tz = 0, tw = 0;
#pragma unroll
    for (int k = 0; k < K; ++k) { tx += p[k].x == best.x; ty += p[k].y == best.y; tz += p[k].z == best.z; tw += p[k].w == best.w; }
    const float4 go = reinterpret_cast<const float4 *>(a.grad_out + pt * a.C)[cq];
    const float4 g{go.x / (float)tx, go.y / (float)ty, go.z / (float)tz, go.w / (float)tw};      // reduce_max: even split among exact ties
    float4 *o = reinterpret_cast<float4 *>(a.grad_x + pt * K * a.C) + cq;
#pragma unroll
    for (int k = 0; k < K; ++k)
        o[(long)k * q] = float4{p[k].x == best.x ? g.x * w[k] : 0.0f, p[k].y == best.y ? g.y * w[k] : 0.0f,
                                p[k].z == best.z ? g.z * w[k] : 0.0f, p[k].w == best.w ? g.w * w[k] : 0.0f};
}

// ---------------------------------------------------------------- geometry code [p, g, g - p, |g - p|] (g already masked)
struct GeoGrad { float p[3], g[3]; };

__device__ __forceinline__ GeoGrad geometry_bwd(const float *p, const float *g, const float *go)
{   // go: the 10 incoming gradients
    float d[3];
    for (int i = 0; i < 3; ++i) d[i] = g[i] - p[i];
    const float euc = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + 1e-20f);
    GeoGrad r;
    for (int i = 0; i < 3; ++i) {
        const float via = go[6 + i] + go[9] * d[i] / euc;            // through diff and through the norm
        r.p[i] = go[i] - via;
        r.g[i] = go[3 + i] + via;
    }
    return r;
}

// cost volume stage 1: out = [geometry(p = xyz1, g = xyz2[idx]*m), feat1 (tiled), feat2[idx]*m]
__global__ __launch_bounds__(ELO_BLOCK) void cv_encode1_bwd_kernel(const elo_cv_encode1_bwd_args a)
{
    const int CT = 10 + 2 * a.C;
    const long points = (long)a.batch * a.npoints;
    const long stride = (long)gridDim.x * blockDim.x, me = (long)blockIdx.x * blockDim.x + threadIdx.x;
    for (long pt = me; pt < points; pt += stride) {                  // geometry: one thread per point walks its K rows
        const float p[3] = {a.xyz1[pt * 3], a.xyz1[pt * 3 + 1], a.xyz1[pt * 3 + 2]};
        float gp[3] = {0.0f, 0.0f, 0.0f};
        for (int k = 0; k < a.K; ++k) {
            const long row = pt * a.K + k;
            const float m = a.mask[row];
            const long cell = cell_of(a.idx, row, a.H2, a.W2);
            const float g[3] = {a.xyz2[cell * 3] * m, a.xyz2[cell * 3 + 1] * m, a.xyz2[cell * 3 + 2] * m};
            const GeoGrad r = geometry_bwd(p, g, a.grad_out + row * CT);
            for (int i = 0; i < 3; ++i) {
                gp[i] += r.p[i];
                if (a.grad_xyz2 && m != 0.0f && r.g[i] != 0.0f) atomicAdd(a.grad_xyz2 + cell * 3 + i, r.g[i] * m);
            }
        }
        if (a.grad_xyz1) for (int i = 0; i < 3; ++i) a.grad_xyz1[pt * 3 + i] = gp[i];
    }
    if (a.grad_feat1) {
        for (long e = me; e < points * a.C; e += stride) {           // feat1 is tiled over K: its gradient is the sum
            const long pt = point_batch(e, a.C);
            const int c = (int)(e - pt * a.C);
            float s = 0.0f;
            for (int k = 0; k < a.K; ++k) s += a.grad_out[(pt * a.K + k) * CT + 10 + c];
            a.grad_feat1[e] = s;
        }
    }
    if (a.grad_feat2) {
        for (long e = me; e < points * a.K * a.C; e += stride) {
            const long row = point_batch(e, a.C);
            const int c = (int)(e - row * a.C);
            if (a.mask[row] == 0.0f) continue;
            const float g = a.grad_out[row * CT + 10 + a.C + c];
            if (g != 0.0f) atomicAdd(a.grad_feat2 + cell_of(a.idx, row, a.H2, a.W2) * a.C + c, g);
        }
    }
}

// cost volume stage 2: xyz_cat = geometry(p = xyz1[b,n], g = xyz1[idx]*m), rest = [feat1 (tiled), cost[idx]*m]
__global__ __launch_bounds__(ELO_BLOCK) void cv_encode2_bwd_kernel(const elo_cv_encode2_bwd_args a)
{
    const int CR = a.C + a.Cc;
    const long points = (long)a.batch * a.npoints;
    const long stride = (long)gridDim.x * blockDim.x, me = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (a.grad_xyz1) {
        for (long pt = me; pt < points; pt += stride) {              // centre and neighbours live in the same grid: all atomics
            const float p[3] = {a.xyz1[pt * 3], a.xyz1[pt * 3 + 1], a.xyz1[pt * 3 + 2]};
            float gp[3] = {0.0f, 0.0f, 0.0f};
            for (int k = 0; k < a.K; ++k) {
                const long row = pt * a.K + k;
                const float m = a.mask[row];
                const long cell = cell_of(a.idx, row, a.H, a.W);
                const float g[3] = {a.xyz1[cell * 3] * m, a.xyz1[cell * 3 + 1] * m, a.xyz1[cell * 3 + 2] * m};
                const GeoGrad r = geometry_bwd(p, g, a.grad_xyz_cat + row * 10);
                for (int i = 0; i < 3; ++i) {
                    gp[i] += r.p[i];
                    if (m != 0.0f && r.g[i] != 0.0f) atomicAdd(a.grad_xyz1 + cell * 3 + i, r.g[i] * m);
                }
            }
            for (int i = 0; i < 3; ++i) atomicAdd(a.grad_xyz1 + pt * 3 + i, gp[i]);
        }
    }
    if (a.grad_feat1) {
        for (long e = me; e < points * a.C; e += stride) {
            const long pt = point_batch(e, a.C);
            const int c = (int)(e - pt * a.C);
            float s = 0.0f;
            for (int k = 0; k < a.K; ++k) s += a.grad_rest[(pt * a.K + k) * CR + c];
            a.grad_feat1[e] = s;
        }
    }
    if (a.grad_cost) {
        for (long e = me; e < points * a.K * a.Cc; e += stride) {
            const long row = point_batch(e, a.Cc);
            const int c = (int)(e - row * a.Cc);
            if (a.mask[row] == 0.0f) continue;
            const float g = a.grad_rest[row * CR + a.C + c];
            if (g != 0.0f) atomicAdd(a.grad_cost + cell_of(a.idx, row, a.H, a.W) * a.Cc + c, g);
        }
    }
}

// ---------------------------------------------------------------- masked softmax pool: out = sum_k softmax_k(l) v
__global__ __launch_bounds__(ELO_BLOCK) void softmax_pool_bwd_kernel(const elo_softmax_pool_bwd_args a)
{
    const long total = (long)a.batch * a.npoints * a.C;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const long pt = point_batch(e, a.C);
        const int c = (int)(e - pt * a.C);
        const float *l = a.logits + pt * a.K * a.C + c;
        const float *v = a.values + pt * a.K * (long)a.values_stride + c;
        const float *m = a.mask + pt * a.K;
        float mx = -INFINITY;
        for (int k = 0; k < a.K; ++k) mx = fmaxf(mx, m[k] == 1.0f ? l[(long)k * a.C] : -1e10f);
        float den = 0.0f, acc = 0.0f;
        for (int k = 0; k < a.K; ++k) {
            const float ex = expf((m[k] == 1.0f ? l[(long)k * a.C] : -1e10f) - mx);
            den += ex;
            acc += ex * v[(long)k * a.values_stride];
        }
        const float out = acc / den, g = a.grad_out[e];
        for (int k = 0; k < a.K; ++k) {
            const float s = expf((m[k] == 1.0f ? l[(long)k * a.C] : -1e10f) - mx) / den;
            a.grad_values[(pt * a.K + k) * a.C + c] = s * g;
            // a masked logit is the constant -1e10 (tf.where): no gradient reaches the tensor there
            a.grad_logits[(pt * a.K + k) * a.C + c] = m[k] == 1.0f ? s * (v[(long)k * a.values_stride] - out) * g : 0.0f;
        }
    }
}

// The same, a thread per (point, four consecutive channels), the K logits and values held in registers (K = 4, 6, 8 -- the cost volume's
// neighbour counts at the refinement levels): one pass with 16-byte loads and stores, each exponential computed once.  Same operations
// per element in the same order (max, exponentials against it, sums in k order): same bits.
template <int K>
__global__ __launch_bounds__(ELO_BLOCK) void softmax_pool_bwd_vec_kernel(const elo_softmax_pool_bwd_args a, const long items)
{
    const long e = (long)blockIdx.x * ELO_BLOCK + threadIdx.x;
    if (e >= items) return;
    const int q = a.C >> 2, vq = a.values_stride >> 2;
    const long pt = point_batch(e, q);
    const int cq = (int)(e - pt * q);
    const float4 *lp = reinterpret_cast<const float4 *>(a.logits + pt * K * a.C) + cq;
    const float4 *vp = reinterpret_cast<const float4 *>(a.values + pt * K * (long)a.values_stride) + cq;
    const float *m = a.mask + pt * K;
    float4 l[K], v[K];
    bool on[K];
#pragma unroll
    for (int k = 0; k < K; ++k) { l[k] = lp[(long)k * q]; v[k] = vp[(long)k * vq]; on[k] = m[k] == 1.0f; }
    float mx[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY}, den[4] = {0.f, 0.f, 0.f, 0.f}, acc[4] = {0.f, 0.f, 0.f, 0.f};
    float ex[K][4];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const float lk[4] = {l[k].x, l[k].y, l[k].z, l[k].w};
#pragma unroll
        for (int i = 0; i < 4; ++i) mx[i] = fmaxf(mx[i], on[k] ? lk[i] : -1e10f);
    }
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const float lk[4] = {l[k].x, l[k].y, l[k].z, l[k].w}, vk[4] = {v[k].x, v[k].y, v[k].z, v[k].w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            ex[k][i] = expf((on[k] ? lk[i] : -1e10f) - mx[i]);
            den[i] += ex[k][i];
            acc[i] += ex[k][i] * vk[i];
        }
    }
    const float4 go = reinterpret_cast<const float4 *>(a.grad_out + pt * a.C)[cq];
    const float g[4] = {go.x, go.y, go.z, go.w};
    float out[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) out[i] = acc[i] / den[i];
    float4 *gv = reinterpret_cast<float4 *>(a.grad_values + pt * K * a.C) + cq, *gl = reinterpret_cast<float4 *>(a.grad_logits + pt * K * a.C) + cq;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const float vk[4] = {v[k].x, v[k].y, v[k].z, v[k].w};
        float sv[4], sl[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float s = ex[k][i] / den[i];
            sv[i] = s * g[i];
            sl[i] = on[k] ? s * (vk[i] - out[i]) * g[i] : 0.0f;      // a masked logit is the constant -1e10 (tf.where): no gradient there
        }
        gv[(long)k * q] = float4{sv[0], sv[1], sv[2], sv[3]};
        gl[(long)k * q] = float4{sl[0], sl[1], sl[2], sl[3]};
    }
}

// ... and for K = 16, 32 (the l2 cost volume's 32 candidates): the logits stay in registers and become the exponentials in place; the values
// are streamed twice (the second time from L2).  Same operations per element, same order.
template <int K>
__global__ __launch_bounds__(ELO_BLOCK) void softmax_pool_bwd_vec2_kernel(const elo_softmax_pool_bwd_args a, const long items)
{
    const long e = (long)blockIdx.x * ELO_BLOCK + threadIdx.x;
    if (e >= items) return;
    const int q = a.C >> 2, vq = a.values_stride >> 2;
    const long pt = point_batch(e, q);
    const int cq = (int)(e - pt * q);
    const float4 *lp = reinterpret_cast<const float4 *>(a.logits + pt * K * a.C) + cq;
    const float4 *vp = reinterpret_cast<const float4 *>(a.values + pt * K * (long)a.values_stride) + cq;
    const float *m = a.mask + pt * K;
    float4 l[K];
    unsigned on = 0;
#pragma unroll
    for (int k = 0; k < K; ++k) { l[k] = lp[(long)k * q]; on |= (m[k] == 1.0f ? 1u : 0u) << k; }
    float mx[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY}, den[4] = {0.f, 0.f, 0.f, 0.f}, acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const bool o = (on >> k) & 1u;
        mx[0] = fmaxf(mx[0], o ? l[k].x : -1e10f); mx[1] = fmaxf(mx[1], o ? l[k].y : -1e10f);
        mx[2] = fmaxf(mx[2], o ? l[k].z : -1e10f); mx[3] = fmaxf(mx[3], o ? l[k].w : -1e10f);
    }
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const bool o = (on >> k) & 1u;
        const float4 v = vp[(long)k * vq];
        l[k] = float4{expf((o ? l[k].x : -1e10f) - mx[0]), expf((o ? l[k].y : -1e10f) - mx[1]), expf((o ? l[k].z : -1e10f) - mx[2]),
                      expf((o ? l[k].w : -1e10f) - mx[3])};
        den[0] += l[k].x; den[1] += l[k].y; den[2] += l[k].z; den[3] += l[k].w;
        acc[0] += l[k].x * v.x; acc[1] += l[k].y * v.y; acc[2] += l[k].z * v.z; acc[3] += l[k].w * v.w;
    }
    const float4 g = reinterpret_cast<const float4 *>(a.grad_out + pt * a.C)[cq];
    const float out[4] = {acc[0] / den[0], acc[1] / den[1], acc[2] / den[2], acc[3] / den[3]};
    float4 *gv = reinterpret_cast<float4 *>(a.grad_values + pt * K * a.C) + cq, *gl = reinterpret_cast<float4 *>(a.grad_logits + pt * K * a.C) + cq;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const bool o = (on >> k) & 1u;
        const float4 v = vp[(long)k * vq];
        const float s0 = l[k].x / den[0], s1 = l[k].y / den[1], s2 = l[k].z / den[2], s3 = l[k].w / den[3];
        gv[(long)k * q] = float4{s0 * g.x, s1 * g.y, s2 * g.z, s3 * g.w};
        gl[(long)k * q] = float4{o ? s0 * (v.x - out[0]) * g.x : 0.0f, o ? s1 * (v.y - out[1]) * g.y : 0.0f,
                                 o ? s2 * (v.z - out[2]) * g.z : 0.0f, o ? s3 * (v.w - out[3]) * g.w : 0.0f};
    }
}

// ---------------------------------------------------------------- softmax_valid: out[b,c] = sum_n softmax_n(w | valid) f
// block (64 channels, 4 waves over the points) per (channel group, batch element): pass 1 max, pass 2 sums, pass 3 gradients
__global__ __launch_bounds__(ELO_BLOCK) void softmax_valid_bwd_kernel(const elo_softmax_valid_bwd_args a)
{
    __shared__ float red[3][ELO_BLOCK / ELO_WAVE][ELO_WAVE];
    const int b = blockIdx.y, lane = threadIdx.x % ELO_WAVE, wave = threadIdx.x / ELO_WAVE, waves = ELO_BLOCK / ELO_WAVE;
    const int c = blockIdx.x * ELO_WAVE + lane;
    const bool live = c < a.C;
    const int cc = live ? c : a.C - 1;
    const float *f = a.feature + (long)b * a.npoints * a.C, *w = a.weight + (long)b * a.npoints * a.C;
    const float *p = a.xyz + (long)b * a.npoints * 3;
    auto valid = [&](int n) { return !((p[n * 3] == 0.0f) & (p[n * 3 + 1] == 0.0f) & (p[n * 3 + 2] == 0.0f)); };
    float mx = -INFINITY;
    for (int n = wave; n < a.npoints; n += waves) if (valid(n)) mx = fmaxf(mx, w[(long)n * a.C + cc]);
    red[0][wave][lane] = mx;
    __syncthreads();
    for (int i = 0; i < waves; ++i) mx = fmaxf(mx, red[0][i][lane]);
    float den = 0.0f, acc = 0.0f;
    for (int n = wave; n < a.npoints; n += waves) {
        if (!valid(n)) continue;
        const float ex = expf(w[(long)n * a.C + cc] - mx);
        den += ex;
        acc += ex * f[(long)n * a.C + cc];
    }
    red[1][wave][lane] = den; red[2][wave][lane] = acc;
    __syncthreads();
    den = acc = 0.0f;
    for (int i = 0; i < waves; ++i) { den += red[1][i][lane]; acc += red[2][i][lane]; }
    const float out = den > 0.0f ? acc / den : 0.0f, g = a.grad_out[(long)b * a.C + cc];
    if (!live) return;
    for (int n = wave; n < a.npoints; n += waves) {
        const long at = ((long)b * a.npoints + n) * a.C + c;
        const bool ok = valid(n) && den > 0.0f;
        const float s = ok ? expf(w[(long)n * a.C + c] - mx) / den : 0.0f;
        a.grad_feature[at] = s * g;
        a.grad_weight[at] = s * (f[(long)n * a.C + c] - out) * g;
    }
}

// with the forward's (maximum, denominator, out) per channel the adjoint is element-wise: thread per (point, 4 channels)
__global__ __launch_bounds__(ELO_BLOCK) void softmax_valid_bwd_elementwise_kernel(const elo_softmax_valid_bwd_args a)
{
    const int q = a.C >> 2;
    const long per_b = (long)a.npoints * q, total = per_b * a.batch;
    for (long i = (long)blockIdx.x * ELO_BLOCK + threadIdx.x; i < total; i += (long)gridDim.x * ELO_BLOCK) {
        const int b = per_b < 0x7fffffffL ? point_batch(i, (int)per_b) : (int)(i / per_b);     // (32-bit divisions: elo_common.h)
        const long r = i - (long)b * per_b;
        const int n = point_batch(r, q), cg = (int)(r - (long)n * q);
        const float *p = a.xyz + ((long)b * a.npoints + n) * 3;
        const float px = p[0], py = p[1], pz = p[2];
        const float4 w = reinterpret_cast<const float4 *>(a.weight)[i], f = reinterpret_cast<const float4 *>(a.feature)[i];
        const float4 M = reinterpret_cast<const float4 *>(a.stats + (size_t)b * 2 * a.C)[cg];
        const float4 D = reinterpret_cast<const float4 *>(a.stats + ((size_t)b * 2 + 1) * a.C)[cg];
        const float4 o = reinterpret_cast<const float4 *>(a.out + (size_t)b * a.C)[cg], g = reinterpret_cast<const float4 *>(a.grad_out + (size_t)b * a.C)[cg];
        const bool ok = !((px == 0.0f) & (py == 0.0f) & (pz == 0.0f));
        auto soft = [&](float wv, float m, float d) { return ok && d > 0.0f ? expf(wv - m) / d : 0.0f; };
        const float4 s{soft(w.x, M.x, D.x), soft(w.y, M.y, D.y), soft(w.z, M.z, D.z), soft(w.w, M.w, D.w)};
        reinterpret_cast<float4 *>(a.grad_feature)[i] = float4{s.x * g.x, s.y * g.y, s.z * g.z, s.w * g.w};
        reinterpret_cast<float4 *>(a.grad_weight)[i] = float4{s.x * (f.x - o.x) * g.x, s.y * (f.y - o.y) * g.y, s.z * (f.z - o.z) * g.z,
                                                              s.w * (f.w - o.w) * g.w};
    }
}

// ---------------------------------------------------------------- warp + spherical re-projection
// forward (elo_warp_project): pts = (R(q) x + t) * keep; out_xyz[cell] += pts, out_feat[cell] += feat for the point(s)
// holding the cell's minimum range.  backward: a winner receives its cell's gradient; then through the warp:
//   grad_t = sum G,   grad_q = sum (Ghat q X* + Ghat* q X) / n2  -  2 q sum (G . (q X q*)_vec) / n2^2,   G = keep * grad_pts,
// X = [0, x], Ghat = [0, G], n2 = |q|^2 + 1e-10 (the reference's inv_q, model_util.py:61-69).
constexpr int ZFLAGS = 4;

__device__ __forceinline__ void hamilton(const float *a, const float *b, float *r)
{
    r[0] = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
    r[1] = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
    r[2] = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
    r[3] = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
}

__global__ __launch_bounds__(ELO_BLOCK) void warp_project_bwd_kernel(const elo_warp_project_bwd_args a)
{
    __shared__ float part[7][ELO_BLOCK / ELO_WAVE];
    const int b = blockIdx.y;
    const size_t cells = (size_t)a.batch * a.H * a.W, pts = (size_t)a.batch * a.npoints;
    const unsigned *minr = a.scratch, *zflag = a.scratch + cells;
    const int *cell_of_pt = (const int *)(a.scratch + cells + ZFLAGS * (size_t)a.batch);
    const unsigned *rbits = a.scratch + cells + ZFLAGS * (size_t)a.batch + pts;
    float acc[7] = {0, 0, 0, 0, 0, 0, 0};
    const float *q = a.q ? a.q + b * 4 : nullptr;
    int zc[3];
    zero_cells(a.H, a.W, a.az_res, a.vert_res, a.vert_off, zc);
    for (long n = (long)blockIdx.x * blockDim.x + threadIdx.x; n < a.npoints; n += (long)gridDim.x * blockDim.x) {
        const long i = (long)b * a.npoints + n;
        const int c = cell_of_pt[i];
        const long cell = (long)b * a.H * a.W + c;
        const unsigned rb = rbits[i];
        // who won the cell: the forward's rule (scatter_min_kernel) -- zero points win the cell they flagged
        const bool zero_cell = (zflag[b * ZFLAGS] == 0u && c == zc[0]) || (zflag[b * ZFLAGS + 1] == 0u && c == zc[1]) ||
                               (zflag[b * ZFLAGS + 2] == 0u && c == zc[2]);
        const bool win = rb == (zero_cell ? 0u : minr[cell]);
        if (a.grad_feat)
            for (int ch = 0; ch < a.C; ++ch) a.grad_feat[i * a.C + ch] = win ? a.grad_out_feat[cell * a.C + ch] : 0.0f;
        float G[3];
        for (int k = 0; k < 3; ++k) {
            G[k] = (win && a.grad_out_xyz ? a.grad_out_xyz[cell * 3 + k] : 0.0f) + (a.grad_warped ? a.grad_warped[i * 3 + k] : 0.0f);
        }
        const float x = a.xyz[i * 3], y = a.xyz[i * 3 + 1], z = a.xyz[i * 3 + 2];
        if (!q) {                                                  // no warp: the points themselves were projected
            if (a.grad_xyz) for (int k = 0; k < 3; ++k) a.grad_xyz[i * 3 + k] = G[k];
            continue;
        }
        const bool keep = !(x == 0.0f && y == 0.0f && z == 0.0f);
        if (!keep) { G[0] = G[1] = G[2] = 0.0f; }
        const float X[4] = {0.0f, x, y, z}, Xc[4] = {0.0f, -x, -y, -z};
        const float Gh[4] = {0.0f, G[0], G[1], G[2]}, Gc[4] = {0.0f, -G[0], -G[1], -G[2]};
        const float qc[4] = {q[0], -q[1], -q[2], -q[3]};
        const float n2 = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3] + 1e-10f;
        float t1[4], t2[4], u1[4], u2[4], M[4];
        hamilton(Gh, q, t1); hamilton(t1, Xc, u1);
        hamilton(Gc, q, t2); hamilton(t2, X, u2);
        hamilton(q, X, t1); hamilton(t1, qc, M);
        const float gm = (G[0] * M[1] + G[1] * M[2] + G[2] * M[3]) / (n2 * n2);
        for (int k = 0; k < 4; ++k) acc[k] += (u1[k] + u2[k]) / n2 - 2.0f * q[k] * gm;
        for (int k = 0; k < 3; ++k) acc[4 + k] += G[k];
        if (a.grad_xyz) {                                          // d pts / d x = R(q): x-gradient = R(q)^T G = (q* Ghat q)_vec / n2
            hamilton(qc, Gh, t1); hamilton(t1, q, u1);
            for (int k = 0; k < 3; ++k) a.grad_xyz[i * 3 + k] = u1[k + 1] / n2;
        }
    }
    if (!q) return;
    for (int k = 0; k < 7; ++k) {
        float v = acc[k];
        for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, ELO_WAVE);
        if ((threadIdx.x & 63) == 0) part[k][threadIdx.x >> 6] = v;
    }
    __syncthreads();
    if (threadIdx.x < 7) {
        float v = 0.0f;
        for (int w = 0; w < ELO_BLOCK / ELO_WAVE; ++w) v += part[threadIdx.x][w];
        if (threadIdx.x < 4) atomicAdd(a.grad_q + b * 4 + threadIdx.x, v);           // accumulated over the blocks of a batch element
        else atomicAdd(a.grad_t + b * 3 + (threadIdx.x - 4), v);
    }
}

#define ELO_REQUIRE(cond, who, what) \
    do { if (!(cond)) return fail(ELO_ERR_ARG, "%s: %s", who, what); } while (0)

unsigned grid_for(long items, unsigned cap = 8192)
{
    const long g = (items + ELO_BLOCK - 1) / ELO_BLOCK;
    return (unsigned)(g < 1 ? 1 : g > cap ? cap : g);
}

}  // namespace
}  // namespace elo

using namespace elo;

extern "C" int elo_group_concat_backward(const elo_group_concat_bwd_args *a, elo_stream_t stream)
{
    const char *who = "elo_group_concat_backward";
    ELO_REQUIRE(a, who, "null argument block");
    ELO_REQUIRE(a->batch >= 0 && a->npoints > 0 && a->K > 0 && a->H2 > 0 && a->W2 > 0 && a->C >= 0, who, "bad sizes");
    ELO_REQUIRE(a->grad_out && a->idx && a->mask, who, "null tensor pointer");
    const long rows = (long)a->batch * a->npoints * a->K;
    if (rows == 0) return ELO_OK;
    hipLaunchKernelGGL(group_concat_bwd_kernel, dim3(grid_for(rows * (3 + a->C))), dim3(ELO_BLOCK), 0, (hipStream_t)stream, *a);
    return check_launch(who);
}

extern "C" int elo_masked_maxpool_backward(const elo_masked_maxpool_bwd_args *a, elo_stream_t stream)
{
    const char *who = "elo_masked_maxpool_backward";
    ELO_REQUIRE(a, who, "null argument block");
    ELO_REQUIRE(a->batch >= 0 && a->npoints > 0 && a->K > 0 && a->C > 0, who, "bad sizes");
    ELO_REQUIRE(a->x && a->mask && a->grad_out && a->grad_x, who, "null tensor pointer");
    const long n = (long)a->batch * a->npoints * a->C;
    if (n == 0) return ELO_OK;
    if (a->C % 4 == 0 && (((uintptr_t)a->x | (uintptr_t)a->grad_out | (uintptr_t)a->grad_x) & 15) == 0 &&
        (a->K == 4 || a->K == 8 || a->K == 16 || a->K == 32)) {
        const long items = n / 4;
        const dim3 grid((unsigned)((items + ELO_BLOCK - 1) / ELO_BLOCK));
        hipStream_t s = (hipStream_t)stream;
        switch (a->K) {
        case 4: hipLaunchKernelGGL(masked_maxpool_bwd_vec_kernel<4>, grid, dim3(ELO_BLOCK), 0, s, *a, items); break;
        case 8: hipLaunchKernelGGL(masked_maxpool_bwd_vec_kernel<8>, grid, dim3(ELO_BLOCK), 0, s, *a, items); break;
        case 16: hipLaunchKernelGGL(masked_maxpool_bwd_vec_kernel<16>, grid, dim3(ELO_BLOCK), 0, s, *a, items); break;
        default: hipLaunchKernelGGL(masked_maxpool_bwd_vec_kernel<32>, grid, dim3(ELO_BLOCK), 0, s, *a, items); break;
        }
        return check_launch(who);
    }
    hipLaunchKernelGGL(masked_maxpool_bwd_kernel, dim3(grid_for(n)), dim3(ELO_BLOCK), 0, (hipStream_t)stream, *a);
    return check_launch(who);
}

extern "C" int elo_cv_encode1_backward(const elo_cv_encode1_bwd_args *a, elo_stream_t stream)
{
    const char *who = "elo_cv_encode1_backward";
    ELO_REQUIRE(a, who, "null argument block");
    ELO_REQUIRE(a->batch >= 0 && a->npoints > 0 && a->K > 0 && a->H2 > 0 && a->W2 > 0 && a->C > 0, who, "bad sizes");
    ELO_REQUIRE(a->xyz1 && a->xyz2 && a->idx && a->mask && a->grad_out, who, "null tensor pointer");
    const long rows = (long)a->batch * a->npoints * a->K;
    if (rows == 0) return ELO_OK;
    hipLaunchKernelGGL(cv_encode1_bwd_kernel, dim3(grid_for(rows * a->C)), dim3(ELO_BLOCK), 0, (hipStream_t)stream, *a);
    return check_launch(who);
}

extern "C" int elo_cv_encode2_backward(const elo_cv_encode2_bwd_args *a, elo_stream_t stream)
{
    const char *who = "elo_cv_encode2_backward";
    ELO_REQUIRE(a, who, "null argument block");
    ELO_REQUIRE(a->batch >= 0 && a->npoints > 0 && a->K > 0 && a->H > 0 && a->W > 0 && a->C > 0 && a->Cc > 0, who, "bad sizes");
    ELO_REQUIRE(a->npoints == a->H * a->W, who, "npoints must equal H*W (every pixel is a centre)");
    ELO_REQUIRE(a->xyz1 && a->idx && a->mask && a->grad_xyz_cat && a->grad_rest, who, "null tensor pointer");
    const long rows = (long)a->batch * a->npoints * a->K;
    if (rows == 0) return ELO_OK;
    hipLaunchKernelGGL(cv_encode2_bwd_kernel, dim3(grid_for(rows * a->Cc)), dim3(ELO_BLOCK), 0, (hipStream_t)stream, *a);
    return check_launch(who);
}

extern "C" int elo_masked_softmax_pool_backward(const elo_softmax_pool_bwd_args *a, elo_stream_t stream)
{
    const char *who = "elo_masked_softmax_pool_backward";
    ELO_REQUIRE(a, who, "null argument block");
    ELO_REQUIRE(a->batch >= 0 && a->npoints > 0 && a->K > 0 && a->C > 0 && a->values_stride >= a->C, who, "bad sizes");
    ELO_REQUIRE(a->logits && a->values && a->mask && a->grad_out && a->grad_logits && a->grad_values, who, "null tensor pointer");
    const long n = (long)a->batch * a->npoints * a->C;
    if (n == 0) return ELO_OK;
    if (a->C % 4 == 0 && a->values_stride % 4 == 0 && (a->K == 4 || a->K == 6 || a->K == 8 || a->K == 16 || a->K == 32) &&
        (((uintptr_t)a->logits | (uintptr_t)a->values | (uintptr_t)a->grad_out | (uintptr_t)a->grad_logits | (uintptr_t)a->grad_values) & 15) == 0) {
        const long items = n / 4;
        const dim3 grid((unsigned)((items + ELO_BLOCK - 1) / ELO_BLOCK));
        hipStream_t s = (hipStream_t)stream;
        if (a->K == 4) hipLaunchKernelGGL(softmax_pool_bwd_vec_kernel<4>, grid, dim3(ELO_BLOCK), 0, s, *a, items);
        else if (a->K == 6) hipLaunchKernelGGL(softmax_pool_bwd_vec_kernel<6>, grid, dim3(ELO_BLOCK), 0, s, *a, items);
        else if (a->K == 8) hipLaunchKernelGGL(softmax_pool_bwd_vec_kernel<8>, grid, dim3(ELO_BLOCK), 0, s, *a, items);
        else if (a->K == 16) hipLaunchKernelGGL(softmax_pool_bwd_vec2_kernel<16>, grid, dim3(ELO_BLOCK), 0, s, *a, items);
        else hipLaunchKernelGGL(softmax_pool_bwd_vec2_kernel<32>, grid, dim3(ELO_BLOCK), 0, s, *a, items);
        return check_launch(who);
    }
    hipLaunchKernelGGL(softmax_pool_bwd_kernel, dim3(grid_for(n)), dim3(ELO_BLOCK), 0, (hipStream_t)stream, *a);
    return check_launch(who);
}

extern "C" int elo_softmax_valid_backward(const elo_softmax_valid_bwd_args *a, elo_stream_t stream)
{
    const char *who = "elo_softmax_valid_backward";
    ELO_REQUIRE(a, who, "null argument block");
    ELO_REQUIRE(a->batch >= 0 && a->npoints > 0 && a->C > 0, who, "bad sizes");
    ELO_REQUIRE(a->feature && a->weight && a->xyz && a->grad_out && a->grad_feature && a->grad_weight, who, "null tensor pointer");
    if (a->batch == 0) return ELO_OK;
    ELO_REQUIRE((a->out == nullptr) == (a->stats == nullptr), who, "out and stats go together");
    if (a->stats && a->C % 4 == 0) {
        const long total = (long)a->batch * a->npoints * (a->C / 4);
        const long blocks = (total + ELO_BLOCK - 1) / ELO_BLOCK;
        hipLaunchKernelGGL(softmax_valid_bwd_elementwise_kernel, dim3((unsigned)(blocks > 4096 ? 4096 : blocks)), dim3(ELO_BLOCK), 0,
                           (hipStream_t)stream, *a);
        return check_launch(who);
    }
    hipLaunchKernelGGL(softmax_valid_bwd_kernel, dim3((a->C + ELO_WAVE - 1) / ELO_WAVE, a->batch), dim3(ELO_BLOCK), 0,
                       (hipStream_t)stream, *a);
    return check_launch(who);
}

extern "C" int elo_warp_project_backward(const elo_warp_project_bwd_args *a, elo_stream_t stream)
{
    const char *who = "elo_warp_project_backward";
    ELO_REQUIRE(a, who, "null argument block");
    ELO_REQUIRE(a->batch >= 0 && a->npoints > 0 && a->C >= 0 && a->H > 0 && a->W > 0, who, "bad sizes");
    ELO_REQUIRE(a->xyz && a->scratch, who, "null tensor pointer");
    ELO_REQUIRE(!a->q || (a->grad_q && a->grad_t), who, "a warped projection needs grad_q and grad_t");
    ELO_REQUIRE(!a->grad_feat || (a->grad_out_feat && a->C > 0), who, "grad_feat without grad_out_feat");
    if (a->batch == 0) return ELO_OK;
    const unsigned gx = grid_for(a->npoints, 64);
    hipLaunchKernelGGL(warp_project_bwd_kernel, dim3(gx, a->batch), dim3(ELO_BLOCK), 0, (hipStream_t)stream, *a);
    return check_launch(who);
}
