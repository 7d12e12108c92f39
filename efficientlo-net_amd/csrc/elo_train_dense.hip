// elo_train_dense.hip -- the dense products of a TRAINING layer, gfx950: out = x W (+ b) on (rows, Cin) matrices with rows up to
// ~10^6 and Cin, Cout <= ~200 (utils/tf_util.py:120-185: conv2d 1x1 = one product per layer; its adjoint dx = dz W^T).
//
// Why not the library: these products are TALL AND SKINNY -- 921 600 x 6 -> 8, 230 400 x 67 -> 128 -- and the GEMM kernels torch
// dispatches to tile for square problems (MT16x256x32 for an 8-wide output): tools/train_layer_shapes.py measures 57-84 us for
// layers whose operands are 6-22 us of HBM traffic.  One pass over the rows, here:
//   * W (Cin x Cout, <= 160 KB) is staged ONCE per workgroup into LDS, already in v_mfma_f32_16x16x4_f32 operand order
//     (one ds_read_b128 per lane = the A operands of four MFMA steps, conflict-free);
//   * a wave owns blocks of 16 rows x ALL output columns: x streams from HBM straight into the B operand -- 16-byte loads
//     when Cin % 4 == 0 (the reduction index is visited in the order the loads deliver it: lane (row j, quarter q) holds
//     k = 16c + 4q + e for step e of chunk c, and W was staged in the same order), 4-byte loads otherwise (k = 16c + 4e + q);
//   * D[i][j] with i = output channel, j = row: a lane ends up with FOUR CONSECUTIVE channels of one row = one 16-byte store;
//   * bias and (STATS) the per-channel sum and sum of squares of the output -- batch norm's moments -- are taken from the
//     accumulators: the separate elo_bn_stats pass over z (a full read of the layer's output) disappears.  Partials per
//     workgroup, combined in fp64 in a fixed order by bn_finalize (elo_train.hip): no atomics, same bits every run.
// fp32 operands, fp32 accumulation: the arithmetic of the library kernels it replaces (which run the same MFMA).
#include <hip/hip_runtime.h>

#include <type_traits>

#include "elo_common.h"

namespace elo {
void bn_finalize_launch(const float *part, int parts, long M, int C, float eps, float momentum, float *mean, float *invstd,
                        float *running_mean, float *running_var, int groups, hipStream_t s);      // elo_train.hip
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int DTB = 256;

// the sum over the 16 lanes of a DPP row, in every lane of it (quad butterflies, then the two mirrors: no LDS crossbar)
__device__ __forceinline__ float row_sum(float v)
{
    auto dpp = [](float x, auto ctrl) {
        const int u = (int)__float_as_uint(x);
        return __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(u, u, decltype(ctrl)::value, 0xf, 0xf, false));
    };
    v += dpp(v, std::integral_constant<int, 0xB1>{});      // quad_perm [1,0,3,2]
    v += dpp(v, std::integral_constant<int, 0x4E>{});      // quad_perm [2,3,0,1]
    v += dpp(v, std::integral_constant<int, 0x141>{});     // row_half_mirror
    v += dpp(v, std::integral_constant<int, 0x140>{});     // row_mirror
    return v;
}

template <int NT, bool BNB = false> struct RowBlocks { static constexpr int value = NT <= 1 ? (BNB ? 4 : 8) : NT <= 4 ? (BNB && NT <= 2 ? 2 : 4) : 2; };   // (BNB: two operand streams)
template <int NT> struct MinBlocks { static constexpr int value = NT <= 2 ? 4 : 2; };     // workgroups per CU the register budget allows

// BNB: the x operand is not read but FORMED -- batch norm's backward applied on the way in (x holds dy; bn.z the layer's
// pre-normalisation output, its moments / gamma / beta and the two sums elo_bn_backward's reduction left): the arithmetic of
// bn_bwd_apply_kernel per element, dz = gamma invstd (g - sum_g / M - xhat sum_gxhat / M); the wave that loads an element also WRITES dz
// (the weight gradient reads it), so the bn_bwd_apply launch and its pass over (rows, C) are gone from the big layers' backward.
struct BnBack {
    const float *z, *mean, *invstd, *gamma, *beta, *sums;
    float *dz;
    int relu;
    float inv_m;
};

// out(rows, N) = x(rows, K) W (+ bias);  W: (K, N) row-major, or (N, K) row-major when `transposed` (out = x W^T).
// nt = ceil(N / 16) <= NT.  part (STATS): [block][2][N] partial sums of out and out^2.
template <int NT, bool VEC, bool STATS, bool FULL, bool BNB>  // FULL: nt == NT, no per-tile branch splits the MFMA stream
__global__ __launch_bounds__(DTB, MinBlocks<NT>::value) void dense_rows_kernel(const float *__restrict__ x, const float *__restrict__ W, const float *__restrict__ bias,
                                                         float *__restrict__ out, long M, int K, int N, int nt, int transposed,
                                                         float *__restrict__ part, const BnBack bn)
{
    constexpr int RB = RowBlocks<NT, BNB>::value;
    extern __shared__ float4 wl[];                          // [KC][nt][64] x 4 steps, then (STATS) the waves' column sums
    const int KC = (K + 15) >> 4;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), j = lane & 15, q = lane >> 4;
    // blockIdx.y: the GROUP of rows (M rows each) with its own batch statistics -- the frames of a Siamese batch in one launch
    const float *const x_all = x;
    x += (size_t)blockIdx.y * M * K;
    out += (size_t)blockIdx.y * M * N;
    if (STATS) part += (size_t)blockIdx.y * gridDim.x * 2 * N;
    // W into operand order: zeros first (the padding of the last chunk and tile), then W in MEMORY order -- coalesced whichever
    // way it is stored -- scattered to where the MFMA steps read it; 8 loads in flight per thread (the first form walked the LDS
    // image and gathered from W one dependent trip at a time: 25 us of a 35 us launch on a 192 x 128 layer)
    for (int idx = threadIdx.x; idx < KC * nt * 64; idx += DTB) wl[idx] = float4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    {
        // 16-byte loads where the storage allows: four consecutive n of one k (W as stored, N % 4 == 0) land in four lanes' slots;
        // four consecutive k of one n (W^T, K % 4 == 0) are the four steps of one lane = one 16-byte LDS store
        float *wf = reinterpret_cast<float *>(wl);
        const int inner = transposed ? K : N, outer = transposed ? N : K;
        auto slot = [&](int k, int n) {
            const int c = k >> 4, kk = k & 15, lq = VEC ? kk >> 2 : kk & 3, e = VEC ? kk & 3 : kk >> 2;
            return (((c * nt + (n >> 4)) * 64 + lq * 16 + (n & 15)) << 2) + e;
        };
        if ((inner & 3) == 0 && (!transposed || VEC) && ((uintptr_t)W & 15) == 0) {       // (a weight is a view of the flat parameter buffer: any 4-byte offset)
            const int iq = inner >> 2, total = outer * iq;
            for (int f0 = threadIdx.x; f0 < total; f0 += DTB * 8) {
                float4 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) { const int f = f0 + u * DTB; v[u] = reinterpret_cast<const float4 *>(W)[f < total ? f : 0]; }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int f = f0 + u * DTB;
                    if (f < total) {
                        const int row = small_div(f, iq), col = (f - row * iq) << 2;
                        if (transposed) *reinterpret_cast<float4 *>(wf + slot(col, row)) = v[u];
                        else { wf[slot(row, col)] = v[u].x; wf[slot(row, col + 1)] = v[u].y; wf[slot(row, col + 2)] = v[u].z; wf[slot(row, col + 3)] = v[u].w; }
                    }
                }
            }
        } else {
            const int total = K * N;
            for (int f0 = threadIdx.x; f0 < total; f0 += DTB * 8) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) { const int f = f0 + u * DTB; v[u] = W[f < total ? f : 0]; }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int f = f0 + u * DTB;
                    if (f < total) {
                        const int row = small_div(f, inner), col = f - row * inner;
                        wf[transposed ? slot(col, row) : slot(row, col)] = v[u];
                    }
                }
            }
        }
    }
    // BNB: per channel quad of the reduction index, seven vectors: xhat = (z - B) A and pre-activation = xhat G + Bt EXACTLY as the forward's
    // bn_apply and bn_bwd_apply form them (the ReLU decision must be the forward's, bit for bit);  dz = P g + Q + R xhat
    float4 *tbl = wl + KC * nt * 64;
    if (BNB) {
        const float *gmean = bn.mean + (size_t)blockIdx.y * K, *ginv = bn.invstd + (size_t)blockIdx.y * K, *gsums = bn.sums + (size_t)blockIdx.y * 2 * K;
        for (int i = threadIdx.x; i < (K >> 2); i += DTB) {
            const float4 m = reinterpret_cast<const float4 *>(gmean)[i], sd = reinterpret_cast<const float4 *>(ginv)[i];
            const float4 g = reinterpret_cast<const float4 *>(bn.gamma)[i], b = reinterpret_cast<const float4 *>(bn.beta)[i];
            const float4 s1 = reinterpret_cast<const float4 *>(gsums)[i], s2 = reinterpret_cast<const float4 *>(gsums + K)[i];
            const float im = bn.inv_m;
            tbl[i * 7 + 0] = sd;
            tbl[i * 7 + 1] = m;
            tbl[i * 7 + 2] = g;
            tbl[i * 7 + 3] = b;
            const float4 P{g.x * sd.x, g.y * sd.y, g.z * sd.z, g.w * sd.w};
            tbl[i * 7 + 4] = P;
            tbl[i * 7 + 5] = float4{-P.x * s1.x * im, -P.y * s1.y * im, -P.z * s1.z * im, -P.w * s1.w * im};
            tbl[i * 7 + 6] = float4{-P.x * s2.x * im, -P.y * s2.y * im, -P.z * s2.z * im, -P.w * s2.w * im};
        }
    }
    __syncthreads();
    const bool vec_store = (N & 3) == 0;
    f32x4 bq[NT];                                           // this lane's four channels of every tile: 16t + 4q ..
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { const int n = 16 * t + 4 * q + r; bq[t][r] = bias && t < nt && n < N ? bias[n] : 0.f; }
    }
    f32x4 s1[STATS ? NT : 1], s2[STATS ? NT : 1];
    if (STATS) {
#pragma unroll
        for (int t = 0; t < NT; ++t) { s1[t] = f32x4{0.f, 0.f, 0.f, 0.f}; s2[t] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    }
    const long macro = (M + 16 * RB - 1) / (16 * RB);
    const int waves = DTB / 64;
    // One chunk = 16 values of the reduction index = 4 MFMA steps per tile and row block.  The wide forms (2 waves per SIMD: the
    // accumulators) run the chunks of ALL their macro-blocks as one stream through a ring of D chunk buffers: chunk c + D - 1 -- of
    // the next macro-block near the end of this one -- is in flight during chunk c's MFMAs.  (Distance 1 inside one macro-block was
    // the first form: a chunk is ~0.85 us of MFMA, a load from HBM 2-3 us -- the matrix pipe idled half the time.)  A macro-block
    // is padded to a multiple of D slots so that the ring position is static; the narrow forms (D = 1) load and use in place.
    constexpr int D = NT >= 8 ? 4 : NT >= 4 ? 3 : 1;        // (NT = 4 keeps four row blocks: a ring of 3 is what its registers hold)
    const int KCp = (KC + D - 1) / D * D;
    const long first = (long)blockIdx.x * waves + wave, stride = (long)gridDim.x * waves;
    auto rows_of = [&](long mb, const float *(&xr)[RB]) {
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) { const long row = mb * (16 * RB) + rb * 16 + j; xr[rb] = x + (size_t)(row < M ? row : M - 1) * K; }
    };
    // (the loads only: columns past K are read from a clamped address and weighted out WHERE THEY ARE USED -- weighting them here
    //  made every fetch wait for its own loads, i.e. no prefetch at all: the first form's matrix pipe idled 60 % of the time)
    const long zoff = BNB ? bn.z - x_all : 0;               // z and dz have x's shape: the same element offsets
    auto fetch = [&](f32x4 (&dst)[RB], f32x4 (&zdst)[BNB ? RB : 1], const float *const (&xr)[RB], int c) {
        if (VEC) {
            const int k0 = 16 * c + 4 * q, kc = k0 < K ? k0 : 0;
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                const float4 v = *reinterpret_cast<const float4 *>(xr[rb] + kc);
                dst[rb] = f32x4{v.x, v.y, v.z, v.w};
                if (BNB) { const float4 w = *reinterpret_cast<const float4 *>(xr[rb] + zoff + kc); zdst[rb] = f32x4{w.x, w.y, w.z, w.w}; }
            }
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int k = 16 * c + 4 * e + q, kc = k < K ? k : 0;
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) dst[rb][e] = xr[rb][kc];
            }
        }
    };
    auto weights = [&](int c) {                              // 1 for the reduction indices < K of chunk c, 0 past them (a multiply, not a
        f32x4 m;                                            //  select: the compiler turns a select into a predicated load -- elo_train.hip)
#pragma unroll
        for (int e = 0; e < 4; ++e) m[e] = (VEC ? 16 * c + 4 * q : 16 * c + 4 * e + q) < K ? 1.f : 0.f;
        return m;
    };
    const float *xc[RB], *xn[RB];
    f32x4 xb[D][RB], zb[BNB ? D : 1][BNB ? RB : 1];
    rows_of(first < macro ? first : 0, xc);
    if (D > 1) {
#pragma unroll
        for (int d = 0; d < D - 1; ++d) fetch(xb[d], zb[BNB ? d : 0], xc, d);
    }
#pragma unroll 1
    for (long mb = first; mb < macro; mb += stride) {
        const long row0 = mb * (16 * RB);
        const bool more = mb + stride < macro;
        rows_of(more ? mb + stride : mb, xn);
        f32x4 acc[RB][NT];
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[rb][t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
        for (int c0 = 0; c0 < KCp; c0 += D) {
#pragma unroll
            for (int u = 0; u < D; ++u) {
                const int c = c0 + u;
                if (D > 1) {
                    // UNCONDITIONAL loads (a padding slot or the slot after the last macro-block re-reads valid rows and is never
                    // used): behind a branch the compiler cannot count the loads in flight and waits for ALL of them (vmcnt(0))
                    // before the next use -- the prefetch distance collapses to zero
                    const int cf = c + D - 1;
                    const bool here = cf < KCp;
                    const float *xf[RB];
#pragma unroll
                    for (int rb = 0; rb < RB; ++rb) xf[rb] = here ? xc[rb] : xn[rb];
                    fetch(xb[(u + D - 1) % D], zb[BNB ? (u + D - 1) % D : 0], xf, here ? cf : cf - KCp);
                } else fetch(xb[0], zb[0], xc, c);
                if (c < KC) {
                    if (BNB) {
                        const int k0 = 16 * c + 4 * q, i = (k0 < K ? k0 : 0) >> 2;
                        const float4 tA = tbl[i * 7 + 0], tB = tbl[i * 7 + 1], tG = tbl[i * 7 + 2], tT = tbl[i * 7 + 3];
                        const float4 tP = tbl[i * 7 + 4], tQ = tbl[i * 7 + 5], tR = tbl[i * 7 + 6];
                        const f32x4 A{tA.x, tA.y, tA.z, tA.w}, B{tB.x, tB.y, tB.z, tB.w}, G{tG.x, tG.y, tG.z, tG.w}, T{tT.x, tT.y, tT.z, tT.w};
                        const f32x4 P{tP.x, tP.y, tP.z, tP.w}, Q{tQ.x, tQ.y, tQ.z, tQ.w}, R{tR.x, tR.y, tR.z, tR.w};
#pragma unroll
                        for (int rb = 0; rb < RB; ++rb) {
                            f32x4 xh = zb[BNB ? u : 0][rb] - B;
                            xh *= A;
                            f32x4 g = xb[u][rb];
                            if (bn.relu) {
                                const f32x4 pre = xh * G + T;
#pragma unroll
                                for (int e = 0; e < 4; ++e) g[e] = pre[e] > 0.f ? g[e] : 0.f;
                            }
                            const f32x4 d = P * g + Q + R * xh;
                            xb[u][rb] = d;
                            const long row = row0 + rb * 16 + j;
                            if (row < M && k0 < K) *reinterpret_cast<float4 *>(bn.dz + (xc[rb] - x_all) + k0) = float4{d[0], d[1], d[2], d[3]};
                        }
                    }
                    if (16 * c + 16 > K) {                  // the ragged last chunk
                        const f32x4 m = weights(c);
#pragma unroll
                        for (int rb = 0; rb < RB; ++rb) xb[u][rb] *= m;
                    }
                    if (FULL) {
                        // (straight-line: the compiler places the LDS reads between the MFMAs itself.  The guarded form below moves
                        //  each prefetched operand into place with a VALU copy in the middle of the stream, and one extra issue slot
                        //  between MFMAs costs tens of cycles -- MI355X_MICROARCH: the pipe was 44 % busy, the waves 53 % in
                        //  SQ_WAIT_INST_ANY)
                        float4 aw[NT];                       // all of the chunk's operands on their way before the first MFMA
#pragma unroll
                        for (int t = 0; t < NT; ++t) aw[t] = wl[(c * NT + t) * 64 + lane];
#pragma unroll
                        for (int t = 0; t < NT; ++t) {
                            const float as[4] = {aw[t].x, aw[t].y, aw[t].z, aw[t].w};
#pragma unroll
                            for (int e = 0; e < 4; ++e)
#pragma unroll
                                for (int rb = 0; rb < RB; ++rb)
                                    acc[rb][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(as[e], xb[u][rb][e], acc[rb][t], 0, 0, 0);
                        }
                    } else {
                        float4 an = wl[(c * nt) * 64 + lane];
#pragma unroll
                        for (int t = 0; t < NT; ++t) {
                            if (t < nt) {
                                const float4 a = an;
                                if (t + 1 < NT && t + 1 < nt) an = wl[(c * nt + t + 1) * 64 + lane];     // the next tile's operands, behind this tile's MFMAs
                                const float as[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
                                for (int e = 0; e < 4; ++e)  // (row blocks innermost: consecutive MFMAs are independent)
#pragma unroll
                                    for (int rb = 0; rb < RB; ++rb)
                                        acc[rb][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(as[e], xb[u][rb][e], acc[rb][t], 0, 0, 0);
                            }
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) xc[rb] = xn[rb];
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            const long row = row0 + rb * 16 + j;
            const bool live = row < M;
            const float lv = live ? 1.f : 0.f;
            float *o = out + (size_t)(live ? row : 0) * N;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                if (t < nt) {
                    const int n0 = 16 * t + 4 * q;
                    const f32x4 v = acc[rb][t] + bq[t];
                    if (live && n0 < N) {
                        if (vec_store) *reinterpret_cast<float4 *>(o + n0) = float4{v[0], v[1], v[2], v[3]};
                        else {
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                if (n0 + r < N) o[n0 + r] = v[r];
                        }
                    }
                    // (columns >= N hold exact zeros -- W and the bias were staged as 0 there --, dead rows are weighted out)
                    if (STATS) { const f32x4 w = v * lv; s1[t] += w; s2[t] += w * v; }
                }
            }
        }
    }
    if (STATS) {
        // totals over the 16 rows a quarter-wave holds, then over the waves through LDS (the staged W is dead by now)
        __syncthreads();
        float *red = reinterpret_cast<float *>(wl);         // [wave][2][16 * nt]
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            if (t < nt) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float a = row_sum(s1[t][r]), b = row_sum(s2[t][r]);
                    if (j == 0) { red[(wave * 2 + 0) * 16 * nt + 16 * t + 4 * q + r] = a; red[(wave * 2 + 1) * 16 * nt + 16 * t + 4 * q + r] = b; }
                }
            }
        }
        __syncthreads();
        for (int i = threadIdx.x; i < 2 * N; i += DTB) {
            const int which = i / N, n = i - which * N;
            float tot = 0.f;
#pragma unroll
            for (int w = 0; w < waves; ++w) tot += red[(w * 2 + which) * 16 * nt + n];
            part[((size_t)blockIdx.x * 2 + which) * N + n] = tot;
        }
    }
}

struct Plan { int NT, nt, KC, grid, groups; long rows; size_t lds; };      // rows: per group

bool plan_dense(long M, int K, int N, bool stats, Plan &p, bool bnb = false, int groups = 1)
{
    p.groups = (stats || bnb) && groups > 1 ? groups : 1;   // (the plain product has no per-group state: one group of all rows)
    if (M % p.groups) return false;
    M /= p.groups;
    p.rows = M;
    p.nt = (N + 15) / 16;
    p.NT = p.nt <= 1 ? 1 : p.nt <= 2 ? 2 : p.nt <= 4 ? 4 : p.nt <= 8 ? 8 : 12;
    if (p.nt > 12) return false;
    p.KC = (K + 15) / 16;
    p.lds = (size_t)p.KC * p.nt * 64 * sizeof(float4);
    const size_t red = stats ? (size_t)(DTB / 64) * 2 * 16 * p.nt * sizeof(float) : 0;
    if (p.lds < red) p.lds = red;
    if (bnb) p.lds += (size_t)(K / 4) * 7 * sizeof(float4);
    if (p.lds > 160 * 1024) return false;
    // a resident grid: the workgroups the 256 CUs hold at once (registers: 4 per CU for the narrow forms, 2 for the wide; LDS),
    // twice that for the narrow streaming forms
    const int rb = p.NT <= 1 ? (bnb ? 4 : 8) : p.NT <= 4 ? (bnb && p.NT <= 2 ? 2 : 4) : 2;          // RowBlocks<NT, BNB>
    const long macro = (M + 16 * rb - 1) / (16 * rb), blocks = (macro + DTB / 64 - 1) / (DTB / 64);
    int per_cu = p.NT <= 2 ? 4 : 2;
    const int by_lds = (int)((160 * 1024) / (p.lds ? p.lds : 1));
    per_cu = by_lds < per_cu ? (by_lds < 1 ? 1 : by_lds) : per_cu;
    long cap = 256l * per_cu * (p.NT <= 2 ? 2 : 1);
    cap = cap > ELO_DENSE_MAX_PARTS ? ELO_DENSE_MAX_PARTS : cap;
    p.grid = (int)(blocks > cap ? cap : blocks);
    return true;
}

template <int NT, bool VEC, bool STATS, bool FULL, bool BNB = false>
int launch_one(const Plan &p, const elo_dense_rows_args *a, hipStream_t s)
{
    auto kern = dense_rows_kernel<NT, VEC, STATS, FULL, BNB>;
    const BnBack bn{a->bn_z, a->bn_mean, a->bn_invstd, a->bn_gamma, a->bn_beta, a->bn_sums, a->bn_dz, a->bn_relu, 1.0f / (float)p.rows};
    static bool raised = false;                              // > 64 KB of dynamic LDS has to be asked for, once per kernel
    if (!raised && p.lds > 64 * 1024) {
        if (hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            return fail(ELO_ERR_LAUNCH, "elo_dense_rows: %zu bytes of LDS refused", p.lds);
        raised = true;
    }
    hipLaunchKernelGGL(kern, dim3(p.grid, p.groups), dim3(DTB), p.lds, s, a->x, a->W, a->bias, a->out, p.rows, a->Cin, a->Cout, p.nt,
                       a->transposed, a->scratch, bn);
    return ELO_OK;
}

template <int NT>
int launch_nt(const Plan &p, const elo_dense_rows_args *a, hipStream_t s)
{
    const bool vec = (a->Cin & 3) == 0, stats = a->scratch != nullptr;
    const bool full = p.nt == NT;
    if (a->bn_z) return full ? launch_one<NT, true, false, true, true>(p, a, s) : launch_one<NT, true, false, false, true>(p, a, s);
#define ELO_DR(V, S) (full ? launch_one<NT, V, S, true>(p, a, s) : launch_one<NT, V, S, false>(p, a, s))
    if (vec) return stats ? ELO_DR(true, true) : ELO_DR(true, false);
    return stats ? ELO_DR(false, true) : ELO_DR(false, false);
#undef ELO_DR
}

}  // namespace
}  // namespace elo

using namespace elo;

extern "C" long elo_dense_rows_scratch_floats(int Cout, int groups) { return 2l * Cout * ELO_DENSE_MAX_PARTS * (groups > 1 ? groups : 1); }

extern "C" int elo_dense_rows_supported(long rows, int Cin, int Cout)
{
    Plan p;
    return rows > 0 && Cin > 0 && Cout > 0 && plan_dense(rows, Cin, Cout, true, p) ? 1 : 0;
}

extern "C" int elo_dense_rows(const elo_dense_rows_args *a, elo_stream_t stream)
{
    const char *who = "elo_dense_rows";
    if (!a) return fail(ELO_ERR_ARG, "%s: null argument block", who);
    if (a->rows <= 0 || a->Cin <= 0 || a->Cout <= 0) return fail(ELO_ERR_ARG, "%s: bad sizes", who);
    if (!a->x || !a->W || !a->out) return fail(ELO_ERR_ARG, "%s: null tensor pointer", who);
    if (((uintptr_t)a->x | (uintptr_t)a->out) & 15) return fail(ELO_ERR_ARG, "%s: x and out must be 16-byte aligned", who);
    const bool stats = a->scratch != nullptr;
    if (stats && (!a->mean || !a->invstd)) return fail(ELO_ERR_ARG, "%s: moments asked for without mean / invstd", who);
    if (stats && (a->running_mean == nullptr) != (a->running_var == nullptr)) return fail(ELO_ERR_ARG, "%s: running_mean and running_var go together", who);
    const bool bnb = a->bn_z != nullptr;
    if (bnb) {
        if (stats || (a->Cin & 3) || !a->bn_mean || !a->bn_invstd || !a->bn_gamma || !a->bn_beta || !a->bn_sums || !a->bn_dz)
            return fail(ELO_ERR_ARG, "%s: the batch-norm-backward operand needs Cin %% 4 == 0, no moments, and all of mean / invstd / gamma / beta / sums / dz", who);
        if (((uintptr_t)a->bn_z | (uintptr_t)a->bn_dz | (uintptr_t)a->bn_mean | (uintptr_t)a->bn_invstd | (uintptr_t)a->bn_gamma | (uintptr_t)a->bn_beta |
             (uintptr_t)a->bn_sums) & 15)
            return fail(ELO_ERR_ARG, "%s: unaligned batch-norm-backward tensor", who);
    }
    Plan p;
    if (a->groups > 64) return fail(ELO_ERR_ARG, "%s: %d groups", who, a->groups);
    if (!plan_dense(a->rows, a->Cin, a->Cout, stats, p, bnb, a->groups))
        return fail(ELO_ERR_LIMIT, "%s: Cin = %d, Cout = %d does not fit (<= 192 output columns, W <= 160 KB of LDS), or %ld rows do not split into %d groups",
                    who, a->Cin, a->Cout, a->rows, a->groups);
    hipStream_t s = (hipStream_t)stream;
    int rc = ELO_OK;
    switch (p.NT) {
    case 1: rc = launch_nt<1>(p, a, s); break;
    case 2: rc = launch_nt<2>(p, a, s); break;
    case 4: rc = launch_nt<4>(p, a, s); break;
    case 8: rc = launch_nt<8>(p, a, s); break;
    default: rc = launch_nt<12>(p, a, s); break;
    }
    if (rc) return rc;
    if (stats)
        bn_finalize_launch(a->scratch, p.grid, p.rows, a->Cout, a->eps, a->momentum, a->mean, a->invstd, a->running_mean, a->running_var, p.groups, s);
    return check_launch(who);
}
