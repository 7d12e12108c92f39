// elo_fused.hip -- fused "gather -> 1x1-conv chain -> pool" inference kernels for MI355X (gfx950).
//
// Why: at batch 1 a frame pair is ~300 launches of 4-14 us each even after the per-operator
// fusion of elo_features.hip + hipBLASLt GEMMs with bias/ReLU epilogues (profiles/r01_b): the
// command processor, not HBM or the matrix cores, is the limit.  These kernels collapse every
// operator between two poolings into ONE launch.
//
// How: one wavefront owns a tile of 32 consecutive rows (a row = one (b,n,k) neighbour slot, or one
// point for the row-wise MLPs).  The tile's activations live in a wave-private LDS image
// act[32][S] (S/4 odd => ds_read_b128 of a column block is bank-conflict free).  A layer is
//     D[32 x Np] = A[32 x Kp] * W[Kp x Np] + bias      on v_mfma_f32_32x32x2_f32 (exact fp32)
// with A read from LDS 16 bytes per lane per 8 k's, W streamed from L2 in pre-packed B-fragment
// order (one contiguous 1 KiB load per wave-instruction, prefetched one step ahead) and D written
// back over the tile IN PLACE (a wave has consumed all its A reads before its first D write, and a
// wave's LDS operations execute in order), so consecutive layers need no barrier at all.
// Poolings (masked max / masked softmax-weighted sum over the K rows of a point) read the
// final activations column-wise from LDS and write (b,n,C) rows coalesced.
//
// Interfaces and the reference lines covered: include/elo.h ("Fused inference kernels").
#include "elo_common.h"

namespace elo {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int TILE = 32;           // rows per wave
constexpr int FUSED_BLOCK = 64;    // one wave per workgroup: tiles are wave-private

__device__ __forceinline__ int ceil8(int x) { return (x + 7) & ~7; }

// Phases of a tile hand data from lane to lane through the wave-private LDS image.  A wavefront's LDS
// operations execute in program order, so no s_barrier is needed; this only stops the COMPILER from
// moving LDS accesses across the hand-over.
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// LDS row stride (floats) for `cols` columns: a multiple of 4 with (S/4) odd.
__host__ __device__ __forceinline__ int row_stride(int cols)
{
    int s = (cols + 3) & ~3;
    if (((s >> 2) & 1) == 0) s += 4;
    return s;
}

// ---- one dense layer on the wave's tile ------------------------------------------------------
template <int NB>
__device__ __forceinline__ void dense_nb(float *act, int S, int in_off, int out_off, const elo_dense &L)
{
    const int lane = threadIdx.x & 63, col = lane & 31, half = lane >> 5;
    const int KS = ceil8(L.K) >> 3;
    f32x16 acc[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const float bv = L.bias[nb * 32 + col];
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nb][r] = bv;
    }
    const float *arow = act + col * S + in_off + 4 * half;
    const float4 *w = reinterpret_cast<const float4 *>(L.w_packed) + lane;
    float4 bnext[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) bnext[nb] = w[(size_t)(nb * KS) * 64];
    for (int ks = 0; ks < KS; ++ks) {
        const float4 a = *reinterpret_cast<const float4 *>(arow + ks * 8);
        float4 b[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) b[nb] = bnext[nb];
        if (ks + 1 < KS) {
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) bnext[nb] = w[(size_t)(nb * KS + ks + 1) * 64];
        }
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b[nb].x, acc[nb], 0, 0, 0);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b[nb].y, acc[nb], 0, 0, 0);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b[nb].z, acc[nb], 0, 0, 0);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b[nb].w, acc[nb], 0, 0, 0);
    }
    // C/D layout of the 32x32 tile: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
            float v = acc[nb][r];
            if (L.relu) v = fmaxf(v, 0.0f);
            act[row * S + out_off + nb * 32 + col] = v;
        }
    }
    wave_sync();
}

__device__ __forceinline__ void dense(float *act, int S, int in_off, int out_off, const elo_dense &L)
{
    switch ((L.N + 31) >> 5) {
    case 1: dense_nb<1>(act, S, in_off, out_off, L); break;
    case 2: dense_nb<2>(act, S, in_off, out_off, L); break;
    case 3: dense_nb<3>(act, S, in_off, out_off, L); break;
    default: dense_nb<4>(act, S, in_off, out_off, L); break;
    }
}

// ---- per-row gather metadata of a tile -------------------------------------------------------
struct TileMeta {
    int *cell;      // [TILE] flat (b*H2 + h)*W2 + w of the gathered pixel, -1 = row not in use
    float *mask;    // [TILE]
};

// rows of a tile = P points x K slots; lane r < TILE fetches its row's idx/mask once.
__device__ __forceinline__ void load_meta(const TileMeta &m, long first_point, long total_points, int P, int K,
                                          const int *__restrict__ idx, const float *__restrict__ mask, int H2, int W2)
{
    const int lane = threadIdx.x & 63;
    if (lane < TILE) {
        const int pi = lane / K;
        const long pt = first_point + pi;
        int cell = -1;
        float mk = 0.0f;
        if (pi < P && pt < total_points) {
            const long gr = pt * K + (lane - pi * K);
            const int *id = idx + gr * 3;
            cell = (id[0] * H2 + id[1]) * W2 + id[2];
            mk = mask[gr];
        }
        m.cell[lane] = cell;
        m.mask[lane] = mk;
    }
    wave_sync();
}

struct Geo { float p[3], g[3], d[3], euc; };

__device__ __forceinline__ float geo_channel(const float *p, const float *g, float m, int ch)
{   // [p, g*m, g*m - p, sqrt(sum((g*m-p)^2) + 1e-20)]   utils/pointnet_util.py:54-62
    float gm[3], d[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) { gm[i] = g[i] * m; d[i] = gm[i] - p[i]; }
    if (ch < 3) return p[ch];
    if (ch < 6) return gm[ch - 3];
    if (ch < 9) return d[ch - 6];
    return sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + 1e-20f);
}

// ---- poolings over the K rows of each point of the tile --------------------------------------
__device__ __forceinline__ void pool_masked_max(const float *act, int S, int off, int C, const TileMeta &m, int P, int K,
                                                long first_point, long total_points, float *__restrict__ out)
{
    const int lane = threadIdx.x & 63;
    for (int pi = 0; pi < P; ++pi) {
        const long pt = first_point + pi;
        if (pt >= total_points) break;
        for (int c = lane; c < C; c += 64) {
            float best = act[(pi * K) * S + off + c] * m.mask[pi * K];
            for (int k = 1; k < K; ++k) best = fmaxf(best, act[(pi * K + k) * S + off + c] * m.mask[pi * K + k]);
            out[pt * C + c] = best;
        }
    }
}

// out = sum_k softmax_k(mask == 1 ? logit : -1e10) * value      (64 channels, one per lane)
__device__ __forceinline__ void pool_masked_softmax(const float *act, int S, int logit_off, int value_off,
                                                    const TileMeta &m, int P, int K, long first_point,
                                                    long total_points, float *__restrict__ out)
{
    const int lane = threadIdx.x & 63;
    for (int pi = 0; pi < P; ++pi) {
        const long pt = first_point + pi;
        if (pt >= total_points) break;
        float mx = -INFINITY;
        for (int k = 0; k < K; ++k) {
            const float l = m.mask[pi * K + k] == 1.0f ? act[(pi * K + k) * S + logit_off + lane] : -1e10f;
            mx = fmaxf(mx, l);
        }
        float den = 0.0f, acc = 0.0f;
        for (int k = 0; k < K; ++k) {
            const float l = m.mask[pi * K + k] == 1.0f ? act[(pi * K + k) * S + logit_off + lane] : -1e10f;
            const float e = expf(l - mx);
            den += e;
            acc += e * act[(pi * K + k) * S + value_off + lane];
        }
        out[pt * 64 + lane] = acc / den;
    }
}

// ================================================================ set-conv / set-upconv stage 1
__global__ __launch_bounds__(FUSED_BLOCK) void setconv_kernel(const elo_setconv_args a, const int S)
{
    extern __shared__ float lds[];
    float *act = lds;
    TileMeta meta{reinterpret_cast<int *>(lds + TILE * S), lds + TILE * S + TILE};
    const int lane = threadIdx.x & 63;
    const int K = a.K, P = TILE / K;
    const long total_points = (long)a.batch * a.npoints;
    const long first_point = (long)xcd_tile(blockIdx.x, gridDim.x) * P;
    if (first_point >= total_points) return;
    load_meta(meta, first_point, total_points, P, K, a.idx, a.mask, a.H2, a.W2);

    // centres of the tile's points: cxyz[pi*3 + c] kept in LDS next to the metadata
    float *cxyz = meta.mask + TILE;
    if (lane < P * 3) {
        const int pi = lane / 3, c = lane - pi * 3;
        const long pt = first_point + pi;
        float v = 0.0f;
        if (pt < total_points) {
            if (a.centre_hw) {
                const int b = (int)(pt / a.npoints);
                const int h = a.centre_hw[pt * 2 + 0], w = a.centre_hw[pt * 2 + 1];
                v = a.xyz1_grid[(((long)b * a.H + h) * a.W + w) * 3 + c];
                if (a.new_xyz) a.new_xyz[pt * 3 + c] = v;                       // :206
            } else {
                v = a.centre_xyz[pt * 3 + c];
            }
        }
        cxyz[lane] = v;
    }
    wave_sync();
    // gather + centre-subtract + concat into act[row][0 .. CTp)                   :203-213 / :277-284
    const int CT = 3 + a.C, CTp = ceil8(CT);
    for (int e = lane; e < TILE * CTp; e += 64) {
        const int row = e / CTp, ch = e - row * CTp;
        const int cell = meta.cell[row];
        float v = 0.0f;
        if (cell >= 0 && ch < CT) {
            const float m = meta.mask[row];
            v = ch < 3 ? a.src_xyz[(long)cell * 3 + ch] * m - cxyz[(row / K) * 3 + ch]
                       : a.src_feat[(long)cell * a.C + (ch - 3)] * m;
        }
        act[row * S + ch] = v;
    }
    wave_sync();
    for (int l = 0; l < a.n_layers; ++l) dense(act, S, 0, 0, a.layers[l]);          // in place, :217-222
    pool_masked_max(act, S, 0, a.layers[a.n_layers - 1].N, meta, P, K, first_point, total_points, a.out);   // :224-230
}

// ================================================================ row-wise MLP over concatenated sources
__global__ __launch_bounds__(FUSED_BLOCK) void mlp_kernel(const elo_mlp_args a, const int S)
{
    extern __shared__ float lds[];
    float *act = lds;
    const int lane = threadIdx.x & 63;
    const long first = (long)xcd_tile(blockIdx.x, gridDim.x) * TILE;
    if (first >= a.rows) return;
    const int w0 = a.src_width[0], w1 = a.n_sources > 1 ? a.src_width[1] : 0, w2 = a.n_sources > 2 ? a.src_width[2] : 0;
    const int CT = w0 + w1 + w2, CTp = ceil8(CT);
    for (int e = lane; e < TILE * CTp; e += 64) {
        const int row = e / CTp, ch = e - row * CTp;
        const long gr = first + row;
        float v = 0.0f;
        if (gr < a.rows && ch < CT) {
            v = ch < w0 ? a.src[0][gr * w0 + ch]
              : ch < w0 + w1 ? a.src[1][gr * w1 + (ch - w0)]
                             : a.src[2][gr * w2 + (ch - w0 - w1)];
        }
        act[row * S + ch] = v;
    }
    wave_sync();
    for (int l = 0; l < a.n_layers; ++l) dense(act, S, 0, 0, a.layers[l]);
    const int N = a.layers[a.n_layers - 1].N;
    for (int e = lane; e < TILE * N; e += 64) {
        const int row = e / N, c = e - row * N;
        const long gr = first + row;
        if (gr < a.rows) a.out[gr * N + c] = act[row * S + c];
    }
}

// ================================================================ cost volume, stage 1
// LDS columns: [0,128) = X (CV chain, later [x3 | enc]),  [128, 128 + max(CTp,128)) = F (feat_cat, later sum_CV)
__global__ __launch_bounds__(FUSED_BLOCK) void cv1_kernel(const elo_cv1_args a, const int S)
{
    extern __shared__ float lds[];
    float *act = lds;
    TileMeta meta{reinterpret_cast<int *>(lds + TILE * S), lds + TILE * S + TILE};
    const int lane = threadIdx.x & 63;
    const int K = a.K, P = TILE / K, C = a.C;
    const long total_points = (long)a.batch * a.npoints;
    const long first_point = (long)xcd_tile(blockIdx.x, gridDim.x) * P;
    if (first_point >= total_points) return;
    load_meta(meta, first_point, total_points, P, K, a.idx, a.mask, a.H2, a.W2);
    const int F = 128, CT = 10 + 2 * C, CTp = ceil8(CT);
    for (int e = lane; e < TILE * CTp; e += 64) {                                     // :54-66
        const int row = e / CTp, ch = e - row * CTp;
        const int cell = meta.cell[row];
        float v = 0.0f;
        if (cell >= 0 && ch < CT) {
            const float m = meta.mask[row];
            const long pt = first_point + row / K;
            if (ch < 10) v = geo_channel(a.xyz1 + pt * 3, a.xyz2 + (long)cell * 3, m, ch);
            else if (ch < 10 + C) v = a.feat1[pt * C + (ch - 10)];
            else v = a.feat2[(long)cell * C + (ch - 10 - C)] * m;
        }
        act[row * S + F + ch] = v;
    }
    wave_sync();
    dense(act, S, F, 0, a.cv0);           // feat_cat -> 128                           :72-76
    dense(act, S, 0, 0, a.cv1);           // -> 64 (in place)
    dense(act, S, 0, 0, a.cv2);           // -> 64 = x                                  (values of the pooling)
    dense(act, S, F, 64, a.cv_xyz);       // xyz_cat (first 10 columns of F) -> enc at [64,128)      :79-82
    dense(act, S, 0, F, a.sum_cv0);       // [x | enc] -> 128 into F                    :84-90
    dense(act, S, F, F, a.sum_cv1);       // -> 64 logits (in place)
    pool_masked_softmax(act, S, F, 0, meta, P, K, first_point, total_points, a.out);    // :92-98
}

// ================================================================ cost volume, stage 2
// LDS columns: [0,64) grouped cost, [64,128) xyz-encoding, [128,128+C) feat1, [192,208) xyz_cat;
// sum_cost0 reads [0,128+C) and writes [64,192); sum_cost1 maps [64,192) -> [64,128).
__global__ __launch_bounds__(FUSED_BLOCK) void cv2_kernel(const elo_cv2_args a, const int S)
{
    extern __shared__ float lds[];
    float *act = lds;
    TileMeta meta{reinterpret_cast<int *>(lds + TILE * S), lds + TILE * S + TILE};
    const int lane = threadIdx.x & 63;
    const int K = a.K, P = TILE / K, C = a.C;
    const long total_points = (long)a.batch * a.npoints;
    const long first_point = (long)xcd_tile(blockIdx.x, gridDim.x) * P;
    if (first_point >= total_points) return;
    load_meta(meta, first_point, total_points, P, K, a.idx, a.mask, a.H, a.W);
    const int Cp = ceil8(C), XYZ = 192;
    for (int e = lane; e < TILE * 64; e += 64) {                                      // grouped cost * mask  :110
        const int row = e >> 6, ch = e & 63;
        const int cell = meta.cell[row];
        act[row * S + ch] = cell >= 0 ? a.cost[(long)cell * 64 + ch] * meta.mask[row] : 0.0f;
    }
    for (int e = lane; e < TILE * Cp; e += 64) {                                      // centre features      :115
        const int row = e / Cp, ch = e - row * Cp;
        const long pt = first_point + row / K;
        act[row * S + 128 + ch] = (meta.cell[row] >= 0 && ch < C) ? a.feat1[pt * C + ch] : 0.0f;
    }
    for (int e = lane; e < TILE * 16; e += 64) {                                      // 10-channel geometry  :111-120
        const int row = e >> 4, ch = e & 15;
        const int cell = meta.cell[row];
        const long pt = first_point + row / K;
        act[row * S + XYZ + ch] = (cell >= 0 && ch < 10)
            ? geo_channel(a.xyz1 + pt * 3, a.xyz1 + (long)cell * 3, meta.mask[row], ch) : 0.0f;
    }
    wave_sync();
    dense(act, S, XYZ, 64, a.xyz_enc);     // -> enc at [64,128)                        :123-126
    dense(act, S, 0, 64, a.sum_cost0);     // [grouped | enc | feat1] -> 128 at [64,192) :129-135
    dense(act, S, 64, 64, a.sum_cost1);    // -> 64 logits at [64,128)
    pool_masked_softmax(act, S, 64, 0, meta, P, K, first_point, total_points, a.out);   // :137-146
}

int check_dense(const elo_dense &L, int K, int N, const char *who, const char *name)
{
    if (!L.w_packed || !L.bias) return fail(ELO_ERR_ARG, "%s: layer %s has null weights", who, name);
    if (L.K != K || (N > 0 && L.N != N))
        return fail(ELO_ERR_ARG, "%s: layer %s is %dx%d, expected %dx%d", who, name, L.K, L.N, K, N);
    if (L.N <= 0 || L.N > 128) return fail(ELO_ERR_LIMIT, "%s: layer %s width %d outside 1..128", who, name, L.N);
    return ELO_OK;
}

size_t tile_lds_bytes(int S) { return sizeof(float) * ((size_t)TILE * S + 2 * TILE + 3 * TILE); }

#define ELO_REQUIRE(cond, who, what) \
    do { if (!(cond)) return fail(ELO_ERR_ARG, "%s: %s", who, what); } while (0)

}  // namespace
}  // namespace elo

using namespace elo;

extern "C" int elo_setconv_fused(const elo_setconv_args *a, elo_stream_t stream)
{
    const char *who = "elo_setconv_fused";
    ELO_REQUIRE(a, who, "null argument block");
    ELO_REQUIRE(a->batch >= 0 && a->npoints > 0 && a->K > 0 && a->H2 > 0 && a->W2 > 0 && a->C >= 0, who, "bad sizes");
    if (a->K > TILE) return fail(ELO_ERR_LIMIT, "%s: K = %d exceeds the %d-row tile", who, a->K, TILE);
    ELO_REQUIRE(a->n_layers >= 1 && a->n_layers <= ELO_MAX_CHAIN, who, "1..3 layers");
    ELO_REQUIRE(a->src_xyz && (a->src_feat || a->C == 0) && a->idx && a->mask && a->out, who, "null tensor pointer");
    ELO_REQUIRE((a->centre_hw && a->xyz1_grid && a->H > 0 && a->W > 0) || a->centre_xyz, who, "no centre source");
    int width = 3 + a->C, cols = (width + 7) & ~7;
    for (int l = 0; l < a->n_layers; ++l) {
        if (int rc = check_dense(a->layers[l], width, 0, who, "mlp")) return rc;
        width = a->layers[l].N;
        cols = cols > ((width + 31) & ~31) ? cols : ((width + 31) & ~31);
    }
    const long points = (long)a->batch * a->npoints;
    if (points == 0) return ELO_OK;
    const int P = TILE / a->K, S = row_stride(cols);
    const unsigned grid = (unsigned)((points + P - 1) / P);
    hipLaunchKernelGGL(setconv_kernel, dim3(grid), dim3(FUSED_BLOCK), tile_lds_bytes(S), (hipStream_t)stream, *a, S);
    return check_launch(who);
}

extern "C" int elo_mlp_fused(const elo_mlp_args *a, elo_stream_t stream)
{
    const char *who = "elo_mlp_fused";
    ELO_REQUIRE(a, who, "null argument block");
    ELO_REQUIRE(a->rows >= 0 && a->n_sources >= 1 && a->n_sources <= 3, who, "bad sizes");
    ELO_REQUIRE(a->n_layers >= 1 && a->n_layers <= ELO_MAX_CHAIN && a->out, who, "1..3 layers and an output");
    int width = 0;
    for (int s = 0; s < a->n_sources; ++s) {
        ELO_REQUIRE(a->src[s] && a->src_width[s] > 0, who, "null / empty source");
        width += a->src_width[s];
    }
    int cols = (width + 7) & ~7;
    for (int l = 0; l < a->n_layers; ++l) {
        if (int rc = check_dense(a->layers[l], width, 0, who, "mlp")) return rc;
        width = a->layers[l].N;
        cols = cols > ((width + 31) & ~31) ? cols : ((width + 31) & ~31);
    }
    if (a->rows == 0) return ELO_OK;
    const int S = row_stride(cols);
    const unsigned grid = (unsigned)((a->rows + TILE - 1) / TILE);
    hipLaunchKernelGGL(mlp_kernel, dim3(grid), dim3(FUSED_BLOCK), tile_lds_bytes(S), (hipStream_t)stream, *a, S);
    return check_launch(who);
}

extern "C" int elo_cv_stage1_fused(const elo_cv1_args *a, elo_stream_t stream)
{
    const char *who = "elo_cv_stage1_fused";
    ELO_REQUIRE(a, who, "null argument block");
    ELO_REQUIRE(a->batch >= 0 && a->npoints > 0 && a->K > 0 && a->H2 > 0 && a->W2 > 0 && a->C > 0, who, "bad sizes");
    if (a->K > TILE) return fail(ELO_ERR_LIMIT, "%s: K = %d exceeds the %d-row tile", who, a->K, TILE);
    ELO_REQUIRE(a->xyz1 && a->feat1 && a->xyz2 && a->feat2 && a->idx && a->mask && a->out, who, "null tensor pointer");
    const int CT = 10 + 2 * a->C;
    if (int rc = check_dense(a->cv0, CT, 128, who, "CV_0")) return rc;
    if (int rc = check_dense(a->cv1, 128, 64, who, "CV_1")) return rc;
    if (int rc = check_dense(a->cv2, 64, 64, who, "CV_2")) return rc;
    if (int rc = check_dense(a->cv_xyz, 10, 64, who, "CV_xyz")) return rc;
    if (int rc = check_dense(a->sum_cv0, 128, 128, who, "sum_CV_0")) return rc;
    if (int rc = check_dense(a->sum_cv1, 128, 64, who, "sum_CV_1")) return rc;
    const long points = (long)a->batch * a->npoints;
    if (points == 0) return ELO_OK;
    const int CTp = (CT + 7) & ~7, P = TILE / a->K;
    const int S = row_stride(128 + (CTp > 128 ? CTp : 128));
    const unsigned grid = (unsigned)((points + P - 1) / P);
    hipLaunchKernelGGL(cv1_kernel, dim3(grid), dim3(FUSED_BLOCK), tile_lds_bytes(S), (hipStream_t)stream, *a, S);
    return check_launch(who);
}

extern "C" int elo_cv_stage2_fused(const elo_cv2_args *a, elo_stream_t stream)
{
    const char *who = "elo_cv_stage2_fused";
    ELO_REQUIRE(a, who, "null argument block");
    ELO_REQUIRE(a->batch >= 0 && a->npoints > 0 && a->K > 0 && a->H > 0 && a->W > 0 && a->C > 0, who, "bad sizes");
    ELO_REQUIRE(a->npoints == a->H * a->W, who, "npoints must equal H*W (every pixel is a centre)");
    if (a->K > TILE) return fail(ELO_ERR_LIMIT, "%s: K = %d exceeds the %d-row tile", who, a->K, TILE);
    if (a->C > 64) return fail(ELO_ERR_LIMIT, "%s: C = %d exceeds 64", who, a->C);
    ELO_REQUIRE(a->xyz1 && a->feat1 && a->cost && a->idx && a->mask && a->out, who, "null tensor pointer");
    if (int rc = check_dense(a->xyz_enc, 10, 64, who, "sum_xyz_encoding")) return rc;
    if (int rc = check_dense(a->sum_cost0, 128 + a->C, 128, who, "sum_cost_volume_0")) return rc;
    if (int rc = check_dense(a->sum_cost1, 128, 64, who, "sum_cost_volume_1")) return rc;
    const long points = (long)a->batch * a->npoints;
    if (points == 0) return ELO_OK;
    const int P = TILE / a->K, S = row_stride(208);
    const unsigned grid = (unsigned)((points + P - 1) / P);
    hipLaunchKernelGGL(cv2_kernel, dim3(grid), dim3(FUSED_BLOCK), tile_lds_bytes(S), (hipStream_t)stream, *a, S);
    return check_launch(who);
}
