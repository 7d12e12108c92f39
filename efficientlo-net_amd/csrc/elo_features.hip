// elo_features.hip -- fused gather / encode / pool kernels of the feature path
// (set-conv, attentive cost volume, set-upconv, softmax_valid, warp+re-projection)
// for MI355X (gfx950, wave64).  Interfaces and the reference lines each kernel
// covers: include/elo.h.
//
// All of these are HBM/L2-bound gathers and reductions (< 1 flop per byte): the
// design goal is one pass over the operator-boundary tensors with coalesced
// channel-contiguous accesses -- a wavefront walks one (b,n[,k]) row, lanes run
// along the channel axis, so every gathered feature row is read as one
// contiguous segment and every output row is written as one.  The dense 1x1
// convolutions between them are GEMMs and stay with hipBLASLt.
#include <type_traits>
#include "elo_common.h"
#include "elo_project_device.h"

namespace elo {
namespace {

constexpr int ROWS_PER_BLOCK = ELO_BLOCK / ELO_WAVE;   // one wave per row

__device__ __forceinline__ long row_of_wave(long total_rows)
{
    const long r = (long)xcd_tile(blockIdx.x, gridDim.x) * ROWS_PER_BLOCK + threadIdx.x / ELO_WAVE;
    return r < total_rows ? r : -1;
}

inline unsigned grid_for_rows(long rows) { return (unsigned)((rows + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK); }

// ------------------------------------------------------------ staged row writers
// The gather/encode kernels emit rows of 3+C / 10+2C / 10 / C+Cc floats (19 ... 138): written row by row as 4-byte
// lane stores these are partial, unaligned cache-line writes (measured: 10x WRITE_SIZE amplification, 463 GB/s,
// profiles/r01_pmc).  So a workgroup builds 64 rows in LDS (wave per row, lanes along channels: gathered feature
// rows are still read as contiguous segments) and then streams the tile out linearly with 16-byte stores:
// 64 rows x CT floats is always a multiple of 256 bytes, so every tile starts on a 256-byte boundary.
constexpr int STAGE_ROWS = 64;

__device__ __forceinline__ void stream_out(const float *tile, float *__restrict__ dst, long first_row, long rows, int CT)
{
    const long valid = (rows - first_row < STAGE_ROWS ? rows - first_row : STAGE_ROWS) * CT;   // floats in this tile
    float *o = dst + first_row * CT;
    const long vec = valid >> 2;
    for (long i = threadIdx.x; i < vec; i += ELO_BLOCK)
        reinterpret_cast<float4 *>(o)[i] = reinterpret_cast<const float4 *>(tile)[i];
    for (long i = (vec << 2) + threadIdx.x; i < valid; i += ELO_BLOCK) o[i] = tile[i];
}

inline unsigned grid_for_stage(long rows) { return (unsigned)((rows + STAGE_ROWS - 1) / STAGE_ROWS); }

// per-row gather metadata of a 64-row tile, fetched once by the first 64 threads
struct StageMeta { long *cell; float *mask; };

__device__ __forceinline__ StageMeta stage_meta(float *after_tile, long first, long rows, const int *__restrict__ idx,
                                                const float *__restrict__ mask, int H2, int W2)
{
    StageMeta m{reinterpret_cast<long *>(after_tile), after_tile + 2 * STAGE_ROWS};
    const int t = threadIdx.x;
    if (t < STAGE_ROWS) {
        const long r = first + t;
        long cell = -1;
        float mk = 0.0f;
        if (r < rows) {
            const int *id = idx + r * 3;
            cell = ((long)id[0] * H2 + id[1]) * W2 + id[2];
            mk = mask[r];
        }
        m.cell[t] = cell;
        m.mask[t] = mk;
    }
    __syncthreads();
    return m;
}

constexpr size_t STAGE_META_BYTES = sizeof(float) * 3 * STAGE_ROWS;

// ------------------------------------------------------------ group_concat
__global__ __launch_bounds__(ELO_BLOCK) void group_concat_kernel(const elo_group_concat_args a, const long rows)
{
    extern __shared__ float tile[];
    const long first = (long)xcd_tile(blockIdx.x, gridDim.x) * STAGE_ROWS;
    if (first >= rows) return;
    const int CT = 3 + a.C;
    const StageMeta m = stage_meta(tile + STAGE_ROWS * CT + (STAGE_ROWS * CT & 1), first, rows, a.idx, a.mask, a.H2, a.W2);
    for (int e = threadIdx.x; e < STAGE_ROWS * CT; e += ELO_BLOCK) {     // independent iterations: loads overlap
        const int lr = e / CT, ch = e - lr * CT;
        const long cell = m.cell[lr];
        if (cell < 0) continue;
        const float mk = m.mask[lr];
        const long bn = point_batch(first + lr, a.K);
        tile[e] = ch < 3 ? a.src_xyz[cell * 3 + ch] * mk - a.centre_xyz[bn * 3 + ch] : a.src_feat[cell * a.C + (ch - 3)] * mk;
    }
    __syncthreads();
    stream_out(tile, a.out, first, rows, CT);
}

// ------------------------------------------------------------ masked max-pool
__global__ __launch_bounds__(ELO_BLOCK) void masked_maxpool_kernel(const elo_masked_maxpool_args a, const long rows)
{
    const long r = row_of_wave(rows);                  // r = b*N + n
    if (r < 0) return;
    const int lane = threadIdx.x % ELO_WAVE;
    const float *x = a.x + r * a.K * a.C;
    const float *m = a.mask + r * a.K;
    for (int c = lane; c < a.C; c += ELO_WAVE) {
        float best = x[c] * m[0];
        for (int k = 1; k < a.K; ++k) best = fmaxf(best, x[(long)k * a.C + c] * m[k]);
        a.out[r * a.C + c] = best;
    }
}

// The same, a THREAD per (point, four consecutive channels): K 16-byte loads, eight in flight, no idle lanes (the wave-per-point form
// above walks K dependent 4-byte loads with C of its 64 lanes -- 16 of 64 at the l0 set-conv's C = 16).  Same products, same order of
// the fmaxf chain: same bits.
__global__ __launch_bounds__(ELO_BLOCK) void masked_maxpool_vec_kernel(const elo_masked_maxpool_args a, const long items)
{
    const long e = (long)blockIdx.x * ELO_BLOCK + threadIdx.x;
    if (e >= items) return;
    const int q = a.C >> 2;
    const long pt = point_batch(e, q);
    const int cq = (int)(e - pt * q);
    const float4 *x = reinterpret_cast<const float4 *>(a.x + pt * a.K * a.C) + cq;
    const float *m = a.mask + pt * a.K;
    float4 best{0.f, 0.f, 0.f, 0.f};
    constexpr int U = 8;
    for (int k0 = 0; k0 < a.K; k0 += U) {
        float4 v[U];
        float w[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { const int k = k0 + u < a.K ? k0 + u : a.K - 1; v[u] = x[(long)k * q]; w[u] = m[k]; }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (k0 + u < a.K) {
                const float4 p{v[u].x * w[u], v[u].y * w[u], v[u].z * w[u], v[u].w * w[u]};
                if (k0 + u == 0) best = p;
                else best = float4{fmaxf(best.x, p.x), fmaxf(best.y, p.y), fmaxf(best.z, p.z), fmaxf(best.w, p.w)};
            }
        }
    }
    reinterpret_cast<float4 *>(a.out + pt * a.C)[cq] = best;
}

// ------------------------------------------------------------ cost volume: encode
struct Geo { float p[3], g[3], d[3], euc; };

__device__ __forceinline__ Geo geometry(const float *p, const float *g, float m)
{
    Geo s;
#pragma unroll
    for (int i = 0; i < 3; ++i) { s.p[i] = p[i]; s.g[i] = g[i] * m; s.d[i] = s.g[i] - s.p[i]; }
    s.euc = sqrtf(s.d[0] * s.d[0] + s.d[1] * s.d[1] + s.d[2] * s.d[2] + 1e-20f);
    return s;
}

__device__ __forceinline__ float geo_channel(const Geo &s, int ch)
{
    return ch < 3 ? s.p[ch] : ch < 6 ? s.g[ch - 3] : ch < 9 ? s.d[ch - 6] : s.euc;
}

__device__ __forceinline__ float geo_value(const float *p, const float *g, float m, int ch)
{   // [p, g*m, g*m - p, sqrt(sum((g*m-p)^2) + 1e-20)]   utils/pointnet_util.py:54-62
    float gm[3], d[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) { gm[i] = g[i] * m; d[i] = gm[i] - p[i]; }
    if (ch < 3) return p[ch];
    if (ch < 6) return gm[ch - 3];
    if (ch < 9) return d[ch - 6];
    return sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + 1e-20f);
}

__global__ __launch_bounds__(ELO_BLOCK) void cv_encode1_kernel(const elo_cv_encode1_args a, const long rows)
{
    extern __shared__ float tile[];
    const long first = (long)xcd_tile(blockIdx.x, gridDim.x) * STAGE_ROWS;
    if (first >= rows) return;
    const int C = a.C, CT = 10 + 2 * C;
    const StageMeta m = stage_meta(tile + STAGE_ROWS * CT + (STAGE_ROWS * CT & 1), first, rows, a.idx, a.mask, a.H2, a.W2);
    for (int e = threadIdx.x; e < STAGE_ROWS * CT; e += ELO_BLOCK) {
        const int lr = e / CT, ch = e - lr * CT;
        const long cell = m.cell[lr];
        if (cell < 0) continue;
        const float mk = m.mask[lr];
        const long bn = point_batch(first + lr, a.K);
        tile[e] = ch < 10 ? geo_value(a.xyz1 + bn * 3, a.xyz2 + cell * 3, mk, ch)
                : ch < 10 + C ? static_cast<const float *>(a.feat1)[bn * C + (ch - 10)]
                              : static_cast<const float *>(a.feat2)[cell * C + (ch - 10 - C)] * mk;
    }
    __syncthreads();
    stream_out(tile, static_cast<float *>(a.out), first, rows, CT);
}

__global__ __launch_bounds__(ELO_BLOCK) void cv_encode2_kernel(const elo_cv_encode2_args a, const long rows)
{
    extern __shared__ float tile[];                     // [64][10] geometry, then [64][C+Cc]
    const long first = (long)xcd_tile(blockIdx.x, gridDim.x) * STAGE_ROWS;
    if (first >= rows) return;
    const int CT = a.C + a.Cc, ALL = 10 + CT;
    float *geo = tile, *rest = tile + STAGE_ROWS * 10;
    const StageMeta m = stage_meta(tile + STAGE_ROWS * ALL + (STAGE_ROWS * ALL & 1), first, rows, a.idx, a.mask, a.H, a.W);
    for (int e = threadIdx.x; e < STAGE_ROWS * ALL; e += ELO_BLOCK) {
        const int lr = e / ALL, ch = e - lr * ALL;
        const long cell = m.cell[lr];
        if (cell < 0) continue;
        const float mk = m.mask[lr];
        const long bn = point_batch(first + lr, a.K);
        if (ch < 10) geo[lr * 10 + ch] = geo_value(a.xyz1 + bn * 3, a.xyz1 + cell * 3, mk, ch);
        else if (ch < 10 + a.C) rest[lr * CT + (ch - 10)] = static_cast<const float *>(a.feat1)[bn * a.C + (ch - 10)];
        else rest[lr * CT + (ch - 10)] = static_cast<const float *>(a.cost)[cell * a.Cc + (ch - 10 - a.C)] * mk;
    }
    __syncthreads();
    stream_out(geo, static_cast<float *>(a.xyz_cat), first, rows, 10);
    stream_out(rest, static_cast<float *>(a.rest), first, rows, CT);
}

// ------------------------------------------------------------ cost volume: encode, vector form
// The output is addressed FLAT in 8-byte slots (a row of 10 + 2C floats is 5 + C slots and starts 8-byte aligned):
// thread t of a workgroup fills slots t, t+256, ... of its 64- or 128-row span, so every store instruction of a wave writes
// 512 contiguous bytes and every feature load is an 8-byte piece of a contiguous row segment.  Only the per-row
// facts go through LDS: one thread per row resolves the row's neighbour cell, mask and the 10 geometry floats
// once (4 loads per ROW instead of per slot), everything else is one global load + one global store per slot.
// Row / channel of a slot come from a multiply-high by ceil(2^32/d) (exact for n, d < 65536), not a division.
struct FastDiv {
    unsigned magic;                                  // 0: divisor 1
    __device__ __forceinline__ unsigned operator()(unsigned n) const { return magic ? __umulhi(n, magic) : n; }
};

inline FastDiv fast_div(unsigned d) { return FastDiv{d <= 1 ? 0u : (unsigned)((0x100000000ull + d - 1) / d)}; }

// rows per workgroup: 64 keeps small launches spread over the CUs, 128 amortises the row-facts phase when there are
// several rounds of workgroups per CU anyway (measured at l0: B=8 10.6 vs 11.4 us, B=64 83 vs 71 us)
// rows per workgroup of the encode kernels: small calls take small workgroups (a 24 576-row l2 call is 384 workgroups of 64 rows on 256 CUs:
// 8.0 us; 768 of 32: 6.2 us -- round 6)
inline int enc_rows(long rows) { return rows > 512 * 1024 ? 128 : rows >= 64 * 1024 ? 64 : 32; }

template <int ENC_ROWS>
struct RowFacts {
    long cell[ENC_ROWS];
    float mask[ENC_ROWS];
    float geo[ENC_ROWS][10];                       // [p, g*m, g*m - p, |g*m - p|]   utils/pointnet_util.py:54-62
};

// xyz_c: the centres' cloud (B*N,3); xyz_n: the neighbours' grid (B*Hn*Wn,3)
template <int ENC_ROWS>
__device__ __forceinline__ void resolve_rows(RowFacts<ENC_ROWS> &rf, long first, int nrows, long bn0, unsigned rem0, FastDiv by_K,
                                             const int *__restrict__ idx, const float *__restrict__ mask,
                                             const float *__restrict__ xyz_c, const float *__restrict__ xyz_n, int Hn, int Wn)
{
    const int t = threadIdx.x;
    if (t < nrows) {
        const long row = first + t, bn = bn0 + by_K(rem0 + t);
        const int *id = idx + row * 3;
        const long cell = ((long)id[0] * Hn + id[1]) * Wn + id[2];
        const float m = mask[row];
        const float *p = xyz_c + bn * 3, *g = xyz_n + cell * 3;
        const float p0 = p[0], p1 = p[1], p2 = p[2];
        const float g0 = g[0] * m, g1 = g[1] * m, g2 = g[2] * m;
        const float d0 = g0 - p0, d1 = g1 - p1, d2 = g2 - p2;
        rf.cell[t] = cell;
        rf.mask[t] = m;
        float *o = rf.geo[t];
        o[0] = p0; o[1] = p1; o[2] = p2; o[3] = g0; o[4] = g1; o[5] = g2; o[6] = d0; o[7] = d1; o[8] = d2;
        o[9] = sqrtf(d0 * d0 + d1 * d1 + d2 * d2 + 1e-20f);
    }
    __syncthreads();
}

// Outputs of the encode / pool kernels are written once and read by a later launch: non-temporal stores keep them from
// allocating in L2 on the way out (measured at l0: A1 9.5 -> 8.4 us at B=8 and 70 -> 53 us at B=64, A2 99 -> 61 us at
// B=64).  Non-temporal LOADS of the pool's inputs were mixed on Infinity-Cache-resident tensors (P1 B=8 15.8 -> 17.1 us, P2 11.9 -> 10.4:
// round 2); round 6 measured them HBM-COLD, where they pay, and uses them (softmax_pool_vec_kernel<.., NT>, softmax_pool_wave_kernel).
#define STREAM_STORE(v, p) __builtin_nontemporal_store(v, p)

constexpr int ENC_UNROLL = 3;                        // slots in flight per thread

// ---- storage types of the feature tensors: fp32, or fp16 storage with fp32 arithmetic (elo.h: ELO_F16) -----------
using half_t = _Float16;

template <class T> struct Store;                     // V2: two elements (a geometry / feature pair), V16: 16 bytes
template <> struct Store<float> {
    typedef float V2 __attribute__((ext_vector_type(2)));
    typedef float V4 __attribute__((ext_vector_type(4)));
    typedef V4 V16;
    static constexpr int PER16 = 4;
};
template <> struct Store<half_t> {
    typedef half_t V2 __attribute__((ext_vector_type(2)));
    typedef half_t V4 __attribute__((ext_vector_type(4)));
    typedef half_t V16 __attribute__((ext_vector_type(8)));
    static constexpr int PER16 = 8;
};

template <int ENC_ROWS, class T>
__global__ __launch_bounds__(ELO_BLOCK) void cv_encode1_vec_kernel(const elo_cv_encode1_args a, const long rows,
                                                                   const FastDiv by_slots, const FastDiv by_K)
{
    typedef typename Store<T>::V2 V2;
    __shared__ RowFacts<ENC_ROWS> rf;
    const long first = (long)xcd_tile(blockIdx.x, gridDim.x) * ENC_ROWS;
    if (first >= rows) return;
    const int C = a.C, HC = C >> 1, HP = 5 + C;      // two-element slots per row: 5 geometry, C/2 + C/2 features
    const int nrows = (int)(rows - first < ENC_ROWS ? rows - first : ENC_ROWS), nslots = nrows * HP;
    const long bn0 = point_batch(first, a.K);
    const unsigned rem0 = (unsigned)(first - bn0 * a.K);
    resolve_rows(rf, first, nrows, bn0, rem0, by_K, a.idx, a.mask, a.xyz1, a.xyz2, a.H2, a.W2);
    const T *feat1 = static_cast<const T *>(a.feat1), *feat2 = static_cast<const T *>(a.feat2);
    V2 *__restrict__ out = reinterpret_cast<V2 *>(static_cast<T *>(a.out) + first * (2 * HP));
    for (int s0 = threadIdx.x; s0 < nslots; s0 += ELO_BLOCK * ENC_UNROLL) {
        V2 f[ENC_UNROLL];
        int j[ENC_UNROLL];
        unsigned lr[ENC_UNROLL];
#pragma unroll
        for (int u = 0; u < ENC_UNROLL; ++u) {
            const int s = s0 + u * ELO_BLOCK < nslots ? s0 + u * ELO_BLOCK : nslots - 1;    // clamped: loads stay unconditional
            lr[u] = by_slots((unsigned)s);
            j[u] = s - (int)lr[u] * HP;
            const long bn = bn0 + by_K(rem0 + lr[u]);
            const T *src = j[u] < 5 + HC ? feat1 + bn * C + 2 * (j[u] < 5 ? 0 : j[u] - 5)             // geometry lanes: unused
                                         : feat2 + rf.cell[lr[u]] * C + 2 * (j[u] - 5 - HC);
            f[u] = *reinterpret_cast<const V2 *>(src);
        }
#pragma unroll
        for (int u = 0; u < ENC_UNROLL; ++u) {
            const float sc = j[u] < 5 + HC ? 1.0f : rf.mask[lr[u]];
            const float2 geo = *reinterpret_cast<const float2 *>(&rf.geo[lr[u]][j[u] < 5 ? 2 * j[u] : 0]);
            const float lo = j[u] < 5 ? geo.x : (float)f[u].x * sc, hi = j[u] < 5 ? geo.y : (float)f[u].y * sc;
            if (s0 + u * ELO_BLOCK < nslots) STREAM_STORE((V2{(T)lo, (T)hi}), &out[s0 + u * ELO_BLOCK]);
        }
    }
}

// Column-owner form of stage 1, for row lengths that tile the workgroup: RPI = 256 / (5 + C) whole rows fit the 256
// threads with few lanes over (C = 16: 12 rows x 21 slots = 252).  Thread t keeps ONE slot column j = t % (5 + C) for
// its whole life and walks rows t / (5 + C), + RPI, ...: what a slot is (geometry / own feature / neighbour feature),
// its source tensor and its channel offset are loop invariants, the per-slot work is one row-fact read, one 8-byte
// load and one 8-byte store, and the wave's stores stay contiguous (slot index = row * (5 + C) + j = t + RPI*(5+C)*i).
// A workgroup spans RPI * ENC_BATCH * rounds rows (a whole number of ENC_BATCH-deep load batches: no clamped tail
// inside a span).
constexpr int ENC_BATCH = 5;

template <int ENC_ROWS, class T>
__global__ __launch_bounds__(ELO_BLOCK) void cv_encode1_col_kernel(const elo_cv_encode1_args a, const long rows,
                                                                   const FastDiv by_slots, const FastDiv by_K,
                                                                   const int rpi, const int span)
{
    typedef typename Store<T>::V2 V2;
    __shared__ RowFacts<ENC_ROWS> rf;
    const long first = (long)xcd_tile(blockIdx.x, gridDim.x) * span;
    if (first >= rows) return;
    const int C = a.C, HC = C >> 1, HP = 5 + C;
    const int nrows = (int)(rows - first < span ? rows - first : span);
    const long bn0 = point_batch(first, a.K);
    const unsigned rem0 = (unsigned)(first - bn0 * a.K);
    resolve_rows(rf, first, nrows, bn0, rem0, by_K, a.idx, a.mask, a.xyz1, a.xyz2, a.H2, a.W2);
    const int r0 = (int)by_slots(threadIdx.x), j = (int)threadIdx.x - r0 * HP;
    if (r0 >= rpi) return;                           // the 256 % (5 + C) lanes over
    const bool geo = j < 5, own = j < 5 + HC;
    const T *src = own ? static_cast<const T *>(a.feat1) + (geo ? 0 : 2 * (j - 5))
                       : static_cast<const T *>(a.feat2) + 2 * (j - 5 - HC);
    const int g2 = geo ? 2 * j : 0;
    V2 *__restrict__ out = reinterpret_cast<V2 *>(static_cast<T *>(a.out) + first * (2 * HP)) + threadIdx.x;
    const int step = rpi * HP;
    for (int lr0 = r0; lr0 < nrows; lr0 += rpi * ENC_BATCH, out += step * ENC_BATCH) {
        V2 f[ENC_BATCH];
        int lr[ENC_BATCH];
#pragma unroll
        for (int u = 0; u < ENC_BATCH; ++u) {
            lr[u] = lr0 + u * rpi < nrows ? lr0 + u * rpi : nrows - 1;          // clamped (last workgroup only)
            const long from = own ? bn0 + by_K(rem0 + lr[u]) : rf.cell[lr[u]];
            f[u] = *reinterpret_cast<const V2 *>(src + from * C);
        }
#pragma unroll
        for (int u = 0; u < ENC_BATCH; ++u) {
            const float sc = own ? 1.0f : rf.mask[lr[u]];
            const float2 gv = *reinterpret_cast<const float2 *>(&rf.geo[lr[u]][g2]);
            const float lo = geo ? gv.x : (float)f[u].x * sc, hi = geo ? gv.y : (float)f[u].y * sc;
            if (lr0 + u * rpi < nrows) STREAM_STORE((V2{(T)lo, (T)hi}), &out[u * step]);
        }
    }
}

// Staged form of stage 1 (round 6): the workgroup BUILDS its rows of the output in LDS, in the output's own byte layout, and then
// streams the tile out as 16-byte vectors.  The column-owner / pair forms above pay one 4- or 8-byte gather and one narrow store per
// two-element slot (8.3 M slots at the 128 x 2048 l0 shape, batch 8: ~13 us in either storage type -- bound by the slot count, not
// by the bytes).  Here a row's two feature rows are fetched as 16-byte chunks (C = 16 in fp16: two loads per source row instead of
// eight), scaled and laid into the tile next to the row's geometry code, and the tile -- contiguous in HBM -- leaves as whole
// 16-byte vectors: every global access is 16 bytes wide and coalesced.  Same values bit for bit (the same conversions in the same order).
//   rows per workgroup ENC_ROWS (a multiple of 4: a tile starts on a 16-byte boundary in both storage types); dynamic LDS: the tile.
template <int ENC_ROWS, class T>
__global__ __launch_bounds__(ELO_BLOCK) void cv_encode1_staged_kernel(const elo_cv_encode1_args a, const long rows, const FastDiv by_K)
{
    typedef typename Store<T>::V16 V16;
    constexpr int E = Store<T>::PER16;               // elements per 16 bytes
    extern __shared__ __align__(16) unsigned char tile_raw[];
    __shared__ RowFacts<ENC_ROWS> rf;
    T *tile = reinterpret_cast<T *>(tile_raw);
    const long first = (long)xcd_tile(blockIdx.x, gridDim.x) * ENC_ROWS;
    if (first >= rows) return;
    const int C = a.C, RW = 10 + 2 * C, CH = C / E;   // row width in elements, 16-byte chunks per feature row
    const int nrows = (int)(rows - first < ENC_ROWS ? rows - first : ENC_ROWS);
    const long bn0 = point_batch(first, a.K);
    const unsigned rem0 = (unsigned)(first - bn0 * a.K);
    resolve_rows(rf, first, nrows, bn0, rem0, by_K, a.idx, a.mask, a.xyz1, a.xyz2, a.H2, a.W2);
    const T *feat1 = static_cast<const T *>(a.feat1), *feat2 = static_cast<const T *>(a.feat2);
    // features: work item = (row, tensor, chunk); 2 * CH items per row, U in flight per thread
    constexpr int U = 4;
    const int per_row = 2 * CH, items = nrows * per_row;
    for (int i0 = threadIdx.x; i0 < items; i0 += ELO_BLOCK * U) {
        V16 f[U];
        int lr[U], ch[U];
        bool own[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = i0 + u * ELO_BLOCK < items ? i0 + u * ELO_BLOCK : items - 1;       // clamped: loads stay unconditional
            lr[u] = i / per_row;
            const int w = i - lr[u] * per_row;
            own[u] = w < CH;
            ch[u] = own[u] ? w : w - CH;
            const long from = own[u] ? bn0 + by_K(rem0 + lr[u]) : rf.cell[lr[u]];
            f[u] = *reinterpret_cast<const V16 *>((own[u] ? feat1 : feat2) + from * C + ch[u] * E);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (i0 + u * ELO_BLOCK >= items) continue;
            const float sc = own[u] ? 1.0f : rf.mask[lr[u]];
            T *dst = tile + lr[u] * RW + 10 + (own[u] ? 0 : C) + ch[u] * E;              // (4-byte aligned: RW and the offsets are even)
#pragma unroll
            for (int e = 0; e < E; e += 2) {
                typename Store<T>::V2 v{(T)((float)f[u][e] * sc), (T)((float)f[u][e + 1] * sc)};
                *reinterpret_cast<typename Store<T>::V2 *>(dst + e) = v;
            }
        }
    }
    for (int i = threadIdx.x; i < nrows * 5; i += ELO_BLOCK) {                           // geometry: five two-element slots per row
        const int r = i / 5, j = i - r * 5;
        const float2 g = *reinterpret_cast<const float2 *>(&rf.geo[r][2 * j]);
        typename Store<T>::V2 v{(T)g.x, (T)g.y};
        *reinterpret_cast<typename Store<T>::V2 *>(tile + r * RW + 2 * j) = v;
    }
    __syncthreads();
    // the tile leaves as it lies: nrows * RW elements from `first * RW` on
    const int total = nrows * RW, nvec = total / E;
    T *out = static_cast<T *>(a.out) + first * RW;
    for (int v = threadIdx.x; v < nvec; v += ELO_BLOCK)
        STREAM_STORE(reinterpret_cast<const V16 *>(tile)[v], reinterpret_cast<V16 *>(out) + v);
    for (int e = nvec * E + 2 * threadIdx.x; e < total; e += 2 * ELO_BLOCK)             // a ragged last tile: the bytes past the last whole vector
        *reinterpret_cast<typename Store<T>::V2 *>(out + e) = *reinterpret_cast<const typename Store<T>::V2 *>(tile + e);
}

#ifndef ELO_ENCODE1_STAGED
#define ELO_ENCODE1_STAGED 1                         // (0: an A/B build without the staged form)
#endif
// Stage 2 has two outputs: xyz_cat rows of 10 elements (two-element slots, straight from the row facts) and rest
// rows of C + Cc elements in 16-byte slots (4 floats / 8 halves: C and Cc multiples of that).
template <int ENC_ROWS, class T>
__global__ __launch_bounds__(ELO_BLOCK) void cv_encode2_vec_kernel(const elo_cv_encode2_args a, const long rows,
                                                                   const FastDiv by_slots, const FastDiv by_K)
{
    typedef typename Store<T>::V2 V2;
    typedef typename Store<T>::V16 V16;
    constexpr int E = Store<T>::PER16;
    __shared__ RowFacts<ENC_ROWS> rf;
    const long first = (long)xcd_tile(blockIdx.x, gridDim.x) * ENC_ROWS;
    if (first >= rows) return;
    const int C = a.C, Cc = a.Cc, QC = C / E, QP = (C + Cc) / E;
    const int nrows = (int)(rows - first < ENC_ROWS ? rows - first : ENC_ROWS), nslots = nrows * QP;
    const long bn0 = point_batch(first, a.K);
    const unsigned rem0 = (unsigned)(first - bn0 * a.K);
    resolve_rows(rf, first, nrows, bn0, rem0, by_K, a.idx, a.mask, a.xyz1, a.xyz1, a.H, a.W);
    const T *feat1 = static_cast<const T *>(a.feat1), *cost = static_cast<const T *>(a.cost);
    V16 *__restrict__ rest = reinterpret_cast<V16 *>(static_cast<T *>(a.rest) + first * (long)(C + Cc));
    for (int s0 = threadIdx.x; s0 < nslots; s0 += ELO_BLOCK * ENC_UNROLL) {
        V16 f[ENC_UNROLL];
        float sc[ENC_UNROLL];
#pragma unroll
        for (int u = 0; u < ENC_UNROLL; ++u) {
            const int s = s0 + u * ELO_BLOCK < nslots ? s0 + u * ELO_BLOCK : nslots - 1;
            const unsigned lr = by_slots((unsigned)s);
            const int j = s - (int)lr * QP;
            const long bn = bn0 + by_K(rem0 + lr);
            const T *src = j < QC ? feat1 + bn * C + E * j : cost + rf.cell[lr] * Cc + E * (j - QC);
            f[u] = *reinterpret_cast<const V16 *>(src);
            sc[u] = j < QC ? 1.0f : rf.mask[lr];
        }
#pragma unroll
        for (int u = 0; u < ENC_UNROLL; ++u) {
            V16 v;
#pragma unroll
            for (int e = 0; e < E; ++e) v[e] = (T)((float)f[u][e] * sc[u]);
            if (s0 + u * ELO_BLOCK < nslots) STREAM_STORE(v, &rest[s0 + u * ELO_BLOCK]);
        }
    }
    V2 *__restrict__ geo = reinterpret_cast<V2 *>(static_cast<T *>(a.xyz_cat) + first * 10);
    const float2 *staged = reinterpret_cast<const float2 *>(&rf.geo[0][0]);
    for (int s = threadIdx.x; s < nrows * 5; s += ELO_BLOCK) STREAM_STORE((V2{(T)staged[s].x, (T)staged[s].y}), &geo[s]);
}

// ------------------------------------------------------------ masked softmax-pool over K
// Vectorised form (C % 4 == 0, aligned rows): a quarter-wave (16 lanes x 4 channels = 64 channels) owns one (b,n)
// point, so a wave streams 4 points' K x C logits and values with 16-byte (fp32) / 8-byte (fp16) loads -- every load
// instruction of the wave covers four full rows -- in ONE pass (online softmax, fp32 arithmetic) and writes 4 channels
// per lane.
// NT (fp16 storage since round 6): the inputs are read once -- nontemporal loads, measured HBM-cold at the 128 x 2048 l0 shape, batch 8:
// P1 23.9 -> 22.3 us, P2 17.6 -> 15.6 (0.58 -> 0.62, 0.54 -> 0.61 of 8 TB/s); on ONE Infinity-Cache-resident set 19.2 -> 20.2 / 14.3 -> 14.5
// (gpurun_out/r06/cold_sweep_f16_e*.txt).  fp32 storage takes the wave-per-point form below.  Also measured for fp16 and NOT taken: 16-byte
// loads with an eighth-wave per point (half the load instructions, twice the work per lane): P1 22.7 -> 26.3 us, P2 16.3 -> 18.5
// (gpurun_out/r06/cold_levels_f16_wide.txt) -- like tools/micro/hbm_probe.hip's one-shot rows, the lighter thread wins.
template <class T, int U, bool NT = false>          // U neighbour rows in flight per lane (6 when K is a multiple of 6, else 4); NT: nontemporal loads
__global__ __launch_bounds__(ELO_BLOCK) void softmax_pool_vec_kernel(const elo_softmax_pool_args a, const long rows)
{
    typedef typename Store<T>::V4 V4;
    const int lanes_per_row = a.C >> 2;                              // 16 for C = 64
    const int rows_per_block = ELO_BLOCK / lanes_per_row;
    const int sub = threadIdx.x % lanes_per_row;
    const long r = (long)xcd_tile(blockIdx.x, gridDim.x) * rows_per_block + threadIdx.x / lanes_per_row;
    if (r >= rows) return;
    const V4 *lg = reinterpret_cast<const V4 *>(static_cast<const T *>(a.logits) + r * a.K * a.C) + sub;
    const V4 *vl = reinterpret_cast<const V4 *>(static_cast<const T *>(a.values) + r * a.K * (long)a.values_stride) + sub;
    const int lstep = a.C >> 2, vstep = a.values_stride >> 2;
    const float *m = a.mask + r * a.K;
    float mx[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY}, den[4] = {0, 0, 0, 0}, acc[4] = {0, 0, 0, 0};
    // e^x on the hardware exp2 (v_exp_f32, ~1 ulp of fp32): a fifth of the instructions of expf's range-reduced
    // polynomial, and the exponentials are what bounds this kernel (the softmax weights feed a convex combination: the
    // result moves by ~1e-7 relative, far inside the 1e-4 parity tolerance).
    auto ex = [](float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896f); };
    for (int k0 = 0; k0 < a.K; k0 += U) {
        V4 l4[U], v4[U];
        float mk[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int k = k0 + u < a.K ? k0 + u : a.K - 1;
            l4[u] = NT ? __builtin_nontemporal_load(lg + (long)k * lstep) : lg[(long)k * lstep];
            v4[u] = NT ? __builtin_nontemporal_load(vl + (long)k * vstep) : vl[(long)k * vstep];
            mk[u] = m[k];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (k0 + u >= a.K) break;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float l = mk[u] == 1.0f ? (float)l4[u][c] : -1e10f, vv = (float)v4[u][c];
                if (l > mx[c]) {
                    const float sc = ex(mx[c] - l);
                    den[c] = den[c] * sc + 1.0f;
                    acc[c] = acc[c] * sc + vv;
                    mx[c] = l;
                } else {
                    const float e = ex(l - mx[c]);
                    den[c] += e;
                    acc[c] += e * vv;
                }
            }
        }
    }
    STREAM_STORE((V4{(T)(acc[0] / den[0]), (T)(acc[1] / den[1]), (T)(acc[2] / den[2]), (T)(acc[3] / den[3])}),
                 &reinterpret_cast<V4 *>(static_cast<T *>(a.out) + r * a.C)[sub]);
}

// Wave-per-point form for C = 64 (every pooled width of the model), round 6: the quarter-wave form above keeps 2 x U loads in flight
// PER LANE and issues them in one burst per workgroup -- the access shape tools/micro/hbm_probe.hip measures at 0.54 of 8 TB/s from
// HBM-cold data (one-shot, 8 loads per thread), where one light load per thread reaches 0.81.  Here every lane moves 16 bytes per
// load and a point's K neighbour rows are spread over FOUR lane groups: a 64-channel row is LPR = 64 / E lanes wide (E = 4 floats or 8
// halves per 16 bytes), lane group g owns the neighbours k = g, g + 4, g + 8, ... (K = 6: two loads per tensor and lane, K = 4: one),
// a point takes 4 * LPR lanes -- the whole wave in fp32, half a wave in fp16 storage.  LAUNCHED FOR FP32 ONLY: in fp16 a point is 768
// bytes per tensor and the form is instruction-bound (a wave per point 52 us, two points per wave 31.9 us, the quarter-wave form 23.5).  Four times as many, that much lighter waves.  Two-pass softmax
// exactly as written in the reference (max over K, exponentials against it, sum): the per-lane maxima and sums meet through two
// butterfly steps across the lane groups (bfly: DPP / v_permlane*_swap, no LDS).  Loads are nontemporal (every byte is read once).
// Same arithmetic contract as the quarter-wave form (fp32 arithmetic, hardware exp2).  Measured HBM-cold at the 128 x 2048 l0 shape,
// batch 8, fp32: P1 42.4 -> 38.0 us (0.65 -> 0.72 of 8 TB/s), P2 31.6 -> 27.5 (0.60 -> 0.69); tools/cold_sweep.py.
// The value of lane (l ^ D) for D = 8, 16, 32 WITHOUT the LDS crossbar: D = 8 is a DPP rotation inside the 16-lane row, D = 16 / 32
// are gfx950's v_permlane16_swap / v_permlane32_swap (swapping a register's odd rows / upper half with a copy's even rows / lower
// half leaves [r0 r0 r2 r2] + [r1 r1 r3 r3], resp. [lo lo] + [hi hi]: both partners in every lane).  __shfl_xor is a ds_bpermute.
template <int D, class Op>
__device__ __forceinline__ float bfly(float v, Op op)
{
    const unsigned u = __float_as_uint(v);
    if constexpr (D == 8) {
        return op(v, __uint_as_float((unsigned)__builtin_amdgcn_update_dpp((int)u, (int)u, 0x128 /* row_ror:8 */, 0xf, 0xf, false)));
    } else if constexpr (D == 16) {
        const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
        return op(__uint_as_float(r[0]), __uint_as_float(r[1]));
    } else {
        static_assert(D == 32, "butterfly distance");
        const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
        return op(__uint_as_float(r[0]), __uint_as_float(r[1]));
    }
}
template <int LPR, class Op>                        // all-reduce over a point's FOUR lane groups (LPR lanes wide each): every lane ends with the total
__device__ __forceinline__ float across_groups(float v, Op op)
{
    v = bfly<LPR>(v, op);
    return bfly<2 * LPR>(v, op);
}

template <class T, int J>                           // J = ceil(K / 4) neighbour rows per lane group
__global__ __launch_bounds__(ELO_BLOCK) void softmax_pool_wave_kernel(const elo_softmax_pool_args a, const long rows)
{
    typedef typename Store<T>::V16 V16;
    constexpr int E = Store<T>::PER16, LPR = 64 / E;  // elements per lane, lanes per 64-channel row (= a lane group's width)
    constexpr int LPP = 4 * LPR, PPW = 64 / LPP;      // lanes per point, points per wave (1 in fp32, 2 in fp16)
    const int lane = threadIdx.x & 63, g = (lane % LPP) / LPR, sub = lane % LPR;
    long r = ((long)xcd_tile(blockIdx.x, gridDim.x) * (ELO_BLOCK / 64) + (threadIdx.x >> 6)) * PPW + lane / LPP;
    const bool mine = r < rows;                       // (fp16: the second half-wave of the last wave may have no point; it follows the
    if (!mine) r = rows - 1;                          //  first one through the butterflies on a clamped row and stores nothing)
    const V16 *lg = reinterpret_cast<const V16 *>(static_cast<const T *>(a.logits) + r * a.K * 64) + sub;
    const V16 *vl = reinterpret_cast<const V16 *>(static_cast<const T *>(a.values) + r * a.K * (long)a.values_stride) + sub;
    const int vstep = a.values_stride / E;
    const float *m = a.mask + r * a.K;
    V16 l4[J], v4[J];
    float mk[J];
    bool live[J];
#pragma unroll
    for (int j = 0; j < J; ++j) {
        const int k = g + 4 * j;
        live[j] = k < a.K;
        const int kc = live[j] ? k : 0;
        l4[j] = __builtin_nontemporal_load(lg + (long)kc * LPR);
        v4[j] = __builtin_nontemporal_load(vl + (long)kc * vstep);
        mk[j] = m[kc];
    }
    auto ex = [](float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896f); };
    float lv[J][E], mx[E];
#pragma unroll
    for (int c = 0; c < E; ++c) mx[c] = -INFINITY;
#pragma unroll
    for (int j = 0; j < J; ++j)
#pragma unroll
        for (int c = 0; c < E; ++c) {
            lv[j][c] = !live[j] ? -INFINITY : mk[j] == 1.0f ? (float)l4[j][c] : -1e10f;
            mx[c] = fmaxf(mx[c], lv[j][c]);
        }
    auto fmax2 = [](float x, float y) { return fmaxf(x, y); };
    auto add2 = [](float x, float y) { return x + y; };
#pragma unroll
    for (int c = 0; c < E; ++c) mx[c] = across_groups<LPR>(mx[c], fmax2);
    float den[E], acc[E];
#pragma unroll
    for (int c = 0; c < E; ++c) den[c] = acc[c] = 0.0f;
#pragma unroll
    for (int j = 0; j < J; ++j)
#pragma unroll
        for (int c = 0; c < E; ++c) {
            const float e = live[j] ? ex(lv[j][c] - mx[c]) : 0.0f;
            den[c] += e;
            acc[c] += e * (float)v4[j][c];
        }
#pragma unroll
    for (int c = 0; c < E; ++c) {
        den[c] = across_groups<LPR>(den[c], add2);
        acc[c] = across_groups<LPR>(acc[c], add2);
    }
    if (g == 0 && mine) {
        V16 o;
#pragma unroll
        for (int c = 0; c < E; ++c) o[c] = (T)(acc[c] / den[c]);
        STREAM_STORE(o, &reinterpret_cast<V16 *>(static_cast<T *>(a.out) + r * 64)[sub]);
    }
}

// scalar form for any C / alignment
__global__ __launch_bounds__(ELO_BLOCK) void softmax_pool_kernel(const elo_softmax_pool_args a, const long rows)
{
    const long r = row_of_wave(rows);                  // r = b*N + n
    if (r < 0) return;
    const int lane = threadIdx.x % ELO_WAVE;
    const float *lg = static_cast<const float *>(a.logits) + r * a.K * a.C;
    const float *vl = static_cast<const float *>(a.values) + r * a.K * (long)a.values_stride;
    const float *m = a.mask + r * a.K;
    float *outp = static_cast<float *>(a.out);
    for (int c = lane; c < a.C; c += ELO_WAVE) {
        float mx = -INFINITY;
        for (int k = 0; k < a.K; ++k) {
            const float l = m[k] == 1.0f ? lg[(long)k * a.C + c] : -1e10f;
            mx = fmaxf(mx, l);
        }
        float den = 0.0f, acc = 0.0f;
        for (int k = 0; k < a.K; ++k) {
            const float l = m[k] == 1.0f ? lg[(long)k * a.C + c] : -1e10f;
            const float e = expf(l - mx);
            den += e;
            acc += e * vl[(long)k * a.values_stride + c];
        }
        outp[r * a.C + c] = acc / den;
    }
}

// ------------------------------------------------------------ softmax_valid over the point axis
// Stage 1, grid (parts, batch, ceil(C/64)): each block reduces a slice of the points with an online
// softmax (4 waves stride over the slice, lanes over channels) and writes one (max, den, acc) triple
// per channel.  Stage 2 merges the `parts` triples (softmax_valid_merge / pose_head_kernel).
struct SvPartials {
    float *mx, *den, *acc;        // each (batch, parts, C)
};

__device__ __forceinline__ SvPartials sv_partials(float *scratch, int batch, int C)
{
    const size_t n = (size_t)batch * ELO_SV_MAX_PARTS * C;
    return SvPartials{scratch, scratch + n, scratch + 2 * n};
}

// F16 is a template parameter: with the storage type behind a run-time flag every load sat under a (uniform) branch, and
// a conditional load costs a full s_waitcnt vmcnt(0) -- the 16 rows a wave has "in flight" became 32 dependent round
// trips (14 us per launch at batch 8 with fp16 features, whatever the grid size: profiles/r02_c3_summary.json).
template <bool F16>
__global__ __launch_bounds__(ELO_BLOCK) void softmax_valid_partial_kernel(const void *__restrict__ feature_,
                                                                          const void *__restrict__ weight_,
                                                                          const float *__restrict__ xyz, int npoints,
                                                                          int C, int parts, float *scratch,
                                                                          const ProjectionClear clear)
{
    typedef typename std::conditional<F16, _Float16, float>::type feat_t;
    const feat_t *__restrict__ feature = reinterpret_cast<const feat_t *>(feature_);
    const feat_t *__restrict__ weight = reinterpret_cast<const feat_t *>(weight_);
    auto ex = [](float x) { return exp_acc(x); };      // v_exp_f32 with the argument's rounding corrected (elo_common.h)
    __shared__ float part[3][ROWS_PER_BLOCK][ELO_WAVE];
    clear_projection(clear);
    const int slice = blockIdx.x, b = blockIdx.y, lane = threadIdx.x % ELO_WAVE, wave = threadIdx.x / ELO_WAVE;
    const int c = blockIdx.z * ELO_WAVE + lane;
    const bool live = c < C;
    const int cc = live ? c : C - 1;                  // clamped: dead lanes load a real channel and drop it
    const long fbase = (long)b * npoints * C;
    const float *p = xyz + (long)b * npoints * 3;
    const int per = (npoints + parts - 1) / parts;
    const int lo = slice * per, hi = min(npoints, lo + per);
    float mx = -INFINITY, den = 0.0f, acc = 0.0f;
    constexpr int SV_ROWS = 16;                       // rows in flight per wave: a slice of <= 64 rows is ONE round trip
    for (int n0 = lo + wave; n0 < hi; n0 += SV_ROWS * ROWS_PER_BLOCK) {
        float l[SV_ROWS], v[SV_ROWS];
        bool ok[SV_ROWS];
#pragma unroll
        for (int u = 0; u < SV_ROWS; ++u) {           // the loop is latency-bound: all loads first
            const int n = n0 + u * ROWS_PER_BLOCK;
            const int nn = n < hi ? n : hi - 1;
            // plain loads, no short-circuit: a conditional load costs a full s_waitcnt vmcnt(0) each (measured: the
            // `a && b && c` form of the zero test serialised three round trips per row)
            const float px = p[nn * 3 + 0], py = p[nn * 3 + 1], pz = p[nn * 3 + 2];
            l[u] = (float)weight[fbase + (long)nn * C + cc];
            v[u] = (float)feature[fbase + (long)nn * C + cc];
            ok[u] = (n < hi) & live & !((px == 0.0f) & (py == 0.0f) & (pz == 0.0f));
        }
        // the batch's own softmax first (16 independent exponentials, no branch), then ONE merge into the running triple
        float bm = -INFINITY;
#pragma unroll
        for (int u = 0; u < SV_ROWS; ++u) bm = ok[u] ? fmaxf(bm, l[u]) : bm;
        if (bm > -INFINITY) {
            float d16 = 0.0f, a16 = 0.0f;
#pragma unroll
            for (int u = 0; u < SV_ROWS; ++u) {
                const float e = ok[u] ? ex(l[u] - bm) : 0.0f;
                d16 += e;
                a16 += e * v[u];
            }
            const float m2 = fmaxf(mx, bm), s0 = ex(mx - m2), s1 = ex(bm - m2);     // exp2(-inf) = 0 on the first batch
            den = den * s0 + d16 * s1;
            acc = acc * s0 + a16 * s1;
            mx = m2;
        }
    }
    part[0][wave][lane] = mx; part[1][wave][lane] = den; part[2][wave][lane] = acc;
    __syncthreads();
    if (wave == 0 && live) {
        float M = -INFINITY;
        for (int i = 0; i < ROWS_PER_BLOCK; ++i) M = fmaxf(M, part[0][i][lane]);
        float D = 0.0f, A = 0.0f;
        for (int i = 0; i < ROWS_PER_BLOCK; ++i) {
            if (part[1][i][lane] == 0.0f) continue;   // that wave saw no valid point
            const float sc = ex(part[0][i][lane] - M);
            D += part[1][i][lane] * sc;
            A += part[2][i][lane] * sc;
        }
        const SvPartials o = sv_partials(scratch, gridDim.y, C);
        const size_t at = ((size_t)b * ELO_SV_MAX_PARTS + slice) * C + c;
        o.mx[at] = M; o.den[at] = D; o.acc[at] = A;
    }
}

// merged softmax-pooled feature of channel c of batch element b (0 when no point is valid)
__device__ __forceinline__ float sv_merge(const SvPartials &s, int b, int c, int C, int parts)
{
    float M = -INFINITY;
    for (int i = 0; i < parts; ++i) M = fmaxf(M, s.mx[((size_t)b * ELO_SV_MAX_PARTS + i) * C + c]);
    float D = 0.0f, A = 0.0f;
    for (int i = 0; i < parts; ++i) {
        const size_t at = ((size_t)b * ELO_SV_MAX_PARTS + i) * C + c;
        if (s.den[at] == 0.0f) continue;
        const float sc = expf(s.mx[at] - M);
        D += s.den[at] * sc;
        A += s.acc[at] * sc;
    }
    return D > 0.0f ? A / D : 0.0f;
}

__global__ void softmax_valid_merge_kernel(float *scratch, int batch, int C, int parts, float *out, float *stats)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= batch * C) return;
    const SvPartials s = sv_partials(scratch, batch, C);
    const int b = i / C, c = i % C;
    out[i] = sv_merge(s, b, c, C, parts);
    if (stats) {                                       // the same merge again, keeping maximum and denominator
        float M = -INFINITY, D = 0.0f;
        for (int k = 0; k < parts; ++k) M = fmaxf(M, s.mx[((size_t)b * ELO_SV_MAX_PARTS + k) * C + c]);
        for (int k = 0; k < parts; ++k) {
            const size_t at = ((size_t)b * ELO_SV_MAX_PARTS + k) * C + c;
            if (s.den[at] != 0.0f) D += s.den[at] * expf(s.mx[at] - M);
        }
        stats[((size_t)b * 2) * C + c] = M;
        stats[((size_t)b * 2 + 1) * C + c] = D;
    }
}

// Scratch of a projection over `images` range images of H*W cells and `pts` points in total (32-bit words):
//   minr[cells] (min range bits per cell) | zflag[ZFLAGS * images] | cell_of[pts] | rbits[pts]
struct ProjScratch { unsigned *minr, *zflag; int *cell_of; unsigned *rbits; };

__host__ __device__ inline ProjScratch proj_scratch(unsigned *s, size_t cells, size_t images, size_t pts)
{
    return ProjScratch{s, s + cells, (int *)(s + cells + ZFLAGS * images), s + cells + ZFLAGS * images + pts};
}

// range bits, cell id and the atomicMin of the cell's range for one (already transformed) point of image b.
// The zero points of a padded / cropped scan (r = 0: tens of thousands in a 150 000-point KITTI cloud) all fall in ONE
// cell and win it (one of three, by the signs of the zeros: atan2f(+-0, +0) = +-0, atan2f(+0, -0) = pi, atan2f(-0, -0)
// = -pi; a warped invalid point is (w + t) * 0 = -0 where w + t < 0; asinf(0/0) is NaN whatever the sign: one row).
// As atomics they queue on one address at ~170 ns each (measured: 940 us for 300 000 points, 5.6 us without atomics;
// one speaker per wave behind a load filter still 200 us), so they do not touch minr at all: one lane per wave raises
// the flag of (image, zero cell) -- zflag <- 0, a plain store of a constant to a word no atomic touches -- and pass B
// lets nothing else win that cell.  For the other points a cell's value only ever decreases, so a (possibly stale)
// read that is already <= rb proves the atomicMin would change nothing.
__device__ __forceinline__ void bin_point(float x, float y, float z, long i, int b, int H, int W, float az_res, float vert_res,
                                          float vert_off, const ProjScratch &ps)
{
    const float r = sqrtf(x * x + y * y + z * z);
    const float at = atan2f(y, x);
    const int cell = cell_of_point(at, z, r, H, W, az_res, vert_res, vert_off);
    const unsigned rb = __float_as_uint(r);        // r >= 0: bit order == float order; NaN sorts last
    ps.cell_of[i] = cell;
    ps.rbits[i] = rb;
    if (rb == 0) {
        const int flag = b * ZFLAGS + zero_kind(at);
        const unsigned long long zeros = __ballot(1);
        const int leader = __ffsll((long long)zeros) - 1;
        if ((int)(threadIdx.x & 63) == leader || flag != __shfl(flag, leader)) ps.zflag[flag] = 0u;
        return;
    }
    unsigned *slot = ps.minr + (long)b * H * W + cell;
    if (__hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > rb) atomicMin(slot, rb);
}

// pass A of the projection for ONE point: warp by (q, t) (q == nullptr: no warp), range bits, cell id, atomicMin of the
// cell's range.  q, t may live in LDS (the fused pose-head + warp launch) or in global memory.
__device__ __forceinline__ void warp_cell_point(const elo_warp_project_args &a, long i, int b, const float *q, const float *t,
                                                const ProjScratch &ps)
{
    float x = a.xyz[i * 3 + 0], y = a.xyz[i * 3 + 1], z = a.xyz[i * 3 + 2];
    if (q) {
        const bool keep = !(x == 0.0f && y == 0.0f && z == 0.0f);          // pwclo_model.py:219-221
        const float q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
        // mul_q_point(q, [0,p])      model_util.py:17-36
        const float v0 = q0 * 0.0f - q1 * x - q2 * y - q3 * z;
        const float v1 = q0 * x + q1 * 0.0f + q2 * z - q3 * y;
        const float v2 = q0 * y - q1 * z + q2 * 0.0f + q3 * x;
        const float v3 = q0 * z + q1 * y - q2 * x + q3 * 0.0f;
        // inv_q                      model_util.py:61-69
        const float n2 = q0 * q0 + q1 * q1 + q2 * q2 + q3 * q3 + 1e-10f;
        const float i0 = q0 / n2, i1 = -q1 / n2, i2 = -q2 / n2, i3 = -q3 / n2;
        // mul_point_q(v, q^-1)[1:]   model_util.py:39-58
        const float w1 = v0 * i1 + v1 * i0 + v2 * i3 - v3 * i2;
        const float w2 = v0 * i2 - v1 * i3 + v2 * i0 + v3 * i1;
        const float w3 = v0 * i3 + v1 * i2 - v2 * i1 + v3 * i0;
        const float k = keep ? 1.0f : 0.0f;
        x = (w1 + t[0]) * k; y = (w2 + t[1]) * k; z = (w3 + t[2]) * k;
        a.warped[i * 3 + 0] = x; a.warped[i * 3 + 1] = y; a.warped[i * 3 + 2] = z;
    }
    bin_point(x, y, z, i, b, a.H, a.W, a.az_res, a.vert_res, a.vert_off, ps);
}

// ------------------------------------------------------------ pose head (one block per batch element)
__device__ __forceinline__ void hamilton(const float *a, const float *b, float *r)
{   // model_util.py:21-34
    r[0] = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
    r[1] = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
    r[2] = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
    r[3] = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
}

__device__ __forceinline__ void normalise_q(const float *q, float *o)
{   // pwclo_model.py:203
    const float n = sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3] + 1e-10f) + 1e-10f;
    for (int i = 0; i < 4; ++i) o[i] = q[i] / n;
}

// -DELO_POSE_CLOCK (a debugging build, tools/pose_clock.sh): thread 0 of workgroup (0, 0) of pose_head_kernel stamps the shader clock at
// its phase boundaries (elo_debug_pose_clock reads the stamps): where the 8-12 us of the smallest launch of a forward go.
#ifdef ELO_POSE_CLOCK
__device__ unsigned long long g_pose_clock[4 * 8 + 1];         // [launch % 4][stamp], [32]: launches so far (a forward has four pose heads)
#define POSE_STAMP(i) do { if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {                                    \
        if ((i) == 0) pose_slot = (int)(g_pose_clock[32]++ % 4);                                                          \
        g_pose_clock[pose_slot * 8 + (i)] = __builtin_readcyclecounter(); } } while (0)
#else
#define POSE_STAMP(i) do { } while (0)
#endif

// grid (X, batch).  X = 1: the plain pose head.  X > 1 (elo_pose_head_warp): every block of a batch element computes the
// same head redundantly -- identical instructions on identical inputs, a few microseconds -- block 0 stores it, and
// then each block warps its 256 points of the NEXT level's cloud by that pose and does pass A of the projection:
// the warp no longer waits for a kernel boundary after the pose head.
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void pose_head_kernel(const elo_pose_head_args a, const int parts,
                                                          const elo_warp_project_args w, const int with_warp)
{
    extern __shared__ float sm[];                      // [C] pooled feature, [hidden] big, [8] heads
    float *feat = sm, *big = sm + a.C, *head = big + a.hidden;
    const int b = blockIdx.y, tid = threadIdx.x;
    const SvPartials s = sv_partials(a.scratch, gridDim.y, a.C);
#ifdef ELO_POSE_CLOCK
    int pose_slot = 0;
#endif
    POSE_STAMP(0);                                     // first instruction of the workgroup
    // The model's head (C = 64, hidden = 256 = one unit per thread): this thread's column of W_big and its rows of
    // W_q / W_t do not depend on anything computed here -- requested now, they arrive while the slices are merged
    // (otherwise four dependent batches of 16 loads sit between the merge and the heads).
    constexpr int HEAD_C = 64;
    const bool model_head = a.C == HEAD_C && a.hidden == BLOCK;
    float wb[HEAD_C], wq[4], wt[3], bias_big = 0.0f;
    auto request_weights = [&]() {
        const int j = tid;                              // this thread owns hidden unit `tid`
#pragma unroll
        for (int c = 0; c < HEAD_C; ++c) wb[c] = a.W_big[(size_t)c * a.hidden + j];
#pragma unroll
        for (int o = 0; o < 4; ++o) wq[o] = a.W_q[(size_t)j * 4 + o];
#pragma unroll
        for (int o = 0; o < 3; ++o) wt[o] = a.W_t[(size_t)j * 3 + o];
        bias_big = a.b_big[j];
    };
    float *mpart = head + 8 + 8 * (BLOCK / ELO_WAVE);                  // [3][4][64]: the four slice groups of the merge
    // merge the slices: 4 threads per channel (C <= 64), each over every 4th slice, then a 4-way combine
    if (model_head) request_weights();
    const int c = tid & 63, q = tid >> 6;
    float M = -INFINITY, D = 0.0f, A = 0.0f;
    if (c < a.C) {
        // slices q, q + 4, ... of channel c; per trip MINE slices per thread, all their loads out together (unconditional, on
        // clamped indices: a load under a condition costs a full wait).  One trip of 16 for the <= 64 slices of a partial-sums
        // launch; trips of 32 for the row tiles of an MLP launch that computed the partial sums itself (elo_pose_head_args.
        // ready_parts: 225 tiles at l0 of a 64 x 1800 scan = two dependent round trips).  The two forms sit in a branch that is
        // uniform for the launch, each with its loads AND their use inside.
        auto trips = [&](auto mine) {
            constexpr int MINE = decltype(mine)::value;
            for (int i0 = q; i0 < parts; i0 += 4 * MINE) {
                float m_[MINE], d_[MINE], a_[MINE];
#pragma unroll
                for (int u = 0; u < MINE; ++u) {
                    const int i = i0 + 4 * u;
                    const size_t at = ((size_t)b * ELO_SV_MAX_PARTS + (i < parts ? i : 0)) * a.C + c;
                    m_[u] = s.mx[at]; a_[u] = s.acc[at];
                    d_[u] = i < parts ? s.den[at] : 0.0f;
                }
#pragma unroll
                for (int u = 0; u < MINE; ++u) {
                    if (d_[u] == 0.0f) continue;
                    if (m_[u] > M) { const float sc = exp_acc(M - m_[u]); D = D * sc + d_[u]; A = A * sc + a_[u]; M = m_[u]; }
                    else { const float sc = exp_acc(m_[u] - M); D += d_[u] * sc; A += a_[u] * sc; }
                }
            }
        };
        if (parts <= 64) trips(std::integral_constant<int, 16>());
        else trips(std::integral_constant<int, 32>());
    }
    POSE_STAMP(1);                                     // this thread's slices merged (the partial sums have landed)
    mpart[(0 * 4 + q) * 64 + c] = M; mpart[(1 * 4 + q) * 64 + c] = D; mpart[(2 * 4 + q) * 64 + c] = A;
    __syncthreads();
    if (tid < 64) {
        float MM = -INFINITY;
        for (int i = 0; i < 4; ++i) MM = fmaxf(MM, mpart[(0 * 4 + i) * 64 + tid]);
        float DD = 0.0f, AA = 0.0f;
        for (int i = 0; i < 4; ++i) {
            if (mpart[(1 * 4 + i) * 64 + tid] == 0.0f) continue;
            const float sc = exp_acc(mpart[(0 * 4 + i) * 64 + tid] - MM);
            DD += mpart[(1 * 4 + i) * 64 + tid] * sc; AA += mpart[(2 * 4 + i) * 64 + tid] * sc;
        }
        if (tid < a.C) feat[tid] = DD > 0.0f ? AA / DD : 0.0f;
    }
    for (int c2 = 64 + tid; c2 < a.C; c2 += blockDim.x) feat[c2] = sv_merge(s, b, c2, a.C, parts);   // C > 64 (not the model)

    __syncthreads();
    POSE_STAMP(2);                                         // softmax_valid's (C) vector is in LDS
    if (model_head) {                                      // conv1d C -> hidden, no activation (:197), same summation order
        float v = bias_big;
#pragma unroll
        for (int c = 0; c < HEAD_C; ++c) v += feat[c] * wb[c];
        big[tid] = v;
    } else {
        for (int j = tid; j < a.hidden; j += blockDim.x) {
            float v = a.b_big[j];
#pragma unroll 16
            for (int c = 0; c < a.C; ++c) v += feat[c] * a.W_big[(size_t)c * a.hidden + j];
            big[j] = v;
        }
    }
    __syncthreads();
    POSE_STAMP(3);                                         // hidden layer done (the W_big column was requested at the top)
    {   // conv1d hidden -> 4 (q) and hidden -> 3 (t): 7 dot products over `hidden`, reduced wave-wide then across waves
        float part[7] = {0, 0, 0, 0, 0, 0, 0};
        if (model_head) {
            const float bj = big[tid];
#pragma unroll
            for (int o = 0; o < 4; ++o) part[o] += bj * wq[o];
#pragma unroll
            for (int o = 0; o < 3; ++o) part[4 + o] += bj * wt[o];
        } else {
            for (int j = tid; j < a.hidden; j += blockDim.x) {
                const float bj = big[j];
#pragma unroll
                for (int o = 0; o < 4; ++o) part[o] += bj * a.W_q[(size_t)j * 4 + o];
#pragma unroll
                for (int o = 0; o < 3; ++o) part[4 + o] += bj * a.W_t[(size_t)j * 3 + o];
            }
        }
#pragma unroll
        for (int o = 0; o < 7; ++o)
            for (int d = 32; d >= 1; d >>= 1) part[o] += __shfl_xor(part[o], d, ELO_WAVE);
        float *wsum = head + 8;                       // [waves][8]
        if ((tid & 63) == 0)
            for (int o = 0; o < 7; ++o) wsum[(tid >> 6) * 8 + o] = part[o];
        __syncthreads();
        if (tid < 7) {
            float v = tid < 4 ? a.b_q[tid] : a.b_t[tid - 4];
            for (int w = 0; w < (int)(blockDim.x >> 6); ++w) v += wsum[w * 8 + tid];
            head[tid] = v;
        }
    }
    __syncthreads();
    POSE_STAMP(4);                                         // the 7 head values are in LDS
    if (tid == 0) {
        float q_det[4], q[4], t[3], qn[4];
        normalise_q(head, q_det);
        const float *t_det = head + 4;
        if (a.q_coarse) {
            const float *qc = a.q_coarse + b * 4, *tc = a.t_coarse + b * 3;
            const float tq[4] = {0.0f, tc[0], tc[1], tc[2]};
            float v[4], w[4], inv[4];
            hamilton(q_det, tq, v);                                                   // :275-276
            const float n2 = q_det[0] * q_det[0] + q_det[1] * q_det[1] + q_det[2] * q_det[2] + q_det[3] * q_det[3] + 1e-10f;
            inv[0] = q_det[0] / n2; inv[1] = -q_det[1] / n2; inv[2] = -q_det[2] / n2; inv[3] = -q_det[3] / n2;
            hamilton(v, inv, w);                                                      // :277
            hamilton(q_det, qc, q);                                                   // :279
            for (int i = 0; i < 3; ++i) t[i] = w[i + 1] + t_det[i];                   // :280
        } else {
            for (int i = 0; i < 4; ++i) q[i] = q_det[i];
            for (int i = 0; i < 3; ++i) t[i] = t_det[i];
        }
        normalise_q(q, qn);
        if (blockIdx.x == 0) {
            for (int i = 0; i < 4; ++i) { a.q[b * 4 + i] = q[i]; a.q_norm[b * 4 + i] = qn[i]; }
            for (int i = 0; i < 3; ++i) a.t[b * 3 + i] = t[i];
            if (a.pose7) {
                float *row = a.pose7 + b * 7;
                if (a.pose7_slots > 1 && a.pose7_cursor) {                     // ring: this replay's slot, then advance
                    const unsigned at = a.pose7_cursor[b];
                    a.pose7_cursor[b] = at + 1u;
                    row += (size_t)(at % (unsigned)a.pose7_slots) * a.batch * 7;
                }
                for (int i = 0; i < 4; ++i) row[i] = qn[i];
                for (int i = 0; i < 3; ++i) row[4 + i] = t[i];
            }
        }
        if (with_warp) {
            for (int i = 0; i < 4; ++i) head[8 + i] = q[i];                   // head[8..15): this level's (q, t) for the warp
            for (int i = 0; i < 3; ++i) head[12 + i] = t[i];
        }
    }
    POSE_STAMP(5);                                         // pose composed and stored
    if (a.next_orders.pool && blockIdx.x == 0 && blockIdx.y == 0) perm_refresh_block(a.next_orders);   // (uniform per workgroup)
    POSE_STAMP(6);                                         // (l0 only: next replay's visiting orders loaded)
    if (!with_warp) return;
    __syncthreads();
    const long n = (long)blockIdx.x * blockDim.x + tid;
    if (n < w.npoints) {
        warp_cell_point(w, (long)b * w.npoints + n, b, head + 8, head + 12,
                        proj_scratch(w.scratch, (size_t)w.batch * w.H * w.W, w.batch, (size_t)w.batch * w.npoints));
    }
    POSE_STAMP(7);                                         // this workgroup's 256 points warped and binned
}

// ------------------------------------------------------------ warp + spherical re-projection
// C: 32-bit words of features per cell (fp16 storage: channels / 2)
__global__ __launch_bounds__(ELO_BLOCK) void project_init_kernel(unsigned *minr, float *out_xyz, unsigned *out_feat,
                                                                 size_t cells, int C, int images)
{
    const size_t n_xyz = cells * 3, n_feat = cells * (size_t)C, total = cells + n_xyz + n_feat;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < (size_t)images * ZFLAGS; i += (size_t)gridDim.x * blockDim.x)
        minr[cells + i] = 0x7f7f7f7fu;                                    // zflag: "no zero point seen"
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        if (i < cells) minr[i] = 0x7f7f7f7fu;
        else if (i < cells + n_xyz) out_xyz[i - cells] = 0.0f;
        else out_feat[i - cells - n_xyz] = 0u;
    }
}

// pass A: one thread per point
__global__ __launch_bounds__(ELO_BLOCK) void warp_cell_kernel(const elo_warp_project_args a, const ProjScratch ps)
{
    const long total = (long)a.batch * a.npoints;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int b = point_batch(i, a.npoints);
        warp_cell_point(a, i, b, a.q ? a.q + b * 4 : nullptr, a.q ? a.t + b * 3 : nullptr, ps);
    }
}

// pass A of the raw-cloud input stage: PreProcess of one point of frame f (model_util.py:346-422) + binning.
// Stacked index: frame f of batch element b is image f*batch + b.
__global__ __launch_bounds__(ELO_BLOCK) void input_cell_kernel(const elo_input_stage_args a, const ProjScratch ps)
{
    const long per_frame = (long)a.batch * a.npoints, total = 2 * per_frame;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int f = i >= per_frame;                                     // 0: frame 1, 1: frame 2
        const long j = i - f * per_frame;
        const int b = point_batch(j, a.npoints);
        const long n = j - (long)b * a.npoints;
        const float *p = a.cloud + ((long)b * 2 * a.npoints + (long)f * a.npoints + n) * a.point_stride;
        float x = p[0], y = p[1], z = p[2], w = 1.0f;
        const float valid = (x != 0.0f || y != 0.0f || z != 0.0f) ? 1.0f : 0.0f;                  // :357-363
        if (sqrtf(__fadd_rn(__fmul_rn(x, x), __fmul_rn(y, y))) > a.crop_xy) x = y = z = w = 0.0f;   // :380-383 (no FMA: the
                                                                                                  // threshold decides like numpy's)
        if (a.T_trans && a.aug_frame[b] == f + 1) {                                               // :392-394, :408-410
            const float *T = a.T_trans + b * 16;
            const float nx = T[0] * x + T[1] * y + T[2] * z + T[3] * w;
            const float ny = T[4] * x + T[5] * y + T[6] * z + T[7] * w;
            const float nz = T[8] * x + T[9] * y + T[10] * z + T[11] * w;
            x = nx; y = ny; z = nz;
        }
        x *= valid; y *= valid; z *= valid;                                                       // :421-422
        a.points[i * 3 + 0] = x; a.points[i * 3 + 1] = y; a.points[i * 3 + 2] = z;
        bin_point(x, y, z, i, f * a.batch + b, a.H, a.W, a.az_res, a.vert_res, a.vert_off, ps);
    }
}

// pass B: the point(s) holding the cell minimum are summed into the cell (tf.scatter_nd adds duplicates).
// fp16 feature storage: a work item is a PAIR of channels, added with one packed fp16 atomic
__global__ __launch_bounds__(ELO_BLOCK) void scatter_min_kernel(const elo_warp_project_args a, const ProjScratch ps)
{
    const int f16 = a.feat_dtype == ELO_F16;
    const int CT = 3 + (f16 ? a.C / 2 : a.C);
    const long total = (long)a.batch * a.npoints * CT;
    const float *pts = a.q ? a.warped : a.xyz;
    const int zc0 = cell_of_point(0.0f, 0.0f, 0.0f, a.H, a.W, a.az_res, a.vert_res, a.vert_off);          // zero_kind 0
    const int zc1 = cell_of_point(atan2f(0.0f, -0.0f), 0.0f, 0.0f, a.H, a.W, a.az_res, a.vert_res, a.vert_off);
    const int zc2 = cell_of_point(atan2f(-0.0f, -0.0f), 0.0f, 0.0f, a.H, a.W, a.az_res, a.vert_res, a.vert_off);
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        // (e < 2^31 in every call of the model: the 32-bit forms of point_batch -- a 64-bit division is ~80 instructions, and this
        //  kernel is ~5 us of a batch-1 forward's critical path, three times)
        const long i = point_batch(e, CT);
        const int ch = (int)(e - i * CT);
        const int b = point_batch(i, a.npoints);
        const int c = ps.cell_of[i];
        const long cell = (long)b * a.H * a.W + c;
        // an image's zero points (range 0) win their cell: only they add there (zeros to xyz, their features to feat)
        const bool zero_won = (c == zc0 && ps.zflag[b * ZFLAGS] == 0u) || (c == zc1 && ps.zflag[b * ZFLAGS + 1] == 0u) ||
                              (c == zc2 && ps.zflag[b * ZFLAGS + 2] == 0u);
        if (ps.rbits[i] != (zero_won ? 0u : ps.minr[cell])) continue;
        // adding +-0 never changes a sum that started at +0 (x + 0 = x; 0 + -0 = +0): skipped, so the zero points of a
        // padded scan -- all winners of one cell -- do not queue on its three words
        if (ch >= 3 && f16) {
            typedef _Float16 half2v __attribute__((ext_vector_type(2)));
            const half2v hv = reinterpret_cast<const half2v *>(a.feat)[(i * a.C) / 2 + (ch - 3)];
            if (hv.x == (_Float16)0.0f && hv.y == (_Float16)0.0f) continue;
            __builtin_amdgcn_global_atomic_fadd_v2f16(reinterpret_cast<half2v *>(a.out_feat) + (cell * a.C) / 2 + (ch - 3), hv);
            continue;
        }
        const float v = ch < 3 ? pts[i * 3 + ch] : reinterpret_cast<const float *>(a.feat)[i * a.C + (ch - 3)];
        if (v == 0.0f) continue;
        if (ch < 3) atomicAdd(a.out_xyz + cell * 3 + ch, v);
        else atomicAdd(reinterpret_cast<float *>(a.out_feat) + cell * a.C + (ch - 3), v);
    }
}

#define ELO_REQUIRE(cond, who, what) \
    do { if (!(cond)) return fail(ELO_ERR_ARG, "%s: %s", who, what); } while (0)

}  // namespace
}  // namespace elo

using namespace elo;

extern "C" int elo_group_concat(const elo_group_concat_args *a, elo_stream_t stream)
{
    const char *who = "elo_group_concat";
    ELO_REQUIRE(a, who, "null argument block");
    ELO_REQUIRE(a->batch >= 0 && a->npoints > 0 && a->K > 0 && a->H2 > 0 && a->W2 > 0 && a->C >= 0, who, "bad sizes");
    ELO_REQUIRE(a->centre_xyz && a->src_xyz && (a->src_feat || a->C == 0) && a->idx && a->mask && a->out, who, "null tensor pointer");
    const long rows = (long)a->batch * a->npoints * a->K;
    if (rows == 0) return ELO_OK;
    hipLaunchKernelGGL(group_concat_kernel, dim3(grid_for_stage(rows)), dim3(ELO_BLOCK), sizeof(float) * (STAGE_ROWS * (3 + a->C) + 1) + STAGE_META_BYTES,
                       (hipStream_t)stream, *a, rows);
    return check_launch(who);
}

extern "C" int elo_masked_maxpool(const elo_masked_maxpool_args *a, elo_stream_t stream)
{
    const char *who = "elo_masked_maxpool";
    ELO_REQUIRE(a, who, "null argument block");
    ELO_REQUIRE(a->batch >= 0 && a->npoints > 0 && a->K > 0 && a->C > 0, who, "bad sizes");
    ELO_REQUIRE(a->x && a->mask && a->out, who, "null tensor pointer");
    const long rows = (long)a->batch * a->npoints;
    if (rows == 0) return ELO_OK;
    if (a->C % 4 == 0 && (((uintptr_t)a->x | (uintptr_t)a->out) & 15) == 0) {
        const long items = rows * (a->C / 4);
        hipLaunchKernelGGL(masked_maxpool_vec_kernel, dim3((unsigned)((items + ELO_BLOCK - 1) / ELO_BLOCK)), dim3(ELO_BLOCK), 0, (hipStream_t)stream, *a, items);
        return check_launch(who);
    }
    hipLaunchKernelGGL(masked_maxpool_kernel, dim3(grid_for_rows(rows)), dim3(ELO_BLOCK), 0, (hipStream_t)stream, *a, rows);
    return check_launch(who);
}

extern "C" int elo_cv_encode1(const elo_cv_encode1_args *a, elo_stream_t stream)
{
    const char *who = "elo_cv_encode1";
    ELO_REQUIRE(a, who, "null argument block");
    ELO_REQUIRE(a->batch >= 0 && a->npoints > 0 && a->K > 0 && a->H2 > 0 && a->W2 > 0 && a->C > 0, who, "bad sizes");
    ELO_REQUIRE(a->xyz1 && a->feat1 && a->xyz2 && a->feat2 && a->idx && a->mask && a->out, who, "null tensor pointer");
    const long rows = (long)a->batch * a->npoints * a->K;
    if (rows == 0) return ELO_OK;
    ELO_REQUIRE(a->dtype == ELO_F32 || a->dtype == ELO_F16, who, "dtype must be ELO_F32 or ELO_F16");
    const int esz = a->dtype == ELO_F16 ? 2 : 4;
    const bool vec = a->C % 2 == 0 && a->C < 500 && a->K < 32768 &&
                     ((uintptr_t)a->feat1 | (uintptr_t)a->feat2 | (uintptr_t)a->out) % (2 * esz) == 0;
    if (a->dtype == ELO_F16 && !vec) return fail(ELO_ERR_ARG, "%s: fp16 needs an even C < 500 and 4-byte aligned tensors", who);
    if (vec) {
        const int per = enc_rows(rows);
        const FastDiv ds = fast_div(5 + a->C), dk = fast_div(a->K);
        hipStream_t s = (hipStream_t)stream;
        const int rpi = ELO_BLOCK / (5 + a->C), batch_rows = rpi * ENC_BATCH;
        // The staged form (the tile built in LDS, 16-byte accesses only) where the feature rows are whole 16-byte chunks and the tile fits.
        // HBM-cold at the 128 x 2048 level shapes, batch 8, fp16 storage: l0 13.3 -> 10.3 us (0.42 -> 0.55 of 8 TB/s), l1 8.9 -> 5.7,
        // l2_origin 15.7 -> 11.2; fp32 unchanged (15.3 us at l0: 82 MB, bound by its bytes there).  A "pair" form of the column-owner
        // kernel (two rows per 16-byte slot column) was also measured -- fp32 l0 16.6 -> 15.4 us, fp16 no gain -- and is superseded by
        // this one.  gpurun_out/r06/cold_levels_*staged.txt, cold_sweep_*_pairs.txt
        const int per16 = 16 / esz;
        if (ELO_ENCODE1_STAGED && a->C % per16 == 0 && rows >= 8192 &&
            ((uintptr_t)a->feat1 | (uintptr_t)a->feat2 | (uintptr_t)a->out) % 16 == 0) {
            const size_t row_bytes = (size_t)(10 + 2 * a->C) * esz;
            const int R = 128 * row_bytes <= 40 * 1024 ? 128 : 64;
            if (R * row_bytes <= 40 * 1024) {
                const dim3 sgrid((unsigned)((rows + R - 1) / R));
                const size_t lds = R * row_bytes;
                if (a->dtype == ELO_F16) {
                    if (R == 128) hipLaunchKernelGGL((cv_encode1_staged_kernel<128, half_t>), sgrid, dim3(ELO_BLOCK), lds, s, *a, rows, dk);
                    else hipLaunchKernelGGL((cv_encode1_staged_kernel<64, half_t>), sgrid, dim3(ELO_BLOCK), lds, s, *a, rows, dk);
                } else {
                    if (R == 128) hipLaunchKernelGGL((cv_encode1_staged_kernel<128, float>), sgrid, dim3(ELO_BLOCK), lds, s, *a, rows, dk);
                    else hipLaunchKernelGGL((cv_encode1_staged_kernel<64, float>), sgrid, dim3(ELO_BLOCK), lds, s, *a, rows, dk);
                }
                return check_launch(who);
            }
        }
        const bool col_tiles = rpi * (5 + a->C) * 16 >= ELO_BLOCK * 15;      // <= 1/16 of the lanes idle
        const int cper = per < 64 ? 64 : per;                                 // (the column-owner form keeps its 64-row workgroups on small calls)
        if (batch_rows <= cper && col_tiles) {
            const int span = cper / batch_rows * batch_rows;
            const dim3 cgrid((unsigned)((rows + span - 1) / span));
            if (a->dtype == ELO_F16) {
                if (cper == 128) hipLaunchKernelGGL((cv_encode1_col_kernel<128, half_t>), cgrid, dim3(ELO_BLOCK), 0, s, *a, rows, ds, dk, rpi, span);
                else hipLaunchKernelGGL((cv_encode1_col_kernel<64, half_t>), cgrid, dim3(ELO_BLOCK), 0, s, *a, rows, ds, dk, rpi, span);
            } else {
                if (cper == 128) hipLaunchKernelGGL((cv_encode1_col_kernel<128, float>), cgrid, dim3(ELO_BLOCK), 0, s, *a, rows, ds, dk, rpi, span);
                else hipLaunchKernelGGL((cv_encode1_col_kernel<64, float>), cgrid, dim3(ELO_BLOCK), 0, s, *a, rows, ds, dk, rpi, span);
            }
            return check_launch(who);
        }
        const dim3 grid((unsigned)((rows + per - 1) / per));
        if (a->dtype == ELO_F16) {
            if (per == 128) hipLaunchKernelGGL((cv_encode1_vec_kernel<128, half_t>), grid, dim3(ELO_BLOCK), 0, s, *a, rows, ds, dk);
            else if (per == 64) hipLaunchKernelGGL((cv_encode1_vec_kernel<64, half_t>), grid, dim3(ELO_BLOCK), 0, s, *a, rows, ds, dk);
            else hipLaunchKernelGGL((cv_encode1_vec_kernel<32, half_t>), grid, dim3(ELO_BLOCK), 0, s, *a, rows, ds, dk);
        } else {
            if (per == 128) hipLaunchKernelGGL((cv_encode1_vec_kernel<128, float>), grid, dim3(ELO_BLOCK), 0, s, *a, rows, ds, dk);
            else if (per == 64) hipLaunchKernelGGL((cv_encode1_vec_kernel<64, float>), grid, dim3(ELO_BLOCK), 0, s, *a, rows, ds, dk);
            else hipLaunchKernelGGL((cv_encode1_vec_kernel<32, float>), grid, dim3(ELO_BLOCK), 0, s, *a, rows, ds, dk);
        }
        return check_launch(who);
    }
    hipLaunchKernelGGL(cv_encode1_kernel, dim3(grid_for_stage(rows)), dim3(ELO_BLOCK), sizeof(float) * (STAGE_ROWS * (10 + 2 * a->C) + 1) + STAGE_META_BYTES,
                       (hipStream_t)stream, *a, rows);
    return check_launch(who);
}

extern "C" int elo_cv_encode2(const elo_cv_encode2_args *a, elo_stream_t stream)
{
    const char *who = "elo_cv_encode2";
    ELO_REQUIRE(a, who, "null argument block");
    ELO_REQUIRE(a->batch >= 0 && a->npoints > 0 && a->K > 0 && a->H > 0 && a->W > 0 && a->C > 0 && a->Cc > 0, who, "bad sizes");
    ELO_REQUIRE(a->npoints == a->H * a->W, who, "npoints must equal H*W (every pixel is a centre)");
    ELO_REQUIRE(a->xyz1 && a->feat1 && a->cost && a->idx && a->mask && a->xyz_cat && a->rest, who, "null tensor pointer");
    const long rows = (long)a->batch * a->npoints * a->K;
    if (rows == 0) return ELO_OK;
    ELO_REQUIRE(a->dtype == ELO_F32 || a->dtype == ELO_F16, who, "dtype must be ELO_F32 or ELO_F16");
    const int esz = a->dtype == ELO_F16 ? 2 : 4, per16 = 16 / esz;
    const bool vec = a->C % per16 == 0 && a->Cc % per16 == 0 && a->C + a->Cc < 2000 && a->K < 32768 &&
                     ((uintptr_t)a->feat1 | (uintptr_t)a->cost | (uintptr_t)a->rest) % 16 == 0 && (uintptr_t)a->xyz_cat % (2 * esz) == 0;
    if (a->dtype == ELO_F16 && !vec) return fail(ELO_ERR_ARG, "%s: fp16 needs C and Cc multiples of 8 and 16-byte aligned tensors", who);
    if (vec) {
        const int per = enc_rows(rows);
        const dim3 grid((unsigned)((rows + per - 1) / per));
        const FastDiv ds = fast_div((a->C + a->Cc) / per16), dk = fast_div(a->K);
        hipStream_t s = (hipStream_t)stream;
        if (a->dtype == ELO_F16) {
            if (per == 128) hipLaunchKernelGGL((cv_encode2_vec_kernel<128, half_t>), grid, dim3(ELO_BLOCK), 0, s, *a, rows, ds, dk);
            else if (per == 64) hipLaunchKernelGGL((cv_encode2_vec_kernel<64, half_t>), grid, dim3(ELO_BLOCK), 0, s, *a, rows, ds, dk);
            else hipLaunchKernelGGL((cv_encode2_vec_kernel<32, half_t>), grid, dim3(ELO_BLOCK), 0, s, *a, rows, ds, dk);
        } else {
            if (per == 128) hipLaunchKernelGGL((cv_encode2_vec_kernel<128, float>), grid, dim3(ELO_BLOCK), 0, s, *a, rows, ds, dk);
            else if (per == 64) hipLaunchKernelGGL((cv_encode2_vec_kernel<64, float>), grid, dim3(ELO_BLOCK), 0, s, *a, rows, ds, dk);
            else hipLaunchKernelGGL((cv_encode2_vec_kernel<32, float>), grid, dim3(ELO_BLOCK), 0, s, *a, rows, ds, dk);
        }
        return check_launch(who);
    }
    hipLaunchKernelGGL(cv_encode2_kernel, dim3(grid_for_stage(rows)), dim3(ELO_BLOCK), sizeof(float) * (STAGE_ROWS * (10 + a->C + a->Cc) + 1) + STAGE_META_BYTES,
                       (hipStream_t)stream, *a, rows);
    return check_launch(who);
}

extern "C" int elo_masked_softmax_pool(const elo_softmax_pool_args *a, elo_stream_t stream)
{
    const char *who = "elo_masked_softmax_pool";
    ELO_REQUIRE(a, who, "null argument block");
    ELO_REQUIRE(a->batch >= 0 && a->npoints > 0 && a->K > 0 && a->C > 0 && a->values_stride >= a->C, who, "bad sizes");
    ELO_REQUIRE(a->logits && a->values && a->mask && a->out, who, "null tensor pointer");
    const long rows = (long)a->batch * a->npoints;
    if (rows == 0) return ELO_OK;
    ELO_REQUIRE(a->dtype == ELO_F32 || a->dtype == ELO_F16, who, "dtype must be ELO_F32 or ELO_F16");
    const int esz = a->dtype == ELO_F16 ? 2 : 4;
    const bool vec = a->C % 4 == 0 && a->C <= 1024 && ELO_BLOCK % (a->C / 4) == 0 && a->values_stride % 4 == 0 &&
                     ((uintptr_t)a->logits | (uintptr_t)a->values | (uintptr_t)a->out) % (4 * esz) == 0;
    if (a->dtype == ELO_F16 && !vec) return fail(ELO_ERR_ARG, "%s: fp16 needs C % 4 == 0 and 8-byte aligned tensors", who);
    // (fp32 storage only: in fp16 a point is 768 bytes per tensor -- even at two points per wave the form is instruction-bound,
    //  31.9 us against the quarter-wave form's 23.5 at the 128 x 2048 l0 shape, batch 8: gpurun_out/r06/cold_sweep_f16_pw1d.txt)
    if (vec && a->dtype == ELO_F32 && a->C == 64 && a->K <= 32 && a->values_stride % (16 / esz) == 0 &&
        ((uintptr_t)a->logits | (uintptr_t)a->values | (uintptr_t)a->out) % 16 == 0 && tuning().pool_wave) {   // wave per point (round 6): 16-byte loads, many light waves
        const long ppb = ELO_BLOCK / 64;                                         // points per workgroup
        const dim3 grid((unsigned)((rows + ppb - 1) / ppb));
        hipStream_t s = (hipStream_t)stream;
        const int J = (a->K + 3) / 4;                                            // neighbour rows per lane group
#define ELO_POOL_WAVE(J_) hipLaunchKernelGGL((softmax_pool_wave_kernel<float, J_>), grid, dim3(ELO_BLOCK), 0, s, *a, rows)
        if (J == 1) ELO_POOL_WAVE(1);
        else if (J == 2) ELO_POOL_WAVE(2);
        else if (J <= 4) ELO_POOL_WAVE(4);
        else ELO_POOL_WAVE(8);
#undef ELO_POOL_WAVE
        return check_launch(who);
    }
    if (vec) {
        const int rows_per_block = ELO_BLOCK / (a->C / 4);
        const dim3 grid((unsigned)((rows + rows_per_block - 1) / rows_per_block));
        hipStream_t s = (hipStream_t)stream;
        if (a->K % 6 == 0) {
            if (a->dtype == ELO_F16) hipLaunchKernelGGL((softmax_pool_vec_kernel<half_t, 6, true>), grid, dim3(ELO_BLOCK), 0, s, *a, rows);
            else hipLaunchKernelGGL((softmax_pool_vec_kernel<float, 6>), grid, dim3(ELO_BLOCK), 0, s, *a, rows);
        } else {
            if (a->dtype == ELO_F16) hipLaunchKernelGGL((softmax_pool_vec_kernel<half_t, 4, true>), grid, dim3(ELO_BLOCK), 0, s, *a, rows);
            else hipLaunchKernelGGL((softmax_pool_vec_kernel<float, 4>), grid, dim3(ELO_BLOCK), 0, s, *a, rows);
        }
        return check_launch(who);
    }
    hipLaunchKernelGGL(softmax_pool_kernel, dim3(grid_for_rows(rows)), dim3(ELO_BLOCK), 0, (hipStream_t)stream, *a, rows);
    return check_launch(who);
}

#ifdef ELO_POSE_CLOCK
extern "C" int elo_debug_pose_clock(unsigned long long *out33)
{
    return hipMemcpyFromSymbol(out33, HIP_SYMBOL(g_pose_clock), 33 * sizeof(unsigned long long)) == hipSuccess ? 0 : -3;
}
#endif

static int sv_parts(int npoints)
{
    constexpr int SV_MAX_SLICES = 64;                // (the scratch layout has room for ELO_SV_MAX_PARTS: the row tiles of elo_mlp_args.sv_*)
    int parts = (npoints + 63) / 64;                 // >= 64 points (16 per wave) per block
    return parts < 1 ? 1 : parts > SV_MAX_SLICES ? SV_MAX_SLICES : parts;
}

extern "C" int elo_softmax_valid(const elo_softmax_valid_args *a, elo_stream_t stream)
{
    const char *who = "elo_softmax_valid";
    ELO_REQUIRE(a, who, "null argument block");
    ELO_REQUIRE(a->batch >= 0 && a->npoints > 0 && a->C > 0, who, "bad sizes");
    ELO_REQUIRE(a->feature && a->weight && a->xyz && a->out && a->scratch, who, "null tensor pointer");
    if (a->batch == 0) return ELO_OK;
    const int parts = sv_parts(a->npoints);
    hipLaunchKernelGGL(softmax_valid_partial_kernel<false>, dim3(parts, a->batch, (a->C + ELO_WAVE - 1) / ELO_WAVE),
                       dim3(ELO_BLOCK), 0, (hipStream_t)stream, a->feature, a->weight, a->xyz, a->npoints, a->C, parts,
                       a->scratch, ProjectionClear{nullptr, nullptr, nullptr, 0, 0, 0});
    hipLaunchKernelGGL(softmax_valid_merge_kernel, dim3((a->batch * a->C + ELO_BLOCK - 1) / ELO_BLOCK), dim3(ELO_BLOCK),
                       0, (hipStream_t)stream, a->scratch, a->batch, a->C, parts, a->out, a->stats);
    return check_launch(who);
}

static int check_warp_project(const elo_warp_project_args *a, const char *who)
{
    ELO_REQUIRE(a, who, "null argument block");
    ELO_REQUIRE(a->batch >= 0 && a->npoints > 0 && a->C >= 0 && a->H > 0 && a->W > 0, who, "bad sizes");
    ELO_REQUIRE(a->xyz && a->out_xyz && a->scratch, who, "null tensor pointer");
    ELO_REQUIRE(a->C == 0 || (a->feat && a->out_feat), who, "features requested without buffers");
    ELO_REQUIRE(!a->q || (a->t && a->warped), who, "warp requested without t / warped");
    ELO_REQUIRE(a->feat_dtype == ELO_F32 || a->feat_dtype == ELO_F16, who, "feat_dtype must be ELO_F32 or ELO_F16");
    ELO_REQUIRE(a->feat_dtype == ELO_F32 || a->C % 2 == 0, who, "fp16 feature storage needs an even C");
    return ELO_OK;
}

// pose head, optionally followed in the same launches by the warp + projection of `w` (nullptr: none)
static int pose_head_impl(const elo_pose_head_args *a, const elo_warp_project_args *w, elo_stream_t stream, const char *who)
{
    ELO_REQUIRE(a, who, "null argument block");
    ELO_REQUIRE(a->batch >= 0 && a->npoints > 0 && a->C > 0 && a->hidden > 0, who, "bad sizes");
    ELO_REQUIRE(a->feature && a->weight && a->xyz && a->W_big && a->b_big && a->W_q && a->b_q && a->W_t && a->b_t &&
                a->q && a->t && a->q_norm && a->scratch, who, "null tensor pointer");
    ELO_REQUIRE((a->q_coarse == nullptr) == (a->t_coarse == nullptr), who, "q_coarse and t_coarse go together");
    ELO_REQUIRE(!a->clear_scratch || (a->clear_xyz && a->clear_cells > 0 && a->clear_C >= 0 && (a->clear_feat || a->clear_C == 0)),
                who, "incomplete clear_* side job");
    ELO_REQUIRE(!a->clear_scratch || a->clear_cells * (a->clear_C > 3 ? a->clear_C : 3) < (1l << 31), who, "clear_* side job: more than 2^31 words");
    ELO_REQUIRE(a->feat_dtype == ELO_F32 || a->feat_dtype == ELO_F16, who, "feat_dtype must be ELO_F32 or ELO_F16");
    const int f16 = a->feat_dtype == ELO_F16;
    ELO_REQUIRE(!f16 || a->clear_C % 2 == 0, who, "fp16 feature storage needs an even clear_C");
    elo_warp_project_args wv{};
    if (w) {
        if (int rc = check_warp_project(w, who)) return rc;
        ELO_REQUIRE(w->batch == a->batch && w->warped, who, "the warp must have the pose head's batch and a `warped` output");
        ELO_REQUIRE(w->feat_dtype == a->feat_dtype, who, "the warp and the pose head must share one feat_dtype");
        ELO_REQUIRE(a->clear_scratch == w->scratch && a->clear_xyz == w->out_xyz && a->clear_feat == w->out_feat &&
                    a->clear_cells == (long)w->batch * w->H * w->W && a->clear_C == w->C, who,
                    "the warp's buffers must be the ones this call clears (clear_*)");
        wv = *w;
        wv.q = a->q;                                  // pass B only asks whether a warp happened (reads `warped`)
        wv.t = a->t;
    }
    ELO_REQUIRE(a->ready_parts >= 0 && a->ready_parts <= ELO_SV_MAX_PARTS, who, "ready_parts is 0..ELO_SV_MAX_PARTS");
    ELO_REQUIRE(!a->ready_parts || a->C == 64, who, "ready_parts: the partial sums of an MLP launch are 64 channels wide");
    if (a->batch == 0) return ELO_OK;
    hipStream_t s = (hipStream_t)stream;
    {
        const int parts = a->ready_parts ? a->ready_parts : sv_parts(a->npoints);
        const ProjectionClear clear{a->clear_scratch, a->clear_xyz, (unsigned *)a->clear_feat, a->clear_cells,
                                    f16 ? a->clear_C / 2 : a->clear_C, a->batch};
        const dim3 pgrid(parts, a->batch, (a->C + ELO_WAVE - 1) / ELO_WAVE);
        if (a->ready_parts) {}                            // the launch that produced feature / weight wrote the partial sums (and cleared)
        else if (f16)
            hipLaunchKernelGGL(softmax_valid_partial_kernel<true>, pgrid, dim3(ELO_BLOCK), 0, s, a->feature, a->weight, a->xyz,
                               a->npoints, a->C, parts, a->scratch, clear);
        else
            hipLaunchKernelGGL(softmax_valid_partial_kernel<false>, pgrid, dim3(ELO_BLOCK), 0, s, a->feature, a->weight, a->xyz,
                               a->npoints, a->C, parts, a->scratch, clear);
        const size_t lds = sizeof(float) * ((size_t)a->C + a->hidden + 8 + 8 * (ELO_BLOCK / ELO_WAVE) + 3 * 4 * 64);
        const unsigned xb = w ? (unsigned)((w->npoints + ELO_BLOCK - 1) / ELO_BLOCK) : 1u;
        hipLaunchKernelGGL((pose_head_kernel<ELO_BLOCK>), dim3(xb, a->batch), dim3(ELO_BLOCK), lds, s, *a, parts, wv, w ? 1 : 0);
    }
    if (w) {
        const size_t cells = (size_t)w->batch * w->H * w->W, pts = (size_t)w->batch * w->npoints;
        const size_t elems = pts * (3 + (f16 ? w->C / 2 : w->C));
        const unsigned gb = (unsigned)((elems + ELO_BLOCK - 1) / ELO_BLOCK);
        hipLaunchKernelGGL(scatter_min_kernel, dim3(gb > 4096 ? 4096 : gb), dim3(ELO_BLOCK), 0, s, wv,
                           proj_scratch(w->scratch, cells, w->batch, pts));
    }
    return check_launch(who);
}

extern "C" int elo_pose_head(const elo_pose_head_args *a, elo_stream_t stream)
{
    return pose_head_impl(a, nullptr, stream, "elo_pose_head");
}

extern "C" int elo_pose_head_warp(const elo_pose_head_args *a, const elo_warp_project_args *w, elo_stream_t stream)
{
    const char *who = "elo_pose_head_warp";
    ELO_REQUIRE(w, who, "null warp block");
    return pose_head_impl(a, w, stream, who);
}

extern "C" int elo_input_stage(const elo_input_stage_args *a, elo_stream_t stream)
{
    const char *who = "elo_input_stage";
    ELO_REQUIRE(a, who, "null argument block");
    ELO_REQUIRE(a->batch >= 0 && a->npoints > 0 && a->H > 0 && a->W > 0 && a->point_stride >= 3, who, "bad sizes");
    ELO_REQUIRE((a->T_trans == nullptr) == (a->aug_frame == nullptr), who, "T_trans and aug_frame come together");
    ELO_REQUIRE(a->az_res > 0.0f && a->vert_res > 0.0f, who, "bad projection constants");
    if (a->batch == 0) return ELO_OK;                 // (an empty batch has no buffers to name)
    ELO_REQUIRE(a->cloud && a->points && a->out_xyz && a->scratch, who, "null tensor pointer");
    hipStream_t s = (hipStream_t)stream;
    const size_t images = 2 * (size_t)a->batch, cells = images * a->H * a->W, pts = images * a->npoints;
    const ProjScratch ps = proj_scratch(a->scratch, cells, images, pts);
    const unsigned gi = (unsigned)((cells * 4 + ELO_BLOCK - 1) / ELO_BLOCK);
    hipLaunchKernelGGL(project_init_kernel, dim3(gi > 4096 ? 4096 : gi), dim3(ELO_BLOCK), 0, s, ps.minr, a->out_xyz,
                       (unsigned *)nullptr, cells, 0, (int)images);
    const unsigned ga = (unsigned)((pts + ELO_BLOCK - 1) / ELO_BLOCK);
    hipLaunchKernelGGL(input_cell_kernel, dim3(ga > 8192 ? 8192 : ga), dim3(ELO_BLOCK), 0, s, *a, ps);
    elo_warp_project_args w = {};
    w.batch = (int)images; w.npoints = a->npoints; w.C = 0; w.H = a->H; w.W = a->W;
    w.az_res = a->az_res; w.vert_res = a->vert_res; w.vert_off = a->vert_off;
    w.xyz = a->points; w.out_xyz = a->out_xyz; w.scratch = a->scratch;
    const size_t elems = pts * 3;
    const unsigned gb = (unsigned)((elems + ELO_BLOCK - 1) / ELO_BLOCK);
    hipLaunchKernelGGL(scatter_min_kernel, dim3(gb > 8192 ? 8192 : gb), dim3(ELO_BLOCK), 0, s, w, ps);
    return check_launch(who);
}

extern "C" int elo_warp_project(const elo_warp_project_args *a, elo_stream_t stream)
{
    const char *who = "elo_warp_project";
    if (int rc = check_warp_project(a, who)) return rc;
    if (a->batch == 0) return ELO_OK;
    hipStream_t s = (hipStream_t)stream;
    const size_t cells = (size_t)a->batch * a->H * a->W, pts = (size_t)a->batch * a->npoints;
    const ProjScratch ps = proj_scratch(a->scratch, cells, a->batch, pts);
    // one init launch instead of three memsets (memset nodes inside a captured hipGraph proved unreliable):
    // minr <- 0x7f7f7f7f (3.39e38f: above every finite range, below NaN bit patterns), outputs <- 0
    const int f16 = a->feat_dtype == ELO_F16, feat_words = f16 ? a->C / 2 : a->C;
    if (!a->prepared) {
        const size_t words = cells * (1 + 3 + (size_t)feat_words);
        const unsigned gi = (unsigned)((words + ELO_BLOCK - 1) / ELO_BLOCK);
        hipLaunchKernelGGL(project_init_kernel, dim3(gi > 4096 ? 4096 : gi), dim3(ELO_BLOCK), 0, s, ps.minr, a->out_xyz,
                           (unsigned *)a->out_feat, cells, feat_words, a->batch);
    }
    const unsigned ga = (unsigned)((pts + ELO_BLOCK - 1) / ELO_BLOCK);
    hipLaunchKernelGGL(warp_cell_kernel, dim3(ga > 4096 ? 4096 : ga), dim3(ELO_BLOCK), 0, s, *a, ps);
    const size_t elems = pts * (3 + (size_t)feat_words);
    const unsigned gb = (unsigned)((elems + ELO_BLOCK - 1) / ELO_BLOCK);
    hipLaunchKernelGGL(scatter_min_kernel, dim3(gb > 4096 ? 4096 : gb), dim3(ELO_BLOCK), 0, s, *a, ps);
    return check_launch(who);
}
