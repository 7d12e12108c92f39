// elo_fused.hip -- fused "gather -> 1x1-conv chain -> pool" inference kernels for MI355X (gfx950).
//
// Why: at batch 1 a frame pair is ~300 launches of 4-14 us each even after the per-operator
// fusion of elo_features.hip + hipBLASLt GEMMs with bias/ReLU epilogues (profiles/r01_b): the
// command processor, not HBM or the matrix cores, is the limit.  These kernels collapse every
// operator between two poolings into ONE launch.
//
// How: four waves own a tile of 16 or 32 consecutive rows (a row = one (b,n,k) neighbour slot, or one
// point for the row-wise MLPs).  The tile's activations live in LDS as MATRIX-CORE OPERANDS: every four
// consecutive columns of a row are one 16-byte "quad" [hi0 hi1 hi2 hi3 | lo0 lo1 lo2 lo3] of fp16 halves
// (x = hi + lo to 2^-20), written once by whoever produces them (the gather, or the previous layer's
// epilogue) and read with ONE ds_read_b128 and no conversion by every wave that consumes them.  A layer is
//     D^T[Np x TILE] = W^T[Np x Kp] * A^T[Kp x TILE]       on v_mfma_f32_16x16x16_f16, three products
// -- the TRANSPOSED product: the W fragment is the MFMA's A operand, the activation quad its B operand, so
// a lane ends up with four CONSECUTIVE output channels of one row: one quad, one ds_write_b128 (or one
// 16-byte / 8-byte global store for a layer whose output leaves the kernel).  W is streamed from L2 in
// pre-packed fragment order (one contiguous 1 KiB load per wave-instruction, prefetched one step ahead).
// Layers run in place on the tile (barrier, write, barrier).  Poolings (masked max / masked
// softmax-weighted sum over the K rows of a point) read the last layer's plain-fp32 output column-wise from
// LDS and write (b,n,C) rows coalesced.  Feature tensors in HBM are fp32 or fp16 (feat_dtype).
//
// Interfaces and the reference lines covered: include/elo.h ("Fused inference kernels").
#include "elo_group_device.h"
#include "elo_project_device.h"
#include <atomic>
#include <cstdlib>
#include <type_traits>

namespace elo {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));

constexpr int FUSED_BLOCK = 256;   // 4 waves cooperate on one tile of rows
constexpr int FUSED_WAVES = 4;
// A thread's index within the 4-wave group that works on one tile.  Every tile kernel but mlp_sv_kernel's paired form is
// launched with FUSED_BLOCK threads and says so in its __launch_bounds__ (the compiler then knows threadIdx.x < 256 and the
// mask folds away); the paired form runs TWO such groups, one tile each, in a 512-thread workgroup.
__device__ __forceinline__ int ftid() { return (int)(threadIdx.x & (FUSED_BLOCK - 1)); }

// how a kernel instance computes its products (template parameter MODE)
constexpr int MODE_SPLIT = 0;      // fp32-class: hi/lo split operands, three fp16 MFMA products
constexpr int MODE_HALF = 1;       // ELO_PRODUCTS_HALF: operands rounded to nearest fp16, one product
constexpr int MODE_CHECKED = 2;    // MODE_SPLIT + a count of operands outside the fp16 range (elo_range_check)

__device__ unsigned long long g_range_violations;      // written by MODE_CHECKED instances only
__device__ unsigned long long g_range_snapshot;        // elo_range_violations: the count taken out by ONE atomic exchange
// read-and-clear in one atomic: a violation recorded by a checked replay on another stream between "read" and "zero" (two
// separate copies until round 4) was erased and that lane's later collection saw 0
__global__ void range_take_kernel() { g_range_snapshot = atomicExch(&g_range_violations, 0ull); }

// -DELO_CV1_CLOCK (a debugging build, tools/cv1_clock.sh): wave 0 of workgroup 0 of cv1_kernel stamps s_memtime at its
// phase boundaries; elo_debug_cv1_clock() reads the stamps
#ifdef ELO_CV1_CLOCK
#define CV1_STAMP(i) do { if (block == 0 && threadIdx.x == 0) g_cv1_clock[i] = __builtin_readcyclecounter(); } while (0)
#ifndef ELO_RR_CLOCK_BLOCK
#define ELO_RR_CLOCK_BLOCK 0
#endif
// cv1_rr_kernel (tools/rr_clock.sh): wave 0 of workgroup ELO_RR_CLOCK_BLOCK, and the same workgroup's wave 7 twelve slots on
#define RR_STAMP(i) do { if (blockIdx.x == ELO_RR_CLOCK_BLOCK && (threadIdx.x & 63) == 0 && (threadIdx.x >> 6) % 7 == 0) \
                             g_cv1_clock[(i) + 12 * (threadIdx.x >> 9 | (threadIdx.x >> 6) / 7)] = __builtin_readcyclecounter(); } while (0)
#ifdef ELO_RR_CLOCK_POOL                           // the stamps go to the inside of the pooling tail instead of the layers
#define RR_POOL_STAMP(i) RR_STAMP(i)
#define RR_LAYER_STAMP(i) do { } while (0)
#else
#define RR_POOL_STAMP(i) do { } while (0)
#define RR_LAYER_STAMP(i) RR_STAMP(i)
#endif
#else
#define CV1_STAMP(i) do { } while (0)
#define RR_STAMP(i) do { } while (0)
#define RR_POOL_STAMP(i) do { } while (0)
#define RR_LAYER_STAMP(i) do { } while (0)
#endif

__device__ __forceinline__ int ceil8(int x) { return (x + 7) & ~7; }
__device__ __forceinline__ int ceil16(int x) { return (x + 15) & ~15; }

// LDS row stride (32-bit words) for `cols` columns.  An operand read is one ds_read_b128 per lane at word
// (lane % 16) * S + 4 * (lane / 16) + const; gfx950 serves it in four groups of 16 lanes ({0-3, 12-15, 20-27}, ...) over
// 64 banks; enumerated against the guide's lane-group table the groups are conflict-free for S % 64 in {4, 8, 24, 40, 56}
// (4 also keeps the epilogue's ds_write_b128 conflict-free).  Measured on cv1_kernel at batch 8 (SQ_LDS_BANK_CONFLICT):
// S % 64 = 8, 24, 40 -> 3.4 M conflict cycles per launch, 4, 12, 20, 28, 36 -> 6.2 M, 16 -> 15.4 M -- and the same
// kernel time for all but 16, so the rule only has to stay away from multiples of 16 (the operand reads as two 8-byte
// halves, which the compiler emitted before they were made explicit 16-byte reads, cost 11.7 M).
__host__ __device__ __forceinline__ int row_stride(int cols)
{
    int s = (cols + 3) & ~3;
    for (;; s += 4) {
        const int m = s & 63;
        if (m == 4 || m == 8 || m == 24 || m == 40 || m == 56) return s;
    }
}

// ---- W fragments through buffer addressing ------------------------------------------------------------------
// buffer_load_dwordx4 v, v_lane_off, s[rsrc], s_off: the 128-bit resource and the fragment's byte offset are scalars,
// the only vector operand is lane*16 -- no vector address arithmetic per load (a flat/global load needs a 64-bit
// vector address: 2-4 VALU per fragment, ~90 fragments per wave and tile in the cost-volume kernel).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t weight_rsrc(const float *w)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(w), 0, 0x7fffffff, 0x00020000);   // raw, DATA_FORMAT_32
}

__device__ __forceinline__ float4 weight_load(__amdgpu_buffer_rsrc_t rsrc, unsigned lane_off, int byte_off)
{
    return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane_off, byte_off, 0));
}

// W is packed per column block in K-PAIRS of 32 k (for v_mfma_f32_16x16x32_f16: the gfx950 instruction that does 8192
// multiply-adds in the 16 cycles v_mfma_f32_16x16x16_f16 needs for 4096) plus, for an odd number of 16-k blocks, one
// 16-k TAIL:   per (cb):  [pair 0 | pair 1 | ... | tail],  KS * BLOCK_BYTES bytes in all (KS = 16-k blocks of the layer)
//   pair (split modes):  2 KiB = [hi8 of every lane (1 KiB) | lo8 of every lane (1 KiB)], a lane's eight halves being
//                        W[32p + 4kq + j][col] for j = 0..3 followed by W[32p + 16 + 4kq + j][col]  (kq = lane >> 4,
//                        col = cb*16 + (lane & 15)) -- the k order in which the activation operand is assembled from the
//                        two quads of the pair, so that no LDS layout changes
//   tail (split modes):  1 KiB, a lane's 16 bytes = [hi4 | lo4] of W[16(KS-1) + 4kq + j][col]
//   MODE_HALF:           the same with the round-to-nearest halves only: pair 1 KiB (16 bytes per lane), tail 512 B (8 bytes)
//   ELO_DENSE_F32:       fp32 weights, a "pair" is simply two consecutive 16-k blocks of four floats per lane
template <int MODE> struct WFrag { static constexpr int BLOCK_BYTES = 1024; };
#ifndef ELO_DENSE_F32
template <> struct WFrag<MODE_HALF> { static constexpr int BLOCK_BYTES = 512; };
#endif

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
struct WPair { uint4 hi, lo; };                    // a lane's registers for one pair (MODE_HALF uses .hi only)
struct ActPair { uint4 hi, lo; };                  // the matching activation operand

template <int MODE>
__device__ __forceinline__ WPair pair_load(__amdgpu_buffer_rsrc_t rsrc, unsigned lane, int byte_off)
{
    WPair w;
    w.hi = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane * 16u, byte_off, 0));
    if constexpr (MODE == MODE_HALF) w.lo = uint4{0u, 0u, 0u, 0u};
    else w.lo = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane * 16u, byte_off + 1024, 0));
    return w;
}

// the tail's fragment, in .x/.y (MODE_HALF: four halves) or all four words (split: [hi4 | lo4]; fp32: four floats)
template <int MODE>
__device__ __forceinline__ uint4 tail_load(__amdgpu_buffer_rsrc_t rsrc, unsigned lane, int byte_off)
{
    if constexpr (MODE == MODE_HALF) {
        const uint2 v = __builtin_bit_cast(uint2, __builtin_amdgcn_raw_buffer_load_b64(rsrc, lane * 8u, byte_off, 0));
        return uint4{v.x, v.y, 0u, 0u};
    } else {
        return __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane * 16u, byte_off, 0));
    }
}

// ---- the tile's element format ------------------------------------------------------------------------------------
// v_mfma_f32_16x16x4_f32 costs 32 cycles per 1024 multiply-adds and, measured (SQ_VALU_MFMA_COEXEC_CYCLES = 0), does
// not overlap with vector work, so its busy time adds to the kernel time.  v_mfma_f32_16x16x16_f16 does 4096
// multiply-adds in 8 cycles.  Both operands are therefore split into fp16 hi + lo (x = hi + lo to 2^-20 relative: hi is
// x with the low 13 mantissa bits cleared -- exactly an fp16 value --, lo = x - hi, both packed with round-toward-zero,
// which saturates instead of producing inf) and a 16-k block is  hi*hi + hi*lo + lo*hi  with fp32 accumulation: 3 x 8
// cycles instead of 4 x 32, the dropped lo*lo term is 2^-20 relative.  W is split at packing time (fused.PackedDense);
// an activation is split ONCE, when it is written to the tile (12 vector instructions per quad; round 1 split it on
// every read: once per column-block pass of every wave, 384 of the 838 vector instructions of a cost-volume tile's
// layers).  ELO_DENSE_F32 keeps plain fp32 in the tile and uses the fp32 MFMA.
//
// A quad = four consecutive columns of one tile row = 16 bytes at word index row*S + col (col % 4 == 0):
//   split modes   [hi0 hi1 hi2 hi3 | lo0 lo1 lo2 lo3]        MODE_HALF   [rn0 rn1 rn2 rn3 | unused]
//   ELO_DENSE_F32 / "plain" layers (the input of a pooling)   four floats
__device__ __forceinline__ unsigned pk_rtz(float x, float y) { return __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(x, y)); }

// x - (float)h, h = half WHICH (0: low, 1: high) of the packed pair `hh`, as ONE v_fma_mix_f32 (the half is converted on
// the way in): written as fma(h, -1, x) with the -1 in a scalar register the optimiser cannot see through -- with a
// literal -1 it rewrites the expression as v_cvt_f32_f16 + v_sub_f32 (two instructions; an inline-asm v_fma_mix_f32
// would be invisible to the matrix-instruction hazard tables).
__device__ __forceinline__ float opaque_minus_one()
{
    float m;
    asm("s_mov_b32 %0, 0xbf800000" : "=s"(m));
    return m;
}

template <int WHICH>
__device__ __forceinline__ float minus_half(float x, unsigned hh)
{
    const half2v h = __builtin_bit_cast(half2v, hh);
    return __builtin_fmaf((float)(WHICH ? h.y : h.x), opaque_minus_one(), x);
}

template <int MODE>
__device__ __forceinline__ uint4 pack_quad(const float4 a, unsigned &violations)
{
#ifdef ELO_DENSE_F32
    return __builtin_bit_cast(uint4, a);
#else
    if constexpr (MODE == MODE_HALF) {
        const half2v p0 = half2v{(_Float16)a.x, (_Float16)a.y}, p1 = half2v{(_Float16)a.z, (_Float16)a.w};
        return uint4{__builtin_bit_cast(unsigned, p0), __builtin_bit_cast(unsigned, p1), 0u, 0u};
    } else {
        if constexpr (MODE == MODE_CHECKED) {
            const float m = fmaxf(fmaxf(fabsf(a.x), fabsf(a.y)), fmaxf(fabsf(a.z), fabsf(a.w)));
            violations += !(m < 65504.0f);                    // also counts NaN
        }
        // hi = x rounded toward zero to fp16 (one v_cvt_pkrtz per pair: for |x| >= 2^-14 that is x with its low 13 mantissa
        // bits cleared; a smaller x gets an fp16 subnormal), lo = x - hi EXACTLY, taken straight from the packed half by
        // v_fma_mix_f32 (no mask, no unpack: 8 instructions per quad instead of 12), then rounded toward zero as well
        const unsigned h01 = pk_rtz(a.x, a.y), h23 = pk_rtz(a.z, a.w);
        return uint4{h01, h23, pk_rtz(minus_half<0>(a.x, h01), minus_half<1>(a.y, h01)), pk_rtz(minus_half<0>(a.z, h23), minus_half<1>(a.w, h23))};
    }
#endif
}

// four fp16 values read from HBM (fp16 feature storage): exact operands, hi = x, lo = 0
__device__ __forceinline__ uint4 quad_of_halves(const uint2 h)
{
#ifdef ELO_DENSE_F32
    const half4 v = __builtin_bit_cast(half4, h);
    return __builtin_bit_cast(uint4, float4{(float)v.x, (float)v.y, (float)v.z, (float)v.w});
#else
    return uint4{h.x, h.y, 0u, 0u};
#endif
}

__device__ __forceinline__ void quad_store(float *act, int word, const uint4 q) { *reinterpret_cast<uint4 *>(act + word) = q; }

// one element of an operand-format region, as fp32 (the poolings read their VALUES this way: hi + lo is x to 2^-22)
__device__ __forceinline__ float act_get(const float *act, int word)
{
#ifdef ELO_DENSE_F32
    return act[word];
#else
    const unsigned short *h = reinterpret_cast<const unsigned short *>(act + (word & ~3)) + (word & 3);
    return (float)__builtin_bit_cast(_Float16, h[0]) + (float)__builtin_bit_cast(_Float16, h[4]);
#endif
}

// one element written on its own (the element-wise staging paths for widths that are not multiples of a quad)
template <int MODE>
__device__ __forceinline__ void act_put(float *act, int word, float v, unsigned &violations)
{
#ifdef ELO_DENSE_F32
    act[word] = v;
#else
    unsigned short *h = reinterpret_cast<unsigned short *>(act + (word & ~3)) + (word & 3);
    if constexpr (MODE == MODE_HALF) {
        h[0] = __builtin_bit_cast(unsigned short, (_Float16)v);
        h[4] = 0;
    } else {
        if constexpr (MODE == MODE_CHECKED) violations += !(fabsf(v) < 65504.0f);
        const float hi = __uint_as_float(__float_as_uint(v) & 0xffffe000u);
        const unsigned p = pk_rtz(hi, v - hi);
        h[0] = (unsigned short)p;
        h[4] = (unsigned short)(p >> 16);
    }
#endif
}

// MODE_CHECKED: one atomic per thread that saw something (none in a healthy run)
template <int MODE>
__device__ __forceinline__ void report_violations(unsigned violations, unsigned long long *counter)
{
    if constexpr (MODE == MODE_CHECKED)
        if (violations) atomicAdd(counter ? counter : &g_range_violations, (unsigned long long)violations);      // (a lane's own word, or the process-wide one)
}

// ---- feature tensors in HBM: fp32 or fp16 ------------------------------------------------------------------------
__device__ __forceinline__ float feat_load(const void *p, long i, int f16)
{
    return f16 ? (float)reinterpret_cast<const _Float16 *>(p)[i] : reinterpret_cast<const float *>(p)[i];
}

__device__ __forceinline__ void feat_store(void *p, long i, float v, int f16)
{
    if (f16) reinterpret_cast<_Float16 *>(p)[i] = (_Float16)v;
    else reinterpret_cast<float *>(p)[i] = v;
}

// four consecutive channels, i % 4 == 0 and the row 16-byte (fp16: 8-byte) aligned
__device__ __forceinline__ void feat_store4(void *p, long i, const float4 v, int f16)
{
    if (f16) {
        const half2v a = half2v{(_Float16)v.x, (_Float16)v.y}, b = half2v{(_Float16)v.z, (_Float16)v.w};
        *reinterpret_cast<uint2 *>(reinterpret_cast<_Float16 *>(p) + i) = uint2{__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b)};
    } else {
        *reinterpret_cast<float4 *>(reinterpret_cast<float *>(p) + i) = v;
    }
}

// ---- matrix-core steps, transposed: acc[t] += W_t^T(16 n x k) * A^T(k x 16 rows) ------------------------------------
// operand layouts (lane = i16 + 16*kq): A[m = i16][k-group kq], B[k-group kq][n = i16], D[m = 4kq + r][n = i16] in acc[r].
// With A := the W fragment (m = output channel) and B := the activation (n = tile row) a lane holds channels
// cb*16 + 4kq + 0..3 of row i16.  A PAIR is 32 k on v_mfma_f32_16x16x32_f16 (k-group = the eight halves described at
// WFrag), a TAIL 16 k on v_mfma_f32_16x16x16_f16; fp32-class = hi*hi + lo*hi + hi*lo.

// activation operand of pair p of the row `arow` points at (arow = act + row*S + in_off + 4*kq): the quads of its two
// 16-k blocks are 16 words apart; hi4 of each are words 0-1, lo4 words 2-3
__device__ __forceinline__ ActPair act_pair(const float *p)
{
#ifdef ELO_DENSE_F32
    return ActPair{*reinterpret_cast<const uint4 *>(p), *reinterpret_cast<const uint4 *>(p + 16)};
#else
    // two 16-byte reads (ds_read_b128: the tile, S and every column offset are multiples of 4 words), regrouped in registers
    const uint4 q0 = *reinterpret_cast<const uint4 *>(__builtin_assume_aligned(p, 16));
    const uint4 q1 = *reinterpret_cast<const uint4 *>(__builtin_assume_aligned(p + 16, 16));
    return ActPair{uint4{q0.x, q0.y, q1.x, q1.y}, uint4{q0.z, q0.w, q1.z, q1.w}};
#endif
}

// acc[rb][c0 + t] += W_t x A_rb for t < MT column blocks and rb < RB row blocks: one W fragment feeds every row block
// (products outermost, so consecutive MFMAs go to different accumulators)
// A_EXACT: the activations are exact fp16 values (read from fp16 feature storage: lo == 0), so the w_hi x a_lo product adds
// exact zeros and is not issued (2 matrix instructions per pair instead of 3)
template <int MODE, int MT, int RB, int TPW, bool A_EXACT = false>
__device__ __forceinline__ void mma_pair(f32x4 (&acc)[RB][TPW], int c0, const ActPair (&a)[RB], const WPair (&w)[MT])
{
#ifdef ELO_DENSE_F32
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                const float4 x = __builtin_bit_cast(float4, h ? a[rb].lo : a[rb].hi);
                const float xs[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
                for (int t = 0; t < MT; ++t) {
                    const float4 wv = __builtin_bit_cast(float4, h ? w[t].lo : w[t].hi);
                    const float ws[4] = {wv.x, wv.y, wv.z, wv.w};
                    acc[rb][c0 + t] = __builtin_amdgcn_mfma_f32_16x16x4f32(ws[j], xs[j], acc[rb][c0 + t], 0, 0, 0);
                }
            }
#else
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int t = 0; t < MT; ++t)
            acc[rb][c0 + t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, w[t].hi), __builtin_bit_cast(half8, a[rb].hi), acc[rb][c0 + t], 0, 0, 0);
    if constexpr (MODE != MODE_HALF) {
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int t = 0; t < MT; ++t)
                acc[rb][c0 + t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, w[t].lo), __builtin_bit_cast(half8, a[rb].hi), acc[rb][c0 + t], 0, 0, 0);
        if constexpr (!A_EXACT) {
#pragma unroll
            for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                for (int t = 0; t < MT; ++t)
                    acc[rb][c0 + t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, w[t].hi), __builtin_bit_cast(half8, a[rb].lo), acc[rb][c0 + t], 0, 0, 0);
        }
    }
#endif
}

// ---- gfx950: a 16x16x16 MFMA must not read a 16x16x32 MFMA's result as SrcC within 5 wait states -----------------
// Measured (tools/micro/mfma_srcc_hazard.hip, profiles/r04_mfma_srcc_hazard.txt): v_mfma_f32_16x16x16_f16 issued 0-4 wait
// states after a v_mfma_f32_16x16x32_f16 whose vDst is its SrcC reads the accumulator as it was BEFORE that instruction's
// update, in every lane; from 5 wait states on it is right.  Two MFMAs of the same shape, or 16x16x16 followed by
// 16x16x32, are interlocked by the hardware.  hipcc (ROCm 7.2, clang 22) treats "SrcC is exactly the previous MFMA's
// vDst" as needing no wait state whatever the two shapes are, so the 16-k tail step of a layer, which continues the
// accumulators of its pair steps, was right only while the scheduler happened to leave other instructions in between --
// the chain kernels' "waves without a W load give wrong rows" (finding 4) was this pair, two wait
// states apart, behind an s_waitcnt that usually stalled long enough.  The guard: ONE asm statement that takes every
// accumulator of the tail step in and out -- the pair MFMAs that write them are ordered before it and the tail MFMAs
// that read them after it by data dependence, whatever the scheduler does -- and holds six wait states (LLVM's figure for
// the not-forwardable case on gfx950: passes + 2).  tools/isa_mfma_hazard.py checks the built library for any such pair.
#ifndef ELO_DENSE_F32
template <int RB, int TPW, int MT>
__device__ __forceinline__ void mfma_shape_guard(f32x4 (&acc)[RB][TPW], int c0)
{
    static_assert(RB * MT == 1 || RB * MT == 2 || RB * MT == 4, "accumulators of one tail step");
    if constexpr (RB * MT == 1) asm volatile("s_nop 5" : "+v"(acc[0][c0]));
    else if constexpr (RB == 1 && MT == 2) asm volatile("s_nop 5" : "+v"(acc[0][c0]), "+v"(acc[0][c0 + 1]));
    else if constexpr (RB == 2 && MT == 1) asm volatile("s_nop 5" : "+v"(acc[0][c0]), "+v"(acc[1][c0]));
    else if constexpr (RB == 1 && MT == 4) asm volatile("s_nop 5" : "+v"(acc[0][c0]), "+v"(acc[0][c0 + 1]), "+v"(acc[0][c0 + 2]), "+v"(acc[0][c0 + 3]));
    else asm volatile("s_nop 5" : "+v"(acc[0][c0]), "+v"(acc[0][c0 + 1]), "+v"(acc[1][c0]), "+v"(acc[1][c0 + 1]));
}
#endif

// AFTER_PAIR: the accumulators were last written by mma_pair (false: by the bias initialisation only)
template <int MODE, int MT, int RB, int TPW, bool AFTER_PAIR = true, bool A_EXACT = false>
__device__ __forceinline__ void mma_tail(f32x4 (&acc)[RB][TPW], int c0, const uint4 (&a)[RB], const uint4 (&w)[MT])
{
#ifndef ELO_DENSE_F32
    if constexpr (AFTER_PAIR) mfma_shape_guard<RB, TPW, MT>(acc, c0);
#endif
#ifdef ELO_DENSE_F32
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            const float4 x = __builtin_bit_cast(float4, a[rb]);
            const float xs[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
            for (int t = 0; t < MT; ++t) {
                const float4 wv = __builtin_bit_cast(float4, w[t]);
                const float ws[4] = {wv.x, wv.y, wv.z, wv.w};
                acc[rb][c0 + t] = __builtin_amdgcn_mfma_f32_16x16x4f32(ws[j], xs[j], acc[rb][c0 + t], 0, 0, 0);
            }
        }
#else
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int t = 0; t < MT; ++t)
            acc[rb][c0 + t] = __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(half4, uint2{w[t].x, w[t].y}), __builtin_bit_cast(half4, uint2{a[rb].x, a[rb].y}), acc[rb][c0 + t], 0, 0, 0);
    if constexpr (MODE != MODE_HALF) {
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int t = 0; t < MT; ++t)
                acc[rb][c0 + t] = __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(half4, uint2{w[t].z, w[t].w}), __builtin_bit_cast(half4, uint2{a[rb].x, a[rb].y}), acc[rb][c0 + t], 0, 0, 0);
        if constexpr (!A_EXACT) {
#pragma unroll
            for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                for (int t = 0; t < MT; ++t)
                    acc[rb][c0 + t] = __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(half4, uint2{w[t].x, w[t].y}), __builtin_bit_cast(half4, uint2{a[rb].z, a[rb].w}), acc[rb][c0 + t], 0, 0, 0);
        }
    }
#endif
}

// ---- where a layer's output goes ----------------------------------------------------------------------------------
struct LayerOut {
    int lds_off;        // column of the tile the output starts at; < 0: not written to LDS
    int plain;          // 1: four plain floats per quad (the input of a pooling); 0: operand format (the next layer's input)
    void *gout;         // optional copy to HBM: (rows x N) feature tensor, nullptr = none
    long grow0, grows;  // global row of tile row 0, number of global rows
    int gf16;
};

__device__ __forceinline__ LayerOut to_tile(int col) { return LayerOut{col, 0, nullptr, 0, 0, 0}; }
__device__ __forceinline__ LayerOut to_pool(int col) { return LayerOut{col, 1, nullptr, 0, 0, 0}; }

// epilogue of one 16 x 16 sub-tile: this lane's four consecutive channels cb*16 + 4kq + 0..3 of tile row `row`
template <int MODE>
__device__ __forceinline__ void store_quad(float *act, int S, int row, int col, int N, const f32x4 acc, bool relu,
                                           const LayerOut &o, unsigned &violations)
{
    // ReLU as max(bits, 0) on the value read as a signed integer (negative floats and -0 are negative integers; "no ReLU" =
    // max(bits, INT_MIN)): one instruction per value and no branch (fmaxf costs a canonicalising v_max x, x on top)
    const int floor = relu ? 0 : (int)0x80000000;
    const float4 v{__int_as_float(max(__float_as_int(acc[0]), floor)), __int_as_float(max(__float_as_int(acc[1]), floor)),
                   __int_as_float(max(__float_as_int(acc[2]), floor)), __int_as_float(max(__float_as_int(acc[3]), floor))};
    // an output that also goes to HBM as fp16 continues in the tile AS STORED (rounded to fp16): a fused pair of layers
    // then computes exactly what the two separate launches would (and an fp16 value is an exact operand: no split)
    uint2 stored{0u, 0u};
    if (o.gout && o.gf16) {
        const half2v a = half2v{(_Float16)v.x, (_Float16)v.y}, b = half2v{(_Float16)v.z, (_Float16)v.w};
        stored = uint2{__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b)};
    }
    if (o.lds_off >= 0) {
        const int word = row * S + o.lds_off + col;
        if (o.plain && o.gout && o.gf16) {            // (mlp_sv_kernel: the pooled value is the fp16 the consumer would read back)
            const half2v a = __builtin_bit_cast(half2v, stored.x), b = __builtin_bit_cast(half2v, stored.y);
            quad_store(act, word, __builtin_bit_cast(uint4, float4{(float)a.x, (float)a.y, (float)b.x, (float)b.y}));
        } else if (o.plain) quad_store(act, word, __builtin_bit_cast(uint4, v));
        else if (o.gout && o.gf16) quad_store(act, word, quad_of_halves(stored));
        else quad_store(act, word, pack_quad<MODE>(v, violations));          // columns >= N hold relu(0 + 0) = 0: the next layer's K padding
    }
    if (o.gout) {
        const long gr = o.grow0 + row;
        if (gr < o.grows) {
            if (col + 3 < N && (N & 3) == 0) {
                if (o.gf16) *reinterpret_cast<uint2 *>(reinterpret_cast<_Float16 *>(o.gout) + gr * N + col) = stored;
                else *reinterpret_cast<float4 *>(reinterpret_cast<float *>(o.gout) + gr * N + col) = v;
            } else {
                const float e[4] = {v.x, v.y, v.z, v.w};
                for (int r = 0; r < 4; ++r)
                    if (col + r < N) feat_store(o.gout, gr * N + col + r, e[r], o.gf16);
            }
        }
    }
}

// ---- one dense layer on the block's tile ---------------------------------------------------
// D[TILE x Np] = relu?(A[TILE x Kp] * W + bias).  The tile is cut into 16x16 output sub-tiles; wave w owns column blocks
// w, w + 4, ... (TPW = 1, 2 or 4 of them) of EVERY row block: a W fragment, once in registers, feeds all TILE / 16 row
// blocks.  (Round 2 first gave a wave one row block and twice the column blocks: every fragment then crossed the CU's
// vector L1 once per row block -- 404 KiB per 32-row tile of the cost volume, 2.3 GB per batch-8 launch, ~85 % of what
// the L1s can deliver during the dense phase; tools/cv1_phases.sh.)  All accumulators live in registers, worked through in
// PASSES of at most two column blocks (the W registers of a 32-k pair are 8 per block; the A operands are re-read from
// LDS per pass).  W streams from L2 in packed order (see WFrag), one pair ahead in a second register set; loads are
// UNCONDITIONAL (indices clamped into the packed array): with per-lane predicates around them the compiler loses track
// of the outstanding loads and waits vmcnt(0) before every MFMA group.  Layers run IN PLACE on the tile: barrier after
// the K loops (every wave has finished reading A), write D, barrier.
//
// The cost-volume chains have fixed widths, so there the FIRST step of the next layer (its first pair, or its tail when
// the layer has a single 16-k block) and its bias are fetched into registers (`Pre`) while this layer's epilogue runs:
// a layer does not start with an exposed L2 round trip.
template <int TPW>
struct Pre {
    static constexpr int MT = TPW < 2 ? TPW : 2;
    WPair w[MT];                 // pass 0's first step: pair 0, or (KS == 1) the tail in .hi
    float4 b[TPW];
};

template <int TILE, int N>
struct Sub {                                          // column blocks per wave of a width-N layer: 1, 2 or 4
    static constexpr int RAW = ((N + 15) / 16 + FUSED_WAVES - 1) / FUSED_WAVES;
    static constexpr int TPW = RAW <= 1 ? 1 : RAW <= 2 ? 2 : 4;
};

template <int TILE, int TPW, int MODE>
__device__ __forceinline__ void prefetch(const elo_dense &L, Pre<TPW> &pre)
{
    constexpr int CSTEP = FUSED_WAVES, MT = Pre<TPW>::MT;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int kq = lane >> 4, cb0 = wave;
    const int KS = ceil16(L.K) >> 4, CB = ceil16(L.N) >> 4;
    const __amdgpu_buffer_rsrc_t wrsrc = weight_rsrc(L.w_packed);
#pragma unroll
    for (int t = 0; t < TPW; ++t)
        pre.b[t] = *reinterpret_cast<const float4 *>(L.bias + min(cb0 + t * CSTEP, CB - 1) * 16 + 4 * kq);
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        const int base = min(cb0 + t * CSTEP, CB - 1) * KS * WFrag<MODE>::BLOCK_BYTES;
        if (KS >= 2) pre.w[t] = pair_load<MODE>(wrsrc, lane, base);
        else { pre.w[t].hi = tail_load<MODE>(wrsrc, lane, base); pre.w[t].lo = uint4{0u, 0u, 0u, 0u}; }
    }
}

template <int TILE, int TPW, int MODE, bool HAS_PRE, int NEXT_TPW>
__device__ __forceinline__ void dense_impl(float *act, int S, int in_off, const elo_dense &L, const LayerOut &out,
                                           unsigned &violations, const Pre<TPW> *pre, const elo_dense *next, Pre<(NEXT_TPW ? NEXT_TPW : 1)> *next_pre)
{
    constexpr int RB = TILE / 16;                 // row blocks: 1 or 2, all of them this wave's
    constexpr int CSTEP = FUSED_WAVES;            // column-block stride between a wave's sub-tiles
    constexpr int MT = TPW < 2 ? TPW : 2, PASSES = TPW / MT;
    // the wave index as a SCALAR: column blocks and with them every W / bias base address live in SGPRs,
    // so a W load is `buffer_load_dwordx4 v, v_lane_off, s[rsrc], s_off` with no vector address arithmetic
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(ftid() >> 6);
    const int i16 = lane & 15, kq = lane >> 4;
    const int cb0 = wave;
    const int KS = ceil16(L.K) >> 4, CB = ceil16(L.N) >> 4;
    const int NP = KS >> 1;                        // 32-k pairs; KS & 1: a 16-k tail
    const bool tail = KS & 1;
    constexpr int BB = WFrag<MODE>::BLOCK_BYTES;
    f32x4 acc[RB][TPW];
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        const float4 bv = HAS_PRE ? pre->b[t] : *reinterpret_cast<const float4 *>(L.bias + min(cb0 + t * CSTEP, CB - 1) * 16 + 4 * kq);
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) acc[rb][t] = f32x4{bv.x, bv.y, bv.z, bv.w};
    }
    const float *arow = act + i16 * S + in_off + 4 * kq;              // row block rb: + rb * 16 * S
    const __amdgpu_buffer_rsrc_t wrsrc = weight_rsrc(L.w_packed);      // buffer addressing: V# + lane offset + scalar offset
#pragma unroll
    for (int pass = 0; pass < PASSES; ++pass) {
        const int c0 = pass * MT;
        int base[MT];                                                  // byte offset of each sub-tile's column block
#pragma unroll
        for (int t = 0; t < MT; ++t) base[t] = min(cb0 + (c0 + t) * CSTEP, CB - 1) * KS * BB;
        auto fetch = [&](int p, WPair (&buf)[MT]) {
#pragma unroll
            for (int t = 0; t < MT; ++t) buf[t] = pair_load<MODE>(wrsrc, lane, base[t] + min(p, NP - 1) * 2 * BB);
        };
        auto pair_step = [&](int p, const WPair (&buf)[MT]) {
            ActPair a[RB];
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) a[rb] = act_pair(arow + rb * 16 * S + p * 32);
            mma_pair<MODE, MT, RB, TPW>(acc, c0, a, buf);
        };
        WPair b0[MT], b1[MT];
        const bool from_pre = HAS_PRE && pass == 0;
        // the tail's fragment travels in the .hi of whichever register set the pair loop has released
        auto fetch_tail = [&](WPair (&buf)[MT]) {
#pragma unroll
            for (int t = 0; t < MT; ++t) buf[t].hi = tail_load<MODE>(wrsrc, lane, base[t] + (KS - 1) * BB);
        };
        auto tail_step = [&](const WPair (&buf)[MT]) {
            uint4 wt[MT], a[RB];
#pragma unroll
            for (int t = 0; t < MT; ++t) wt[t] = buf[t].hi;
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) a[rb] = *reinterpret_cast<const uint4 *>(__builtin_assume_aligned(arow + rb * 16 * S + (KS - 1) * 16, 16));
            mma_tail<MODE, MT, RB, TPW>(acc, c0, a, wt);
        };
        if (from_pre) {
#pragma unroll
            for (int t = 0; t < MT; ++t) b0[t] = pre->w[t];                  // pair 0, or (NP == 0) the tail in .hi
        } else if (NP > 0) {
            fetch(0, b0);
        } else {
            fetch_tail(b0);
        }
        int p = 0;
        for (; p + 1 < NP; p += 2) {                                         // two pairs per trip: b0, then b1
            fetch(p + 1, b1);
            pair_step(p, b0);
            if (p + 2 < NP) fetch(p + 2, b0);
            else if (tail) fetch_tail(b0);
            pair_step(p + 1, b1);
        }
        if (p < NP) {                                                        // an odd number of pairs: the last one sits in b0
            if (tail) fetch_tail(b1);
            pair_step(p, b0);
            if (tail) tail_step(b1);
        } else if (tail) {
            tail_step(b0);
        }
    }
    if (NEXT_TPW) prefetch<TILE, NEXT_TPW ? NEXT_TPW : 1, MODE>(*next, *next_pre);       // rides behind the two barriers
    __syncthreads();                               // all A reads done: the tile may be overwritten
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        const int cb = cb0 + t * CSTEP;
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
            if (cb < CB) store_quad<MODE>(act, S, rb * 16 + i16, cb * 16 + 4 * kq, L.N, acc[rb][t], L.relu, out, violations);
    }
    __syncthreads();
}

// runtime width (set-conv, row-wise MLP)
template <int TILE, int MODE>
__device__ __forceinline__ void dense(float *act, int S, int in_off, const elo_dense &L, const LayerOut &out, unsigned &violations)
{
    constexpr int CSTEP = FUSED_WAVES;
    const int per_wave = ((ceil16(L.N) >> 4) + CSTEP - 1) / CSTEP;      // column blocks per wave: 1 or 2 (N <= 128: check_dense)
    Pre<1> *none = nullptr;
    if (per_wave <= 1) dense_impl<TILE, 1, MODE, false, 0>(act, S, in_off, L, out, violations, (const Pre<1> *)nullptr, nullptr, none);
    else dense_impl<TILE, 2, MODE, false, 0>(act, S, in_off, L, out, violations, (const Pre<2> *)nullptr, nullptr, none);
}

// compile-time width N with the cross-layer prefetch; NEXT = width of the following layer (0 = none)
template <int TILE, int N, int NEXT, int MODE>
__device__ __forceinline__ void dense_pf(float *act, int S, int in_off, const LayerOut &out, const elo_dense &L,
                                         const Pre<Sub<TILE, N>::TPW> &pre, const elo_dense *next,
                                         Pre<(NEXT ? Sub<TILE, NEXT ? NEXT : 16>::TPW : 1)> *next_pre, unsigned &violations)
{
    dense_impl<TILE, Sub<TILE, N>::TPW, MODE, true, NEXT ? Sub<TILE, NEXT ? NEXT : 16>::TPW : 0>(
        act, S, in_off, L, out, violations, &pre, next, next_pre);
}

// ---- per-row gather metadata of a tile -------------------------------------------------------
struct TileMeta {
    int *cell;      // [32] flat (b*H2 + h)*W2 + w of the gathered pixel, -1 = row not in use
    float *mask;    // [32]
    float *cxyz;    // [96] centre xyz of the tile's points (+ [32] words behind it: their pixels, in-kernel grouping only)
};

__device__ __forceinline__ TileMeta tile_meta(float *lds, int rows, int S)
{
    float *base = lds + rows * S;
    return TileMeta{reinterpret_cast<int *>(base), base + 32, base + 64};
}

// rows of a tile = P points x K slots; thread r < TILE fetches its row's idx/mask once.
template <int TILE>
__device__ __forceinline__ void load_meta(const TileMeta &m, long first_point, long total_points, int P, int K,
                                          const int *__restrict__ idx, const float *__restrict__ mask, int H2, int W2)
{
    const int r = threadIdx.x;
    if (r < TILE) {
        const int pi = small_div(r, K);
        const long pt = first_point + pi;
        int cell = -1;
        float mk = 0.0f;
        if (pi < P && pt < total_points) {
            const long gr = pt * K + (r - pi * K);
            const int *id = idx + gr * 3;
            cell = (id[0] * H2 + id[1]) * W2 + id[2];
            mk = mask[gr];
        }
        m.cell[r] = cell;
        m.mask[r] = mk;
    }
}

// In-kernel grouping of a tile's P points (one wave per point): fills meta.cell / meta.mask (and the
// centres in meta.cxyz) exactly as load_meta would from the stand-alone grouping op's outputs.
// lds_off: [KT] decoded visiting order; scratch: FUSED_WAVES x 2*KT words (select-k only).
template <int TILE, bool SELECT>
__device__ __forceinline__ void group_tile(const TileMeta &m, int *lds_off, unsigned *scratch, const elo_group_spec &g,
                                           long first_point, long total_points, int npoints, int P, int K,
                                           const float *__restrict__ centre_grid, int H1, int W1,
                                           const int *__restrict__ centre_hw, const float *__restrict__ grid2,
                                           int H2, int W2, float *__restrict__ new_xyz)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: point index, batch, centre address stay off the vector unit
    const int KT = g.kernel_h * g.kernel_w;
    if (tid < TILE) { m.cell[tid] = -1; m.mask[tid] = 0.0f; }
    // The tile's centres: thread pi fetches point pi's pixel and xyz (two dependent round trips when the centres come
    // from an index tensor) NOW, beside the loads that stage the visiting order, and leaves them in LDS -- fetched by
    // the wave that groups the point they were two serial round trips in front of every point's probes (~3 k of the
    // ~10 k cycles a point takes: tools/cv1_clock.sh).
    int *chw = reinterpret_cast<int *>(m.cxyz + 96);                             // [32] (h << 16) | w of the centres
    {
        const long pt = first_point + tid;
        const bool mine = tid < P && pt < total_points;
        const long pq = mine ? pt : first_point;
        // (launchers keep batch * npoints below 2^31: 32-bit divisions -- a 64-bit one is a ~100-instruction routine)
        const int b = (int)((unsigned)pq / (unsigned)npoints), n = (int)((unsigned)pq - (unsigned)b * (unsigned)npoints);
        int hc, wc;                                          // (a branch: behind a select the division is computed either way)
        if (centre_hw) { hc = centre_hw[pq * 2 + 0]; wc = centre_hw[pq * 2 + 1]; }
        else { hc = n / W1; wc = n - hc * W1; }
        const float *c = centre_grid + (((long)b * H1 + hc) * W1 + wc) * 3;
        const float cx = c[0], cy = c[1], cz = c[2];
        if (mine) {
            chw[tid] = (hc << 16) | wc;
            m.cxyz[tid * 3 + 0] = cx; m.cxyz[tid * 3 + 1] = cy; m.cxyz[tid * 3 + 2] = cz;
            if (new_xyz) { new_xyz[pt * 3 + 0] = cx; new_xyz[pt * 3 + 1] = cy; new_xyz[pt * 3 + 2] = cz; }
        }
    }
    stage_offsets(lds_off, g.random_hw, g.kernel_h, g.kernel_w, g.decoded_hw);   // ends with __syncthreads()
    ELO_GROUP_STAMP(10);
    const float r2 = g.distance * g.distance;
    for (int pi = wave; pi < P; pi += FUSED_WAVES) {
        const long pt = first_point + pi;
        if (pt >= total_points) continue;
        const int b = (int)((unsigned)pt / (unsigned)npoints);
        auto uniform = [](float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); };   // back to SGPRs
        const int hwc = __builtin_amdgcn_readfirstlane(chw[pi]), hc = hwc >> 16, wc = hwc & 0xffff;
        const float cx = uniform(m.cxyz[pi * 3 + 0]), cy = uniform(m.cxyz[pi * 3 + 1]), cz = uniform(m.cxyz[pi * 3 + 2]);
        ELO_GROUP_STAMP(11);
        int *o_idx = g.idx_out ? g.idx_out + pt * K * 3 : nullptr;
        float *o_mask = g.mask_out ? g.mask_out + pt * K : nullptr;
        auto emit = [&](int slot, int hw) {
            m.cell[pi * K + slot] = (b * H2 + (hw >> 16)) * W2 + (hw & 0xffff);
            m.mask[pi * K + slot] = 1.0f;
            if (o_idx) { o_idx[slot * 3 + 0] = b; o_idx[slot * 3 + 1] = hw >> 16; o_idx[slot * 3 + 2] = hw & 0xffff; }
            if (o_mask) o_mask[slot] = 1.0f;
        };
        int count = 0;
        if (!(pick_max(sq3(cx, cy, cz), ELO_EPS) <= ELO_EPS)) {                // valid centre (:62-70)
            const GridBuf gb = grid_buffer(grid2 + (size_t)b * H2 * W2 * 3);      // b is scalar: the resource lives in SGPRs
            if (SELECT)
                count = wave_select_k(gb, H2, W2, KT, K, lds_off, div_stride(hc, g.stride_h), div_stride(wc, g.stride_w), cx, cy, cz, r2,
                                      scratch + (size_t)wave * select_scratch_words(KT, K),
                                      reinterpret_cast<int *>(scratch) + (size_t)wave * select_scratch_words(KT, K) + KT, emit);
            else
                count = wave_random_k(gb, H2, W2, KT, K, lds_off, div_stride(hc, g.stride_h), div_stride(wc, g.stride_w), cx, cy, cz, r2, emit);
        }
        ELO_GROUP_STAMP(16 + (pi >= FUSED_WAVES));
        if (pi < FUSED_WAVES) ELO_GROUP_STAMP(18);
        for (int k = count + lane; k < K; k += 64) {                           // zero-filled slots: index (0,0,0), mask 0
            m.cell[pi * K + k] = 0;
            m.mask[pi * K + k] = 0.0f;
            if (o_idx) { o_idx[k * 3 + 0] = 0; o_idx[k * 3 + 1] = 0; o_idx[k * 3 + 2] = 0; }
            if (o_mask) o_mask[k] = 0.0f;
        }
    }
    __syncthreads();
}


// ---- staging a tile's rows: vector loads, ALL requested before the first LDS write --------------------------
// A segment = W consecutive elements of one global feature row per tile row (fp32: W % 4 == 0, fp16: W % 8 == 0,
// rows 16-byte aligned), written to the tile's columns [col0, col0 + W) in operand format where the row's mask says so,
// zeros for masked rows and rows not in use.  A thread owns the 16-byte items tid, tid+256, ... of a segment; the
// segments of a gather are loaded back to back and stored afterwards, so the whole gather is ONE round trip to L2.
// (The element-per-thread loops this replaces had their load under a condition, which costs a full s_waitcnt vmcnt(0)
// per iteration: 4...13 dependent round trips per tile.)
#ifndef ELO_TILE_WAVES
#define ELO_TILE_WAVES 6                   // waves per SIMD setconv_kernel / mlp_kernel / cv2_kernel are compiled for
#endif
#ifndef ELO_CV1_WAVES
#define ELO_CV1_WAVES 5                    // waves per SIMD the cost-volume stage-1 kernels are compiled for
#endif
#ifndef ELO_TILE32_WAVES
#define ELO_TILE32_WAVES 5                 // ... and the 32-row instances of setconv_kernel / mlp_kernel: at 6 (80 VGPRs) they spill
#endif                                     // 6-8 registers to scratch (round 2's ISA metadata: private_segment_fixed_size 20-28)
constexpr int SEG_ITEMS = 4;                        // per thread: TILE(32) * W(128) * 4 bytes / 16 / FUSED_BLOCK

__device__ __forceinline__ bool seg_ok(const void *p, int W, int rows, int f16)
{
    const int per = f16 ? 8 : 4;
    return W > 0 && W % per == 0 && ((uintptr_t)p & 15) == 0 && rows * (W / per) <= SEG_ITEMS * FUSED_BLOCK;
}

__device__ __forceinline__ void seg_split(int it, int q, int &row, int &c4)
{
    if ((q & (q - 1)) == 0) { const int sh = __builtin_ctz(q); row = it >> sh; c4 = it & (q - 1); }   // uniform branch
    else { row = small_div(it, q); c4 = it - row * q; }
}

// rowof(row) -> source row (long), negative = row not in use (the load then reads source row 0 and is dropped)
template <int TILE, class RowOf>
__device__ __forceinline__ void seg_load(uint4 (&r)[SEG_ITEMS], const void *__restrict__ src, int W, int f16, RowOf rowof)
{
    const int q = W >> (f16 ? 3 : 2), items = TILE * q;
#pragma unroll
    for (int u = 0; u < SEG_ITEMS; ++u) {
        const int it = ftid() + u * FUSED_BLOCK < items ? ftid() + u * FUSED_BLOCK : items - 1;
        int row, c4;
        seg_split(it, q, row, c4);
        const long sr = rowof(row);
        r[u] = reinterpret_cast<const uint4 *>(src)[(sr < 0 ? 0 : sr) * q + c4];
    }
}

// keepof(row): the row's mask as a bool (masks are 0/1: x * mask is a select)
template <int TILE, int MODE, class RowOf, class KeepOf>
__device__ __forceinline__ void seg_store(float *act, int S, int col0, int W, int f16, const uint4 (&r)[SEG_ITEMS], RowOf rowof,
                                          KeepOf keepof, unsigned &violations)
{
    const int q = W >> (f16 ? 3 : 2), items = TILE * q;
#pragma unroll
    for (int u = 0; u < SEG_ITEMS; ++u) {
        const int it = ftid() + u * FUSED_BLOCK;
        if (it < items) {
            int row, c4;
            seg_split(it, q, row, c4);
            const bool keep = rowof(row) >= 0 && keepof(row);
            const uint4 z{0u, 0u, 0u, 0u};
            if (f16) {
                const int word = row * S + col0 + 8 * c4;
                quad_store(act, word, keep ? quad_of_halves(uint2{r[u].x, r[u].y}) : z);
                quad_store(act, word + 4, keep ? quad_of_halves(uint2{r[u].z, r[u].w}) : z);
            } else {
                quad_store(act, row * S + col0 + 4 * c4, keep ? pack_quad<MODE>(__builtin_bit_cast(float4, r[u]), violations) : z);
            }
        }
    }
}

// zero the columns [from, to) of every tile row (from, to multiples of 4: whole quads; else element-wise)
template <int TILE, int MODE>
__device__ __forceinline__ void zero_cols(float *act, int S, int from, int to)
{
    if (((from | to) & 3) == 0) {
        const int w = (to - from) >> 2;
        for (int e = ftid(); e < TILE * w; e += FUSED_BLOCK) { const int er = small_div(e, w); quad_store(act, er * S + from + 4 * (e - er * w), uint4{0u, 0u, 0u, 0u}); }
    } else {
        const int w = to - from;
        unsigned none = 0;
        for (int e = ftid(); e < TILE * w; e += FUSED_BLOCK) { const int er = small_div(e, w); act_put<MODE>(act, er * S + from + (e - er * w), 0.0f, none); }
    }
}

// the 10 geometry floats of a row + 6 zeros: [p, g*m, g*m - p, |g*m - p|]   utils/pointnet_util.py:54-62.
// Thread r < TILE loads its row's centre p and neighbour g (unconditional, clamped) -- call geo_load with the other
// loads of the gather, geo_store with the stores.
struct GeoRow { float p[3], g[3]; };

__device__ __forceinline__ GeoRow geo_load(const float *__restrict__ pc, const float *__restrict__ pg)
{
    return GeoRow{{pc[0], pc[1], pc[2]}, {pg[0], pg[1], pg[2]}};
}

// four quads at word `word`: the 10 geometry channels and the K padding of a 16-k block
template <int MODE>
__device__ __forceinline__ void geo_store(float *act, int word, const GeoRow &r, float m, bool used, unsigned &violations)
{
    const float g0 = r.g[0] * m, g1 = r.g[1] * m, g2 = r.g[2] * m;
    const float d0 = g0 - r.p[0], d1 = g1 - r.p[1], d2 = g2 - r.p[2];
    const float e = sqrtf(d0 * d0 + d1 * d1 + d2 * d2 + 1e-20f);
    const uint4 z{0u, 0u, 0u, 0u};
    quad_store(act, word, used ? pack_quad<MODE>(float4{r.p[0], r.p[1], r.p[2], g0}, violations) : z);
    quad_store(act, word + 4, used ? pack_quad<MODE>(float4{g1, g2, d0, d1}, violations) : z);
    quad_store(act, word + 8, used ? pack_quad<MODE>(float4{d2, e, 0.0f, 0.0f}, violations) : z);
    quad_store(act, word + 12, z);
}

// ---- poolings over the K rows of each point of the tile (one (point, channel) per thread) ------
// The K rows of a point are fetched in chunks of 8 INDEPENDENT LDS reads (a serial "read mask -> branch -> read
// value" chain cost ~0.3 us per neighbour, 7 us per cost-volume tile).  `off` names a PLAIN fp32 region.
__device__ __forceinline__ void pool_masked_max(const float *act, int S, int off, int C, const TileMeta &m, int P, int K,
                                                long first_point, long total_points, void *__restrict__ out, int f16)
{
    for (int q = threadIdx.x; q < P * C; q += FUSED_BLOCK) {
        const int pi = q / C, c = q - pi * C;
        const long pt = first_point + pi;
        if (pt >= total_points) continue;
        const float *col = act + (pi * K) * S + off + c;
        const float *mk = m.mask + pi * K;
        float best = -INFINITY;
        for (int k0 = 0; k0 < K; k0 += 8) {
            float v[8], w[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int k = min(k0 + u, K - 1);          // clamped duplicates do not change a maximum
                v[u] = col[k * S];
                w[u] = mk[k];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) best = fmaxf(best, v[u] * w[u]);
        }
        feat_store(out, pt * C + c, best, f16);
    }
}

// out = sum_k softmax_k(mask == 1 ? logit : -1e10) * value      (64 channels)
// e^x on the hardware exp2 (v_exp_f32: ~1 ulp), a fifth of the instructions of expf's range-reduced polynomial; the
// pooled softmax weights feed a convex combination, far inside the 1e-4 parity tolerance.
// logits: a PLAIN fp32 region; values: an OPERAND-format region (they were a layer's input).
__device__ __forceinline__ float exp_hw(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896f); }

// One chunk of up to 8 rows (n of them real) merged into the running (maximum, denominator, weighted sum): the chunk's own
// maximum first, then its 8 exponentials side by side, then ONE merge -- no branch.  (Until round 3 this was the textbook
// online form, a data-dependent branch and two dependent exponentials per row: 370 cycles per row, a THIRD of the
// register-resident kernel's time at K = 6: tools/rr_clock.sh.)  A single chunk (K <= 8) merges into (-inf, 0, 0) exactly:
// exp2(-inf) = 0, exp2(0) = 1.  All rows masked: every x is -1e10, the weights are uniform, as in the reference.
// NMAX = 6: the cost volumes' K = 6 as ONE chunk of six slots -- the eight-slot form computed two dead slots (a quarter of the
// ~100 instructions per point and channel; dead slots add exact zeros, so the sums are the same).
// FIRST: the chunk opens the running triple (it merges into (-inf, 0, 0): the merge's factors are exactly 0 and 1, so the triple IS
// the chunk's -- two exponentials and eight more instructions per point and channel not issued)
template <int NMAX = 8, bool FIRST = false>
__device__ __forceinline__ void softmax_chunk8(const float (&l)[NMAX], const float (&v)[NMAX], const float (&w)[NMAX], int n,
                                               float &mx, float &den, float &acc)
{
    float x[NMAX], cm = -INFINITY;
#pragma unroll
    for (int u = 0; u < NMAX; ++u) {
        x[u] = w[u] == 1.0f ? l[u] : -1e10f;
        cm = u < n ? fmaxf(cm, x[u]) : cm;
    }
    float cd = 0.0f, ca = 0.0f;
#pragma unroll
    for (int u = 0; u < NMAX; ++u) {
        const float e = u < n ? exp_hw(x[u] - cm) : 0.0f;
        cd += e;
        ca += e * v[u];
    }
    if constexpr (FIRST) { mx = cm; den = cd; acc = ca; return; }
    const float m2 = fmaxf(mx, cm), s0 = exp_hw(mx - m2), s1 = exp_hw(cm - m2);
    den = den * s0 + cd * s1;
    acc = acc * s0 + ca * s1;
    mx = m2;
}

__device__ __forceinline__ void pool_masked_softmax(const float *act, int S, int logit_off, int value_off,
                                                    const TileMeta &m, int P, int K, long first_point,
                                                    long total_points, void *__restrict__ out, int f16)
{
    for (int q = threadIdx.x; q < P * 64; q += FUSED_BLOCK) {
        const int pi = q >> 6, c = q & 63;
        const long pt = first_point + pi;
        if (pt >= total_points) continue;
        const float *lcol = act + (pi * K) * S + logit_off + c;
        const int vword = (pi * K) * S + value_off + c;
        const float *mk = m.mask + pi * K;
        float mx = -INFINITY, den = 0.0f, acc = 0.0f;
        if (K == 6) {                                     // (uniform)
            float l[6], v[6], w[6];
#pragma unroll
            for (int u = 0; u < 6; ++u) { l[u] = lcol[u * S]; v[u] = act_get(act, vword + u * S); w[u] = mk[u]; }
            softmax_chunk8<6, true>(l, v, w, 6, mx, den, acc);
        } else if (K == 4) {                              // stage 2 of the cost volume
            float l[4], v[4], w[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { l[u] = lcol[u * S]; v[u] = act_get(act, vword + u * S); w[u] = mk[u]; }
            softmax_chunk8<4, true>(l, v, w, 4, mx, den, acc);
        } else
        for (int k0 = 0; k0 < K; k0 += 8) {
            float l[8], v[8], w[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int k = min(k0 + u, K - 1);
                l[u] = lcol[k * S];
                v[u] = act_get(act, vword + k * S);
                w[u] = mk[k];
            }
            softmax_chunk8(l, v, w, K - k0, mx, den, acc);
        }
        feat_store(out, pt * 64 + c, acc / den, f16);
    }
}

// ================================================================ set-conv / set-upconv stage 1
// blockIdx.y selects one of two independent jobs of identical shape (e.g. the embedding and the embedding-mask
// set-upconv of a refinement level): one launch instead of two.
template <typename Args>
struct JobPair { Args job[2]; };                  // one kernarg block: job[blockIdx.y] is a uniform (scalar) access

// LDS columns: [0, C) gathered features * mask, [C, C+3) gathered xyz * mask - centre, zeros up to ceil16(3 + C).
// 5 waves per SIMD (<= 102 VGPRs).  History: 140 VGPRs uncapped (3 workgroups per CU); a cap of 4 cost 12 bytes of
// scratch and gave +4 % at batch 8; with the scalar wave index and buffer-addressed W fragments the kernels need 92-96
// VGPRs without any spill.  A cap of 6 (80 VGPRs) spills 36-144 bytes in the cost-volume kernels: not taken.
// `block` of `nblocks`: the tile this workgroup works on (blockIdx.x / gridDim.x of a plain launch, or the workgroup's
// position inside its share of a heterogeneous launch, cv1_setconv_kernel)
template <int TILE, int MODE>
__device__ __forceinline__ void setconv_tile(const elo_setconv_args &a, const int S, float *lds, unsigned block, unsigned nblocks)
{
    float *act = lds;
    const TileMeta meta = tile_meta(lds, TILE, S);
    const int tid = threadIdx.x;
    const int K = a.K, P = TILE / K, f16 = a.feat_dtype == ELO_F16;
    unsigned bad = 0;
    const long total_points = (long)a.batch * a.npoints;
    const long first_point = (long)xcd_tile(block, nblocks) * P;
    if (first_point >= total_points) return;
    if (a.group.random_hw) {
        int *lds_off = reinterpret_cast<int *>(meta.cxyz + 128);
        group_tile<TILE, false>(meta, lds_off, nullptr, a.group, first_point, total_points, a.npoints, P, K, a.xyz1_grid,
                                a.H, a.W, a.centre_hw, a.src_xyz, a.H2, a.W2, a.new_xyz);
    } else {
        load_meta<TILE>(meta, first_point, total_points, P, K, a.idx, a.mask, a.H2, a.W2);
        if (tid >= 64 && tid < 64 + P * 3) {              // centres of the tile's points (second wave)
            const int q = tid - 64, pi = q / 3, c = q - pi * 3;
            const long pt = first_point + pi;
            float v = 0.0f;
            if (pt < total_points) {
                if (a.centre_hw) {
                    const int b = point_batch(pt, a.npoints);
                    const int h = a.centre_hw[pt * 2 + 0], w = a.centre_hw[pt * 2 + 1];
                    v = a.xyz1_grid[(((long)b * a.H + h) * a.W + w) * 3 + c];
                    if (a.new_xyz) a.new_xyz[pt * 3 + c] = v;                   // :206
                } else {
                    v = a.centre_xyz[pt * 3 + c];
                }
            }
            meta.cxyz[q] = v;
        }
        __syncthreads();
    }
    // gather + centre-subtract + concat into act[row][0 .. CTp)                   :203-213 / :277-284
    const int C = a.C, CT = 3 + C, CTp = ceil16(CT);
    if (seg_ok(a.src_feat, C, TILE, f16)) {
        auto cell_of = [&](int row) { return (long)meta.cell[row]; };
        uint4 rf[SEG_ITEMS];
        seg_load<TILE>(rf, a.src_feat, C, f16, cell_of);
        float x = 0.0f, y = 0.0f, z = 0.0f;
        if (tid < TILE) {                                                     // thread t < TILE: the xyz quad of row t
            const float *p = a.src_xyz + (long)(meta.cell[tid] < 0 ? 0 : meta.cell[tid]) * 3;
            x = p[0]; y = p[1]; z = p[2];
        }
        seg_store<TILE, MODE>(act, S, 0, C, f16, rf, cell_of, [&](int row) { return meta.mask[row] != 0.0f; }, bad);
        if (tid < TILE) {
            const float m = meta.mask[tid];
            const float *c = meta.cxyz + small_div(tid, K) * 3;
            const float4 d{x * m - c[0], y * m - c[1], z * m - c[2], 0.0f};
            quad_store(act, tid * S + C, meta.cell[tid] >= 0 ? pack_quad<MODE>(d, bad) : uint4{0u, 0u, 0u, 0u});
        }
        zero_cols<TILE, MODE>(act, S, C + 4, CTp);
    } else {
        for (int e = tid; e < TILE * CTp; e += FUSED_BLOCK) {
            const int row = small_div(e, CTp), ch = e - row * CTp;
            const int cell = meta.cell[row];
            float v = 0.0f;
            if (cell >= 0 && ch < CT) {
                const float m = meta.mask[row];
                v = ch < C ? feat_load(a.src_feat, (long)cell * C + ch, f16) * m
                           : a.src_xyz[(long)cell * 3 + (ch - C)] * m - meta.cxyz[small_div(row, K) * 3 + (ch - C)];
            }
            act_put<MODE>(act, row * S + ch, v, bad);
        }
    }
    __syncthreads();
    for (int l = 0; l < a.n_layers; ++l)                                       // in place, :217-222
        dense<TILE, MODE>(act, S, 0, a.layers[l], l == a.n_layers - 1 ? to_pool(0) : to_tile(0), bad);
    pool_masked_max(act, S, 0, a.layers[a.n_layers - 1].N, meta, P, K, first_point, total_points, a.out, f16);   // :224-230
    report_violations<MODE>(bad, a.range_counter);
}

template <int TILE, int MODE>
__global__ __launch_bounds__(FUSED_BLOCK, TILE == 32 ? ELO_TILE32_WAVES : ELO_TILE_WAVES) void setconv_kernel(const JobPair<elo_setconv_args> jobs, const int S)
{
    extern __shared__ __align__(16) float lds[];
    setconv_tile<TILE, MODE>(jobs.job[blockIdx.y], S, lds, blockIdx.x, gridDim.x);
}


// ================================================================ set-conv with a narrow MLP (widths <= 32, K == 32)
// The first two set-conv layers (6->8->8->16 and 19->16->16->32 on 7200 / 1808 centres) are far too small for
// matrix-core tiles: a 4-wave tile spends its time in barriers and dependent loads.  Here HALF A WAVE owns a
// centre: its 32 lanes are the K = 32 neighbour rows.  Grouping (32 probes per step, ballot + popcount), the
// gather of the row, the whole MLP (registers, weights broadcast from LDS) and the max over the 32 lanes (DPP)
// happen without a single barrier after the weights are staged; 8 centres per workgroup.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_max_step(float v)
{
    const float o = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(-INFINITY), __float_as_int(v), CTRL,
                                                               ROW_MASK, 0xf, false));
    return fmaxf(o, v);
}

// the same on the BIT PATTERNS of non-negative floats (a ReLU output times a 0 / 1 mask): integer order = float order there,
// and the compiler folds update_dpp + max into ONE v_max_i32_dpp -- fmaxf is llvm.maxnum, which costs a canonicalising
// v_max x, x per operand that came out of a DPP move (3 instructions per step instead of 1: 240 of the 1225 vector
// instructions of the 6 -> 8 -> 8 -> 16 kernel)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_imax_step(int v)
{
    return max(__builtin_amdgcn_update_dpp((int)0x80000000, v, CTRL, ROW_MASK, 0xf, false), v);
}
__device__ __forceinline__ float half_wave_max_nonneg(float f)
{
    int v = __float_as_int(f);
    v = dpp_imax_step<0xb1, 0xf>(v);
    v = dpp_imax_step<0x4e, 0xf>(v);
    v = dpp_imax_step<0x114, 0xf>(v);
    v = dpp_imax_step<0x118, 0xf>(v);
    v = dpp_imax_step<0x142, 0xa>(v);
    return __int_as_float(v);
}

// max over each 32-lane half; valid in lane 31 / 63
__device__ __forceinline__ float half_wave_max(float v)
{
    v = dpp_max_step<0xb1, 0xf>(v);      // quad_perm:[1,0,3,2]
    v = dpp_max_step<0x4e, 0xf>(v);      // quad_perm:[2,3,0,1]
    v = dpp_max_step<0x114, 0xf>(v);     // row_shr:4
    v = dpp_max_step<0x118, 0xf>(v);     // row_shr:8
    v = dpp_max_step<0x142, 0xa>(v);     // row_bcast:15 into rows 1 and 3 only: halves stay separate
    return v;
}

template <int KIN, int NOUT>
__device__ __forceinline__ void small_layer(const float (&x)[32], float (&y)[32], const float *w, const float *b, bool relu)
{
#pragma unroll
    for (int n = 0; n < NOUT; ++n) y[n] = b[n];
#pragma unroll
    for (int k = 0; k < KIN; ++k) {
#pragma unroll
        for (int n = 0; n < NOUT; ++n) y[n] = fmaf(x[k], w[k * NOUT + n], y[n]);
    }
    if (relu) {                                    // on the bit patterns (relu4's form): fmaxf is llvm.maxnum and costs a canonicalising
#pragma unroll                                     // v_max x, x per value on top of the maximum itself
        for (int n = 0; n < NOUT; ++n) y[n] = __int_as_float(max(__float_as_int(y[n]), 0));
    }
}

constexpr int SMALL_STEPS = 5;                    // window steps of 32 slots held in flight: windows up to 160 slots

template <int CIN, int N1, int N2, int N3>
__global__ __launch_bounds__(ELO_BLOCK) void setconv_small_kernel(const elo_setconv_args a)
{
    constexpr int G = 32, PER_BLOCK = ELO_BLOCK / G, C = CIN - 3;
    constexpr int W1 = 0, B1 = W1 + CIN * N1, W2 = B1 + N1, B2 = W2 + N1 * N2, W3 = B2 + N2, B3 = W3 + N2 * N3,
                  WEND = B3 + N3;
    __shared__ float wsm[WEND];
    __shared__ int slot_hw[PER_BLOCK][G];
    const int tid = threadIdx.x;
    // The kernel is a chain of dependent L2 round trips, so everything whose address is known is requested at once and
    // only consumed later: (1) the weights (into registers; written to LDS just before the MLP), (2) the visiting
    // order of all window steps; then, once the centre is known, (3) every window slot of every step.  Loads are
    // unconditional (clamped indices, selected pointers): a conditional load costs its own s_waitcnt vmcnt(0).
    constexpr int WITER = (WEND + ELO_BLOCK - 1) / ELO_BLOCK;
    float wv[WITER];
#pragma unroll
    for (int u = 0; u < WITER; ++u) {
        const int i = tid + u * ELO_BLOCK < WEND ? tid + u * ELO_BLOCK : WEND - 1;
        const float *src = i < B1 ? a.layers[0].w_plain + (i - W1) : i < W2 ? a.layers[0].bias + (i - B1)
                         : i < B2 ? a.layers[1].w_plain + (i - W2) : i < W3 ? a.layers[1].bias + (i - B2)
                         : i < B3 ? a.layers[2].w_plain + (i - W3) : a.layers[2].bias + (i - B3);
        wv[u] = *src;
    }
    const int g = tid / G, lane = tid % G, shift = (tid & 63) / G * G;
    const elo_group_spec &gs = a.group;
    const int KT = gs.kernel_h * gs.kernel_w, kW = gs.kernel_w, hh = gs.kernel_h / 2, hw2 = gs.kernel_w / 2;
    int off[SMALL_STEPS];                                    // the visiting order first, decoded after the centre load is out
    const int *order = gs.decoded_hw ? gs.decoded_hw : gs.random_hw;        // decoded by the host, or raw (decoded below)
#pragma unroll
    for (int st = 0; st < SMALL_STEPS; ++st) off[st] = order[st * G + lane < KT ? st * G + lane : 0];
    const long total = (long)a.batch * a.npoints;
    const long pt = (long)xcd_tile(blockIdx.x, gridDim.x) * PER_BLOCK + g;
    const bool live = pt < total;
    const long ptc = live ? pt : total - 1;                 // dead groups shadow the last point and store nothing
    int b, n, hc, wc;
    split_point(ptc, a.npoints, b, n);
    if (a.centre_hw) {                                       // one 8-byte load, in the first batch
        const int2 c2 = reinterpret_cast<const int2 *>(a.centre_hw)[ptc];
        hc = c2.x; wc = c2.y;
    } else { hc = n / a.W; wc = n - hc * a.W; }
    if (!gs.decoded_hw) {
#pragma unroll
        for (int st = 0; st < SMALL_STEPS; ++st) off[st] = ((off[st] / kW - hh) << 16) | ((off[st] % kW - hw2) & 0xffff);
    }
    const float *cp = a.xyz1_grid + (((long)b * a.H + hc) * a.W + wc) * 3;
    const float cx = cp[0], cy = cp[1], cz = cp[2];
    const float *grid2 = a.src_xyz + (size_t)b * a.H2 * a.W2 * 3;
    const int base_h = div_stride(hc, gs.stride_h), base_w = div_stride(wc, gs.stride_w);
    RawSlot raw[SMALL_STEPS];
#pragma unroll
    for (int st = 0; st < SMALL_STEPS; ++st) raw[st] = fetch_slot(grid2, a.H2, a.W2, off[st], base_h, base_w, st * G + lane < KT);
    if (live && lane < 3 && a.new_xyz) a.new_xyz[pt * 3 + lane] = lane == 0 ? cx : lane == 1 ? cy : cz;

    // ---- random-k, 32 window slots per step (fused_conv_g.cu:74-152): the walk stops once 32 hits are taken
    const float r2 = gs.distance * gs.distance;
    slot_hw[g][lane] = -1;
    int taken = 0;
    const bool centre_ok = !(pick_max(sq3(cx, cy, cz), ELO_EPS) <= ELO_EPS);
#pragma unroll
    for (int st = 0; st < SMALL_STEPS; ++st) {
        // (uniform) nothing left to decide: the window has no slot here, or both points of the wave have their 32 hits --
        // the slots were fetched ahead in one batch; what is skipped is their judging (dense scans fill up in 2 steps of 5)
        if (st > 0 && (st * G >= KT || __builtin_amdgcn_ballot_w64(taken < G) == 0)) break;
        const Probe pr = judge(raw[st], cx, cy, cz, r2);
        const bool hit = pr.hit && centre_ok && taken < G;          // (st * G < KT is in raw[st].in_grid)
        const unsigned long long mh = group_ballot<G>(hit, shift);
        const int slot = taken + __popcll(mh & ((1ull << lane) - 1ull));
        if (hit && slot < G) slot_hw[g][slot] = pr.hw;
        taken += __popcll(mh);
    }
#pragma unroll
    for (int u = 0; u < WITER; ++u)
        if (tid + u * ELO_BLOCK < WEND) wsm[tid + u * ELO_BLOCK] = wv[u];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const int hw = slot_hw[g][lane];                         // this lane's neighbour (row `lane` of the centre)
    const float m = hw >= 0 ? 1.0f : 0.0f;
    const long cell = hw >= 0 ? ((long)b * a.H2 + (hw >> 16)) * a.W2 + (hw & 0xffff) : 0;   // empty slot: index (0,0,0)
    if (live && gs.idx_out) {
        int *o = gs.idx_out + (pt * G + lane) * 3;
        o[0] = hw >= 0 ? b : 0; o[1] = hw >= 0 ? hw >> 16 : 0; o[2] = hw >= 0 ? hw & 0xffff : 0;
    }
    if (live && gs.mask_out) gs.mask_out[pt * G + lane] = m;

    // ---- gather + centre-subtract + concat [features | xyz difference], then the MLP in registers (:203-222)
    float x[32], y[32];
    const int f16 = a.feat_dtype == ELO_F16;
    const float *sx = a.src_xyz + cell * 3;
    const float s0 = sx[0], s1 = sx[1], s2 = sx[2];
    if (f16) {
        const _Float16 *sf = reinterpret_cast<const _Float16 *>(a.src_feat) + cell * C;
#pragma unroll
        for (int c = 0; c < C; ++c) x[c] = (float)sf[c];
    } else {
        const float *sf = reinterpret_cast<const float *>(a.src_feat) + cell * C;
#pragma unroll
        for (int c = 0; c < C; ++c) x[c] = sf[c];               // all of the row's loads go out before the first use
    }
    x[C + 0] = s0 * m - cx; x[C + 1] = s1 * m - cy; x[C + 2] = s2 * m - cz;
#pragma unroll
    for (int c = 0; c < C; ++c) x[c] *= m;
    __syncthreads();                                          // weights staged (issued first, needed only now)
    small_layer<CIN, N1>(x, y, wsm + W1, wsm + B1, a.layers[0].relu);
    small_layer<N1, N2>(y, x, wsm + W2, wsm + B2, a.layers[1].relu);
    small_layer<N2, N3>(x, y, wsm + W3, wsm + B3, a.layers[2].relu);
    // ---- masked max over the 32 rows (:224-230); lane 31 of the half-wave holds it
    // (the launcher takes this kernel only when the last layer has a ReLU: outputs >= 0, the integer form of the reduction)
#pragma unroll
    for (int c = 0; c < N3; ++c) y[c] = half_wave_max_nonneg(y[c] * m);
    if (live && lane == G - 1) {
#pragma unroll
        for (int c = 0; c < N3; c += 4) feat_store4(a.out, pt * N3 + c, float4{y[c], y[c + 1], y[c + 2], y[c + 3]}, f16);
    }
}

// ================================================================ row-wise MLP over concatenated sources
// stage 1: columns [0, w0 + w1 + w2) = the sources; its last layer writes `out` to HBM straight from the accumulators
// and, with a second stage, to columns [0, N) of the tile; stage 2: [out (N) | before | after] -> layers2 -> out2.
// The tile's rows are the global rows [first, min(first + TILE, end)).  keep_final: the LAST layer's output also stays in the tile,
// columns [0, N), as plain floats (the values as stored: fp16 storage rounds them) -- the input of mlp_sv_kernel's reduction.
template <int TILE, int MODE>
__device__ __forceinline__ void mlp_tile(const elo_mlp_args &a, float *act, const int S, const long first, const long end,
                                         const bool keep_final, unsigned &bad)
{
    const int tid = ftid(), f16 = a.feat_dtype == ELO_F16;
    auto row_of = [&](int row) { return first + row < end ? first + row : -1L; };
    auto all = [](int) { return true; };
    const int w0 = a.src_width[0], w1 = a.n_sources > 1 ? a.src_width[1] : 0, w2 = a.n_sources > 2 ? a.src_width[2] : 0;
    const int CT = w0 + w1 + w2, CTp = ceil16(CT);
    if (seg_ok(a.src[0], w0, TILE, f16) && (w1 == 0 || seg_ok(a.src[1], w1, TILE, f16)) && (w2 == 0 || seg_ok(a.src[2], w2, TILE, f16))) {
        uint4 r0[SEG_ITEMS], r1[SEG_ITEMS], r2[SEG_ITEMS];
        seg_load<TILE>(r0, a.src[0], w0, f16, row_of);
        if (w1) seg_load<TILE>(r1, a.src[1], w1, f16, row_of);
        if (w2) seg_load<TILE>(r2, a.src[2], w2, f16, row_of);
        seg_store<TILE, MODE>(act, S, 0, w0, f16, r0, row_of, all, bad);
        if (w1) seg_store<TILE, MODE>(act, S, w0, w1, f16, r1, row_of, all, bad);
        if (w2) seg_store<TILE, MODE>(act, S, w0 + w1, w2, f16, r2, row_of, all, bad);
        zero_cols<TILE, MODE>(act, S, CT, CTp);
    } else {
        for (int e = tid; e < TILE * CTp; e += FUSED_BLOCK) {
            const int row = small_div(e, CTp), ch = e - row * CTp;
            const long gr = first + row;
            float v = 0.0f;
            if (gr < end && ch < CT) {
                v = ch < w0 ? feat_load(a.src[0], gr * w0 + ch, f16)
                  : ch < w0 + w1 ? feat_load(a.src[1], gr * w1 + (ch - w0), f16)
                                 : feat_load(a.src[2], gr * w2 + (ch - w0 - w1), f16);
            }
            act_put<MODE>(act, row * S + ch, v, bad);
        }
    }
    __syncthreads();
    const int N = a.layers[a.n_layers - 1].N;
    const bool two = a.n_layers2 > 0;
    for (int l = 0; l < a.n_layers; ++l)
        dense<TILE, MODE>(act, S, 0, a.layers[l],
                          l == a.n_layers - 1 ? LayerOut{two || keep_final ? 0 : -1, !two && keep_final ? 1 : 0, a.out, first, end, f16} : to_tile(0), bad);
    if (!two) return;
    // ---- second stage: [out | before | after] -> layers2 -> out2       (`out` already sits at columns [0, N))
    const int wb = a.w_before, wa = a.w_after, CT2 = N + wb + wa;
    if ((N & 3) == 0 && (wb == 0 || seg_ok(a.before, wb, TILE, f16)) && (wa == 0 || seg_ok(a.after, wa, TILE, f16))) {
        uint4 rb[SEG_ITEMS], ra[SEG_ITEMS];
        if (wb) seg_load<TILE>(rb, a.before, wb, f16, row_of);
        if (wa) seg_load<TILE>(ra, a.after, wa, f16, row_of);
        if (wb) seg_store<TILE, MODE>(act, S, N, wb, f16, rb, row_of, all, bad);
        if (wa) seg_store<TILE, MODE>(act, S, N + wb, wa, f16, ra, row_of, all, bad);
    } else {
        for (int e = tid; e < TILE * (wb + wa); e += FUSED_BLOCK) {
            const int row = small_div(e, wb + wa), ch = e - row * (wb + wa);
            const long gr = first + row;
            const long grc = gr < end ? gr : end - 1;
            const float v = ch < wb ? feat_load(a.before, grc * wb + ch, f16) : feat_load(a.after, grc * wa + (ch - wb), f16);
            act_put<MODE>(act, row * S + N + ch, gr < end ? v : 0.0f, bad);
        }
    }
    zero_cols<TILE, MODE>(act, S, CT2, ceil16(CT2));
    __syncthreads();
    for (int l = 0; l < a.n_layers2; ++l)
        dense<TILE, MODE>(act, S, 0, a.layers2[l],
                          l == a.n_layers2 - 1 ? LayerOut{keep_final ? 0 : -1, keep_final ? 1 : 0, a.out2, first, end, f16} : to_tile(0), bad);
}

template <int TILE, int MODE>
__global__ __launch_bounds__(FUSED_BLOCK, TILE == 32 ? ELO_TILE32_WAVES : ELO_TILE_WAVES) void mlp_kernel(const JobPair<elo_mlp_args> jobs, const int S)
{
    const elo_mlp_args &a = jobs.job[blockIdx.y];
    extern __shared__ __align__(16) float lds[];
    unsigned bad = 0;
    {   // side job (elo_mlp_args.clear_*, job 0's): every workgroup of the launch clears its share of a later projection's buffers
        const elo_mlp_args &j0 = jobs.job[0];
        clear_projection(ProjectionClear{j0.clear_scratch, j0.clear_xyz, (unsigned *)j0.clear_feat, j0.clear_cells,
                                         j0.feat_dtype == ELO_F16 ? j0.clear_C / 2 : j0.clear_C, j0.clear_images});
    }
    const long first = (long)xcd_tile(blockIdx.x, gridDim.x) * TILE;
    if (first >= a.rows) return;
    mlp_tile<TILE, MODE>(a, lds, S, first, a.rows, false, bad);
    report_violations<MODE>(bad, a.range_counter);
}

// The row-wise MLP(s) that end in softmax_valid's inputs, WITH the first half of softmax_valid (model_util.py:319-343: a
// softmax over the valid points of every channel's logits, the weighted sum of the features): a tile's final rows are in
// LDS when its last layer is done, so the workgroup reduces them to one (max, denominator, weighted sum) triple per channel
// -- what softmax_valid_partial_kernel computes from HBM in a launch of its own (7-9 us of a batch-1 level) -- and the pose
// head merges the tiles' triples instead of that launch's slices (elo_pose_head_args.ready_parts).
//   PAIR: job 0 produces the logits (the embedding-mask branch, `weight`), job 1 the features (`predict`): the two
//     4-wave groups of a 512-thread workgroup run them SIDE BY SIDE on the same rows (twins back to back on one tile were
//     measured slower than the paired launch: a level's tail is a latency chain), each on its own tile;
//   single (the coarse level: the features are an existing tensor, sv_feature): staged into LDS at the start.
// Tiles never straddle two batch elements (tiles_pb per element); the triple of tile t of element b goes to slot
// (b, t) of sv_scratch -- the layout of softmax_valid_partial_kernel's scratch.
template <int TILE, int MODE, bool PAIR>
__global__ __launch_bounds__(PAIR ? 2 * FUSED_BLOCK : FUSED_BLOCK, TILE == 32 ? 4 : ELO_TILE_WAVES)      // (32 rows: the held registers do not fit mlp_kernel's 96)
void mlp_sv_kernel(const JobPair<elo_mlp_args> jobs, const int S, const int tiles_pb)
{
    constexpr int C = 64;
    const int group = PAIR ? __builtin_amdgcn_readfirstlane(threadIdx.x >> 8) : 0;
    const elo_mlp_args &j0 = jobs.job[0];
    const elo_mlp_args &a = jobs.job[group];
    extern __shared__ __align__(16) float lds[];
    const int tile_words = TILE * S;
    float *act = lds + group * tile_words;
    float *staged = lds + (PAIR ? 2 : 1) * tile_words;        // single: the tile's feature rows [TILE][64]
    unsigned bad = 0;
    clear_projection(ProjectionClear{j0.clear_scratch, j0.clear_xyz, (unsigned *)j0.clear_feat, j0.clear_cells,
                                     j0.feat_dtype == ELO_F16 ? j0.clear_C / 2 : j0.clear_C, j0.clear_images});
    const unsigned tile = xcd_tile(blockIdx.x, gridDim.x);
    const int b = (int)(tile / (unsigned)tiles_pb), t = (int)(tile - (unsigned)b * (unsigned)tiles_pb);
    const long first = (long)b * j0.sv_npoints + (long)t * TILE, end = (long)(b + 1) * j0.sv_npoints;
    const int f16 = j0.feat_dtype == ELO_F16;
    // Requested now, CONSUMED after the MLP (nothing below waits for them before the tile's own loads are out):
    //   the tile's points -- lane r < TILE of the first wave holds row r's xyz; a row is valid unless all three are exactly 0
    //   (model_util.py:325), handed to the reduction as one ballot;
    //   single: the tile's feature rows, 16 bytes (8 for fp16 storage) per thread and 16 rows, staged into LDS behind the MLP.
    const int vr = threadIdx.x < TILE ? (int)threadIdx.x : 0;
    const long vrow = first + vr < end ? first + vr : end - 1;
    const float px = j0.sv_xyz[vrow * 3 + 0], py = j0.sv_xyz[vrow * 3 + 1], pz = j0.sv_xyz[vrow * 3 + 2];
    constexpr int STAGE = PAIR ? 1 : TILE / 16;
    uint4 held[STAGE];
    if constexpr (!PAIR) {
#pragma unroll
        for (int u = 0; u < STAGE; ++u) {
            const int e = (int)threadIdx.x + u * FUSED_BLOCK, fr = e >> 4, q = e & 15;
            const long grow = first + fr < end ? first + fr : end - 1;
            if (f16) { const uint2 h = reinterpret_cast<const uint2 *>(j0.sv_feature)[grow * (C / 4) + q]; held[u] = uint4{h.x, h.y, 0u, 0u}; }
            else held[u] = reinterpret_cast<const uint4 *>(j0.sv_feature)[grow * (C / 4) + q];
        }
    }
    mlp_tile<TILE, MODE>(a, act, S, first, end, true, bad);       // (ends with the last layer's barrier: both tiles are complete)
    report_violations<MODE>(bad, a.range_counter);
    if constexpr (!PAIR) {
#pragma unroll
        for (int u = 0; u < STAGE; ++u) {
            const int e = (int)threadIdx.x + u * FUSED_BLOCK, fr = e >> 4, q = e & 15;
            float4 v;
            if (f16) {
                const half2v h0 = __builtin_bit_cast(half2v, held[u].x), h1 = __builtin_bit_cast(half2v, held[u].y);
                v = float4{(float)h0.x, (float)h0.y, (float)h1.x, (float)h1.y};
            } else {
                v = __builtin_bit_cast(float4, held[u]);
            }
            *reinterpret_cast<float4 *>(staged + fr * C + 4 * q) = v;
        }
        __syncthreads();
    }
    if (threadIdx.x >= C) return;
    const unsigned long long okm = __ballot((threadIdx.x < TILE) & (first + vr < end) & !((px == 0.0f) & (py == 0.0f) & (pz == 0.0f)));
    const int c = threadIdx.x;
    const float *logit = lds + c, *value = PAIR ? lds + tile_words + c : staged + c;
    const int vs = PAIR ? S : C;
    float M = -INFINITY, D = 0.0f, A = 0.0f;
#pragma unroll
    for (int r0 = 0; r0 < TILE; r0 += 16) {                     // 16 rows at a time: softmax_valid_partial_kernel's step
        float l[16], v[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) { l[r] = logit[(r0 + r) * S]; v[r] = value[(r0 + r) * vs]; }
        const unsigned ok = (unsigned)(okm >> r0) & 0xffffu;
        float bm = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) bm = (ok >> r & 1u) ? fmaxf(bm, l[r]) : bm;
        if (bm > -INFINITY) {
            float d16 = 0.0f, a16 = 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float e = (ok >> r & 1u) ? exp_acc(l[r] - bm) : 0.0f;
                d16 += e;
                a16 += e * v[r];
            }
            const float m2 = fmaxf(M, bm), s0 = exp_acc(M - m2), s1 = exp_acc(bm - m2);      // exp_acc(-inf) = 0 on the first step
            D = D * s0 + d16 * s1;
            A = A * s0 + a16 * s1;
            M = m2;
        }
    }
    float *sc = j0.sv_scratch;
    const size_t n = (size_t)(gridDim.x / (unsigned)tiles_pb) * ELO_SV_MAX_PARTS * C;      // (batch, ELO_SV_MAX_PARTS, C) x {max, den, acc}
    const size_t at = ((size_t)b * ELO_SV_MAX_PARTS + t) * C + c;
    sc[at] = M; sc[n + at] = D; sc[2 * n + at] = A;
}

// ================================================================ cost volume, stage 1
// LDS columns: [0,128) = X (CV chain, later [x | enc]); F = 128: [F, F+C) feat1, [F+C, F+2C) feat2[idx]*m,
// [F+2C, F+2C+10) the geometry [p, q, q-p, |q-p|] and zeros up to the 16-k block boundary (CV_0's input rows are
// ordered that way by the host; the geometry block alone is CV_xyz's input and lives until then); sum_CV_0 then
// writes its 128 outputs to [64,192) -- over enc, its own input: the layer's barrier separates the reads from the
// write -- so x at [0,64) survives for the pooling and the tile is max(192, 128 + 2C + 16) columns wide instead of 256
// (27.5 KB instead of 35.7 KB at 32 rows: a fifth workgroup per CU).
__host__ __device__ __forceinline__ int cv1_feat_cols(int C)
{
    const int kp = (10 + 2 * C + 15) & ~15;       // CV_0's padded K
    return kp > 2 * C + 16 ? kp : 2 * C + 16;     // and CV_xyz's 16-k block at 2C
}

// A layer descriptor read from the kernel-argument segment AT THE LAYER, through a pointer the compiler cannot see
// through: with the by-value argument block it hoists the s_loads of all six descriptors (72 scalar registers) to the
// kernel's entry and keeps them live across the in-kernel grouping, whose own scalars then do not fit (67-68 SGPRs
// spilled to VGPR lanes in cv1_kernel / cv1_setconv_kernel).  Valid where the elo_cv1_args block is the kernel's FIRST
// parameter (all three kernels below).
__device__ __forceinline__ elo_dense kernarg_dense(size_t byte_offset)
{
    static_assert(sizeof(elo_dense) == 48, "three 16-byte scalar loads");
    typedef const __attribute__((address_space(4))) char *KernargBytes;
    typedef unsigned int Quad __attribute__((ext_vector_type(4)));
    typedef const __attribute__((address_space(4))) Quad *KernargQuads;
    KernargBytes p = (KernargBytes)__builtin_amdgcn_kernarg_segment_ptr() + byte_offset;
    asm volatile("" : "+s"(p));
    KernargQuads q = reinterpret_cast<KernargQuads>(p);
    struct Raw { Quad a, b, c; } raw = {q[0], q[1], q[2]};
    return __builtin_bit_cast(elo_dense, raw);
}
#define CV1_LAYER(field) kernarg_dense(offsetof(elo_cv1_args, field))

// GROUP = false: an instance without the in-kernel select-k (the neighbours come from a.idx / a.mask, written by
// elo_fused_conv_select_k_dense in front of this launch): the window registers and the grouping's scalars are gone
template <int TILE, int MODE, bool GROUP = true>
__device__ __forceinline__ void cv1_tile(const elo_cv1_args &a, const int S, float *lds, unsigned block, unsigned nblocks)
{
    float *act = lds;
    const TileMeta meta = tile_meta(lds, TILE, S);
    const int tid = threadIdx.x;
    const int K = a.K, P = TILE / K, C = a.C, f16 = a.feat_dtype == ELO_F16;
    unsigned bad = 0;
    const long total_points = (long)a.batch * a.npoints;
    const long first_point = (long)xcd_tile(block, nblocks) * P;
    if (first_point >= total_points) return;
    CV1_STAMP(0);
    if (GROUP && a.group.random_hw) {                 // select-k of frame 2 around every warped frame-1 pixel (:49-51)
        // the visiting order and select-k's scratch live IN the activation tile, which is idle until the gather that
        // follows the grouping's closing barrier: 3.8 KB less LDS per workgroup (a sixth one per CU)
        int *lds_off = reinterpret_cast<int *>(act);
        unsigned *scratch = reinterpret_cast<unsigned *>(lds_off + ((a.group.kernel_h * a.group.kernel_w + 3) & ~3));   // 16-byte aligned
        group_tile<TILE, true>(meta, lds_off, scratch, a.group, first_point, total_points, a.npoints, P, K, a.xyz1, a.H2,
                               a.W2, nullptr, a.xyz2, a.H2, a.W2, nullptr);
    } else {
        load_meta<TILE>(meta, first_point, total_points, P, K, a.idx, a.mask, a.H2, a.W2);
        __syncthreads();
    }
    CV1_STAMP(1);
    const int F = 128, G = F + 2 * C;                                                             // :54-66
    {
        auto cell_of = [&](int row) { return (long)meta.cell[row]; };
        auto centre_of = [&](int row) { return meta.cell[row] >= 0 ? first_point + small_div(row, K) : -1L; };
        uint4 r1[SEG_ITEMS], r2[SEG_ITEMS];
        seg_load<TILE>(r1, a.feat1, C, f16, centre_of);
        seg_load<TILE>(r2, a.feat2, C, f16, cell_of);
        GeoRow gr;
        const bool grow = tid < TILE;
        const int gcell = grow ? meta.cell[tid] : -1;
        if (grow) gr = geo_load(a.xyz1 + (gcell >= 0 ? first_point + small_div(tid, K) : 0) * 3, a.xyz2 + (long)(gcell >= 0 ? gcell : 0) * 3);
        seg_store<TILE, MODE>(act, S, F, C, f16, r1, centre_of, [](int) { return true; }, bad);
        seg_store<TILE, MODE>(act, S, F + C, C, f16, r2, cell_of, [&](int row) { return meta.mask[row] != 0.0f; }, bad);
        if (grow) geo_store<MODE>(act, tid * S + G, gr, meta.mask[tid], gcell >= 0, bad);           // columns [G, G+16)
        zero_cols<TILE, MODE>(act, S, G + 16, F + cv1_feat_cols(C));
    }
    Pre<Sub<TILE, 128>::TPW> p128;
    Pre<Sub<TILE, 64>::TPW> p64;
    const elo_dense cv0 = CV1_LAYER(cv0);
    prefetch<TILE, Sub<TILE, 128>::TPW, MODE>(cv0, p128);           // in flight while the barrier drains the gather
    __syncthreads();
    CV1_STAMP(2);
    const elo_dense cv1 = CV1_LAYER(cv1);
    dense_pf<TILE, 128, 64, MODE>(act, S, F, to_tile(0), cv0, p128, &cv1, &p64, bad);          // feat_cat -> 128          :72-76
    CV1_STAMP(3);
    const elo_dense cv2 = CV1_LAYER(cv2);
    dense_pf<TILE, 64, 64, MODE>(act, S, 0, to_tile(0), cv1, p64, &cv2, &p64, bad);            // -> 64 (in place)
    CV1_STAMP(4);
    const elo_dense cv_xyz = CV1_LAYER(cv_xyz);
    dense_pf<TILE, 64, 64, MODE>(act, S, 0, to_tile(0), cv2, p64, &cv_xyz, &p64, bad);         // -> 64 = x   (values of the pooling)
    CV1_STAMP(5);
    const elo_dense sum_cv0 = CV1_LAYER(sum_cv0);
    dense_pf<TILE, 64, 128, MODE>(act, S, G, to_tile(64), cv_xyz, p64, &sum_cv0, &p128, bad);  // xyz_cat -> enc at [64,128)   :79-82
    CV1_STAMP(6);
    const elo_dense sum_cv1 = CV1_LAYER(sum_cv1);
    dense_pf<TILE, 128, 64, MODE>(act, S, 0, to_tile(64), sum_cv0, p128, &sum_cv1, &p64, bad); // [x | enc] -> 128 at [64,192): over enc, x stays  :84-90
    CV1_STAMP(7);
    dense_pf<TILE, 64, 0, MODE>(act, S, 64, to_pool(64), sum_cv1, p64, nullptr, nullptr, bad);   // -> 64 logits at [64,128) (plain fp32)
    CV1_STAMP(8);
    pool_masked_softmax(act, S, 64, 0, meta, P, K, first_point, total_points, a.out, f16);   // :92-98
    CV1_STAMP(9);
    report_violations<MODE>(bad, a.range_counter);
}

template <int TILE, int MODE>
__global__ __launch_bounds__(FUSED_BLOCK, ELO_CV1_WAVES) void cv1_kernel(const elo_cv1_args a, const int S)
{
    extern __shared__ __align__(16) float lds[];
    cv1_tile<TILE, MODE>(a, S, lds, blockIdx.x, gridDim.x);
}

#ifndef ELO_CV1_META_WAVES
#define ELO_CV1_META_WAVES 5               // (6 spills 24 registers in the 32-row instance)
#endif
template <int TILE, int MODE>
__global__ __launch_bounds__(FUSED_BLOCK, ELO_CV1_META_WAVES) void cv1_meta_kernel(const elo_cv1_args a, const int S)
{
    extern __shared__ __align__(16) float lds[];
    cv1_tile<TILE, MODE, false>(a, S, lds, blockIdx.x, gridDim.x);
}

// ---- heterogeneous launch: cost-volume stage 1 and one or two set-conv jobs in ONE grid ---------------------------------
// Inside a refinement level the cost volume (stage 1 -> stage 2) and the two set-upconvs only share INPUTS
// (pwclo_model.py:242-250), and at the head of the pyramid the initial cost volume (:170) does not need the layer-3
// set-conv (:138): as consecutive launches of a batch-1 forward each of them leaves most CUs idle and pays its own
// ≈4.5 us launch boundary (forked streams inside the hipGraph cost more than they gain).  Here the first
// n_cv workgroups run cost-volume tiles and the rest set-conv tiles: one launch, both branches in flight together.
struct SideJobs {
    elo_setconv_args job[2];
    int njobs, S, blocks_per_job;
};

template <int TILE_CV, int TILE_SC, int MODE>
__global__ __launch_bounds__(FUSED_BLOCK, ELO_CV1_WAVES) void cv1_setconv_kernel(const elo_cv1_args a, const int S, const unsigned n_cv,
                                                                     const SideJobs side)
{
    extern __shared__ __align__(16) float lds[];
    if (blockIdx.x < n_cv) {
        cv1_tile<TILE_CV, MODE>(a, S, lds, blockIdx.x, n_cv);
    } else {
        const unsigned r = blockIdx.x - n_cv;
        const unsigned j = r / side.blocks_per_job;                    // uniform: one scalar index into the kernarg block
        setconv_tile<TILE_SC, MODE>(side.job[j], side.S, lds, r - j * side.blocks_per_job, side.blocks_per_job);
    }
}

// ================================================================ cost volume, stage 1: register-resident chain
// The tile kernel above runs a layer as "four waves share 32 rows, each owns a quarter of the columns": every layer
// boundary is an LDS round trip and two workgroup barriers, and EVERY 32-row tile pulls all 172 KB of packed weights
// through the CU's vector L1 (64 B/clk): at batch 8 that is 3.6 MB per CU and launch -- 56 k cycles of L1 time against
// 46 k cycles of matrix-core time -- which is what kept that kernel at 17 % matrix utilisation whatever was done to its
// instruction mix (round 2's phase table; round 3 measured the same wall with a wave that streamed W on its own).
// Here ONE WAVE owns 16 rows (of the 128 rows = P points x K neighbour slots of its workgroup) and ALL columns, and the chain stays in its registers: the
// transposed product D^T = W^T A^T leaves lane (i16, kq) with channels cb*16 + 4kq .. +3 of row i16 -- exactly the quad
// that lane feeds as the B operand of the NEXT layer's k-block cb (the pair layout of WFrag takes quad kq of blocks 2p and
// 2p+1) -- so a layer's epilogue (bias is the accumulator's initial value, ReLU, hi/lo split) writes the next layer's
// operands in place: no activation ever touches LDS.  The eight waves of a workgroup walk the SAME weight stream, so W
// crosses the L1 once per 128 rows: each wave fetches one 1 KB chunk of every 8 KB "superstep" (two k-steps of two column
// blocks) into a register three supersteps ahead, drops it into a two-slot LDS ring, and all eight read their matrix
// operands from there (LDS: 256 B/clk); one workgroup barrier per superstep of 12 matrix instructions per wave, waiting on
// LDS only (the global loads stay in flight across it).  16 rows per wave keep the kernel at <= 128 VGPRs: four waves
// per SIMD, whose matrix bursts and epilogues interleave.  LDS is used once more, after the chain, to transpose
// (logits, values) for the softmax pooling over the K rows of a point (the scratch aliases the ring).
// Neighbours come from a.idx / a.mask.  Same arithmetic in the same order as cv1_tile: the outputs are equal bit for bit
// (tests/test_ops_gpu.py::test_register_resident_cost_volume_equals_the_tile_kernel).
#ifndef ELO_DENSE_F32
constexpr int RR_PITCH = 36;                       // words per row of the pooling scratch (32 channels + 4: the pooling runs in two halves)
constexpr int RR_WAVES = 8;                        // waves per workgroup: 16 rows each, one W stream for all 128 rows
constexpr int RR_ROWS = RR_WAVES * 16;
constexpr int RR_SLOT_U4 = 512;                    // a ring slot: 8 KB = 8 chunks of 64 x 16 bytes, one chunk per wave
constexpr int RR_NSLOT = 2;                        // ring slots: superstep S is read while S + 1 is being written
#ifndef ELO_RR_STAGE
#define ELO_RR_STAGE 3                             // supersteps a wave's global W load runs ahead of its LDS write
#endif
constexpr int RR_STAGE = ELO_RR_STAGE;
constexpr size_t RR_LDS_BYTES = sizeof(float) * (2 * RR_ROWS * RR_PITCH + RR_ROWS);   // pooling scratch (36.5 KB); >= the 16 KB ring it aliases

struct RrW { WPair w[2]; };                        // one k-step of two column blocks (a tail travels in .hi)

// workgroup barrier that waits for this wave's LDS traffic only: __syncthreads() also drains vmcnt, i.e. the W loads
// that are meant to stay in flight across it
__device__ __forceinline__ void rr_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// The W stream of a register-resident chain.  PLAN: NL layers with KP[l] k-pairs (+ TL[l]: a 16-k tail) and NCB[l]
// column blocks; steps run layer by layer, pass by pass (two column blocks), k-step by k-step; two steps = a superstep =
// one 8 KB ring slot of 8 chunks, chunk w = (step 2S + (w >> 2), column block (w >> 1) & 1, hi / lo half w & 1) carried
// by wave w.  Every layer has an even number of steps, so a superstep never straddles two layers.  All arguments named
// S / g are constants after unrolling: the layer look-ups fold.
template <class PLAN, int MODE = MODE_SPLIT>
struct RrStream {
    const float *w[PLAN::NL];                        // packed weights of the layers
    uint4 *ring;
    int wave, lane;
    uint4 st[RR_STAGE];                              // this wave's chunks on their way from L2 to the ring

    static __device__ __forceinline__ constexpr int steps(int l) { return PLAN::NCB[l] / 2 * (PLAN::KP[l] + PLAN::TL[l]); }
    static __device__ __forceinline__ constexpr int first(int l) { int g = 0; for (int i = 0; i < l; ++i) g += steps(i); return g; }
    static constexpr int total() { int g = 0; for (int i = 0; i < PLAN::NL; ++i) g += PLAN::NCB[i] / 2 * (PLAN::KP[i] + PLAN::TL[i]); return g; }

    __device__ __forceinline__ void issue(int S)                          // request this wave's chunk of superstep S
    {
        if (2 * S >= total()) return;
        // MODE_HALF: a pair chunk has no lo half, the odd waves have nothing to stage and issue no load.  (Round 3 shipped
        // them fetching their partner's kilobyte and dropping it, because without a load on their path "one or two waves of a
        // few workgroups gave wrong rows".  The loads were never the cause: that build's schedule put a 16-k tail MFMA two
        // wait states behind the pair MFMA whose accumulator it continues -- mfma_shape_guard, above mma_tail.
        // tools/micro/patches/elo_fused_experiments.patch + -DELO_RR_EQUAL_LOADS restores round 3's form for tools/rr_bisect.sh.)
        if (MODE == MODE_HALF && (wave & 1)) return;
        const int g = 2 * S;
        int l = 0;
#pragma unroll
        for (int i = 1; i < PLAN::NL; ++i) l = g >= first(i) ? i : l;
        const int kp = PLAN::KP[l], tl = PLAN::TL[l], spp = kp + tl, KS = 2 * kp + tl;
        constexpr int BB = WFrag<MODE>::BLOCK_BYTES;
        constexpr int LO = MODE == MODE_HALF ? 0 : 1024;                  // the lo8 half of a pair chunk (MODE_HALF: there is none)
        const int slA = g - first(l), slB = slA + 1;
        const int passA = slA / spp, kA = slA - passA * spp, passB = slB / spp, kB = slB - passB * spp;
        const bool tailA = kA >= kp, tailB = kB >= kp;
        const int offA = 2 * passA * KS * BB + (tailA ? (KS - 1) * BB : kA * 2 * BB);
        const int offB = 2 * passB * KS * BB + (tailB ? (KS - 1) * BB : kB * 2 * BB);
        const int second = wave >> 2;                                     // (scalars)
        const bool is_tail = second ? tailB : tailA;
        const int off = (second ? offB : offA) + ((wave >> 1) & 1) * KS * BB + ((wave & 1) && !is_tail ? LO : 0);
        if (MODE == MODE_HALF && is_tail) {                                // a 16-k tail is 8 bytes per lane there
            const uint2 v = __builtin_bit_cast(uint2, __builtin_amdgcn_raw_buffer_load_b64(weight_rsrc(w[l]), lane * 8u, off, 0));
            st[S % RR_STAGE] = uint4{v.x, v.y, 0u, 0u};
            return;
        }
        st[S % RR_STAGE] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(weight_rsrc(w[l]), lane * 16u, off, 0));
    }
    __device__ __forceinline__ void commit(int S)                         // ... into chunk `wave` of slot S % RR_NSLOT
    {
        if (2 * S >= total() || (MODE == MODE_HALF && (wave & 1))) return;
        ring[(S % RR_NSLOT) * RR_SLOT_U4 + wave * 64 + lane] = st[S % RR_STAGE];
    }
    __device__ __forceinline__ void fetch(int g, RrW &dst) const          // step g's operands: chunks (g % 2) * 4 + 2t (+ 1: lo)
    {
        if (g >= total()) return;
        const uint4 *slot = ring + ((g / 2) % RR_NSLOT) * RR_SLOT_U4 + (g % 2) * 256 + lane;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            dst.w[t].hi = slot[t * 128];
            if constexpr (MODE != MODE_HALF) dst.w[t].lo = slot[t * 128 + 64];
            else dst.w[t].lo = uint4{0u, 0u, 0u, 0u};
        }
    }
    // end of superstep S: superstep S + 1 goes into the slot S - 1 used (every wave read it before the last barrier)
    __device__ __forceinline__ void advance(int S)
    {
        commit(S + 1);
        issue(S + 1 + RR_STAGE);
        rr_barrier();
    }
    __device__ __forceinline__ void start()                               // the first RR_STAGE supersteps requested
    {
#pragma unroll
        for (int S = 0; S < RR_STAGE; ++S) issue(S);
    }
    __device__ __forceinline__ void prime() { commit(0); issue(RR_STAGE); }   // superstep 0 into the ring (then: rr_barrier())
};

// One layer on the wave's 16 rows.  in[p]: operand pairs, tail: the 16-k tail block (TAIL); NCB column blocks, two per
// pass, a pass = KP pair steps (+ the tail step).  The W stream of the WHOLE kernel is one sequence of steps (layer by
// layer, pass by pass), two steps = one superstep = one ring slot.  Every step reads its operands from the ring (fetch:
// the slot is complete since the last barrier); at the end of an odd step the wave moves its chunk of the NEXT superstep
// from its staging register into the other slot, requests a later one (advance), and joins the barrier.  G0 = this
// layer's first step.  bias0: the bias quads of pass 0 on entry, of `next_bias`'s pass 0 on exit.  The biases of later passes
// come from the workgroup's LDS table (rr_stage_biases): as global loads one pass ahead they were a dependent L2 round trip
// PER PASS -- a pass is 100-1000 cycles of work, the round trip ~700 under load: tools/rr_clock.sh showed ~500 + 250 k cycles
// for a pass of k steps whatever the MFMA count, and a build without any MFMA ran as long.
// emit(pass, t, acc): epilogue of column block 2*pass + t.
// XMASK bit k: pair k of `in` holds exact fp16 values (lo == 0); XTAIL: so does the tail block -- see mma_pair's A_EXACT
// init(cb, b): the accumulators' initial value for column block cb, given the block's bias quad b (default: the bias itself).
struct RrBiasInit { __device__ __forceinline__ f32x4 operator()(int, const float4 b) const { return f32x4{b.x, b.y, b.z, b.w}; } };
template <int KP, bool TAIL, int NCB, int G0, int MODE = MODE_SPLIT, unsigned XMASK = 0u, bool XTAIL = false, class Fetch, class Advance, class Emit,
          class Init = RrBiasInit>
__device__ __forceinline__ void rr_layer(const ActPair (&in)[KP > 0 ? KP : 1], const uint4 &tail, const float *bias,
                                         const float *next_bias, float4 (&bias0)[2], Fetch fetch, Advance advance, Emit emit, Init init = Init())
// (L.bias / next_bias: this layer's and the next one's biases IN THE LDS TABLE, see rr_stage_biases)
{
    constexpr int PASSES = NCB / 2, SPP = KP + (TAIL ? 1 : 0);
    static_assert(G0 % 2 == 0 && (PASSES * SPP) % 2 == 0, "a superstep never straddles two layers");
    const int kq = (threadIdx.x & 63) >> 4;
    f32x4 acc[1][2];
    float4 bnext[2] = {bias0[0], bias0[1]};
#pragma unroll
    for (int pass = 0; pass < PASSES; ++pass) {
#pragma unroll
        for (int t = 0; t < 2; ++t) acc[0][t] = init(2 * pass + t, bnext[t]);
        if (pass + 1 < PASSES) {                                    // the next pass's bias quads, a pass ahead
#pragma unroll
            for (int t = 0; t < 2; ++t) bnext[t] = *reinterpret_cast<const float4 *>(bias + (2 * (pass + 1) + t) * 16 + 4 * kq);
        } else if (next_bias) {
#pragma unroll
            for (int t = 0; t < 2; ++t) bnext[t] = *reinterpret_cast<const float4 *>(next_bias + t * 16 + 4 * kq);
        }
#pragma unroll
        for (int k = 0; k < SPP; ++k) {
            const int g = G0 + pass * SPP + k;
            RrW cur;
            fetch(g, cur);                                           // (the other waves of the SIMD cover the LDS round trip)
            if (k < KP) {
                const ActPair a1[1] = {in[k < KP ? k : 0]};
                const WPair w2[2] = {cur.w[0], cur.w[1]};
                if ((XMASK >> (k & 31)) & 1u) mma_pair<MODE, 2, 1, 2, true>(acc, 0, a1, w2);     // (k is a constant after unrolling)
                else mma_pair<MODE, 2, 1, 2>(acc, 0, a1, w2);
            } else {
                const uint4 a1[1] = {tail};
                const uint4 w2[2] = {cur.w[0].hi, cur.w[1].hi};
                mma_tail<MODE, 2, 1, 2, (KP > 0), XTAIL>(acc, 0, a1, w2);
            }
            if (g % 2 == 1) advance(g / 2);
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) emit(pass, t, acc[0][t]);
    }
    bias0[0] = bnext[0]; bias0[1] = bnext[1];
}

// max(x, 0) or x as ONE integer instruction and no branch: read as a signed integer every negative float (and -0) is
// negative, every positive one keeps its order, so ReLU is max(bits, 0); "no ReLU" is max(bits, INT_MIN).  (fmaxf is
// llvm.maxnum: the compiler puts a canonicalising v_max x, x in front of it -- 256 extra instructions per 32 rows.)
__device__ __forceinline__ float4 relu4(const f32x4 a, int relu)
{
    const int floor = relu ? 0 : (int)0x80000000;
    auto mx = [&](float x) { return __int_as_float(max(__float_as_int(x), floor)); };
    return float4{mx(a[0]), mx(a[1]), mx(a[2]), mx(a[3])};
}

// the quad of column block 2*pass + t goes into half `t` of pair `pass` of the next layer's operands
__device__ __forceinline__ void put_quad(ActPair &dst, int t, const uint4 q)
{
    if (t == 0) { dst.hi.x = q.x; dst.hi.y = q.y; dst.lo.x = q.z; dst.lo.y = q.w; }
    else { dst.hi.z = q.x; dst.hi.w = q.y; dst.lo.z = q.z; dst.lo.w = q.w; }
}

// hi + lo of an operand quad as four floats (what act_get reads back in the tile kernel)
__device__ __forceinline__ float4 quad_value(unsigned h01_, unsigned h23_, unsigned l01_, unsigned l23_)
{
    const half2v h01 = __builtin_bit_cast(half2v, h01_), h23 = __builtin_bit_cast(half2v, h23_);
    const half2v l01 = __builtin_bit_cast(half2v, l01_), l23 = __builtin_bit_cast(half2v, l23_);
    return float4{(float)h01.x + (float)l01.x, (float)h01.y + (float)l01.y, (float)h23.x + (float)l23.x, (float)h23.y + (float)l23.y};
}

// The tail of a register-resident chain: masked softmax over the K rows of a point of the 64 logits, weighted sum of the
// 64 values (two operand pairs: hi + lo is the value, as the tile kernels' act_get reads it), 32 channels at a time: the
// rows go through LDS (logits | values | mask; the scratch overwrites the W ring, which is dead: the last superstep's
// barrier is behind every wave's last ring read), then a half-wave per point, lane = channel; the same expressions in
// the same order as pool_masked_softmax.
template <bool F16>
__device__ __forceinline__ void rr_pool(float *lds, const float4 (&logit)[4], const ActPair &v0, const ActPair &v1, float mk, int r,
                                        int wave, int lane, int K, int P, long first_point, long total_points, void *out)
{
    const int kq = lane >> 4;
    float *lg = lds, *xv = lds + RR_ROWS * RR_PITCH, *mrow = lds + 2 * RR_ROWS * RR_PITCH;
    if (kq == 0) mrow[r] = mk;
    RR_POOL_STAMP(2);
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        if (half) rr_barrier();                      // the first half's reads are done (rr_barrier: LDS traffic only -- __syncthreads would
                                                     // also wait for the first half's global stores, ~1-2 us)
#pragma unroll
        for (int c2 = 0; c2 < 2; ++c2) {
            const int cb = 2 * half + c2;
            *reinterpret_cast<float4 *>(lg + r * RR_PITCH + c2 * 16 + 4 * kq) = logit[cb];
            const ActPair &xp = half ? v1 : v0;      // pair `half` = column blocks 2*half, 2*half + 1 of the values
            *reinterpret_cast<float4 *>(xv + r * RR_PITCH + c2 * 16 + 4 * kq) = c2 == 0 ? quad_value(xp.hi.x, xp.hi.y, xp.lo.x, xp.lo.y)
                                                                                       : quad_value(xp.hi.z, xp.hi.w, xp.lo.z, xp.lo.w);
        }
        if (!half) RR_POOL_STAMP(3); else RR_POOL_STAMP(6);
        rr_barrier();
        if (!half) RR_POOL_STAMP(4); else RR_POOL_STAMP(7);
        const int ch = lane & 31;
        for (int pp = wave * 2 + (lane >> 5); pp < P; pp += 2 * RR_WAVES) {
            const long p = first_point + pp;
            if (p >= total_points) break;
            const float *lcol = lg + (pp * K) * RR_PITCH + ch, *vcol = xv + (pp * K) * RR_PITCH + ch;
            float mx = -INFINITY, den = 0.0f, sum = 0.0f;
            if (K == 6) {                                 // (uniform)
                float l[6], v[6], w[6];
#pragma unroll
                for (int u = 0; u < 6; ++u) { l[u] = lcol[u * RR_PITCH]; v[u] = vcol[u * RR_PITCH]; w[u] = mrow[pp * 6 + u]; }
                softmax_chunk8<6, true>(l, v, w, 6, mx, den, sum);
            } else
            for (int k0 = 0; k0 < K; k0 += 8) {
                float l[8], v[8], w[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int k = min(k0 + u, K - 1);
                    l[u] = lcol[k * RR_PITCH];
                    v[u] = vcol[k * RR_PITCH];
                    w[u] = mrow[pp * K + k];
                }
                softmax_chunk8(l, v, w, K - k0, mx, den, sum);
            }
            feat_store(out, p * 64 + 32 * half + ch, sum / den, F16);
        }
        if (!half) RR_POOL_STAMP(5);
    }
}

// ---- pooling INSIDE the wave, for K | 16: the K rows of a point are K adjacent lanes of one 16-lane DPP row (row r = wave*16
// + i16), so the reduction over K needs no LDS and no barrier -- where the LDS form is a fifth of a workgroup's time
// (tools/rr_clock.sh: ~6.7 k of ~32 k cycles: two write / barrier / half-wave-per-point rounds, with the waves' skew in the
// barriers).  Both forms give the same bits: a maximum does not depend on the order, and the softmax sums are scanned in
// the order softmax_chunk8 adds them.  Values must be >= 0 (a ReLU in front: the launchers check), so that the maxima can be
// taken on the bit patterns (dpp_imax_step: one v_max_i32_dpp per step).
// (`old` = the maximum's identity, as in dpp_imax_step: these controls read a valid lane everywhere, and with the identity there
//  the compiler folds move + max into ONE v_max_i32_dpp; with old = v it emitted v_mov_b32_dpp + v_max_i32: 128 of the 1.6 k vector
//  instructions of a setconv_narrow wave, 64 of the 600 of a setconv_rr wave)
template <int CTRL>
__device__ __forceinline__ int dpp_imax_all(int v) { return max(__builtin_amdgcn_update_dpp((int)0x80000000, v, CTRL, 0xf, 0xf, false), v); }

template <int NOUT, bool F16>
__device__ __forceinline__ void rr_pool_max_inwave(const float4 (&last)[NOUT / 16], float mk, int r, int lane, int K, long first_point,
                                                   long total_points, void *out)
{
    const int kq = lane >> 4, i16 = lane & 15;
    const long p = first_point + (K == 16 ? r >> 4 : r >> 3);
    const bool writer = (i16 & (K - 1)) == 0 && p < total_points;
#pragma unroll
    for (int cb = 0; cb < NOUT / 16; ++cb) {
        const float q[4] = {last[cb].x * mk, last[cb].y * mk, last[cb].z * mk, last[cb].w * mk};
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            int v = __float_as_int(q[e]);
            v = dpp_imax_all<0xb1>(v);               // quad_perm:[1,0,3,2]
            v = dpp_imax_all<0x4e>(v);               // quad_perm:[2,3,0,1]
            v = dpp_imax_all<0x141>(v);              // row_half_mirror: 8 lanes
            if (K == 16) v = dpp_imax_all<0x140>(v); // row_mirror: 16 lanes
            o[e] = __int_as_float(v);
        }
        if (writer) feat_store4(out, p * NOUT + cb * 16 + 4 * kq, float4{o[0], o[1], o[2], o[3]}, F16);
    }
}

// masked softmax over K = 4 rows of the 64 logits, weighted sum of the 64 values (rr_pool's arguments): lane 3 of each quad
// of lanes ends up with softmax_chunk8's sums in softmax_chunk8's order (a left fold over the rows: row_shr:1 three times)
template <bool F16>
__device__ __forceinline__ void rr_pool_softmax4_inwave(const float4 (&logit)[4], const ActPair &v0, const ActPair &v1, float mk, int r,
                                                        int lane, long first_point, long total_points, void *out)
{
    const int kq = lane >> 4, i16 = lane & 15;
    const long p = first_point + (r >> 2);
    const bool writer = (i16 & 3) == 3 && p < total_points;
    // row_shr:1 with bound_ctrl: lane 0 of a row reads 0 -- and the compiler folds move + add into ONE v_add_f32_dpp (with
    // old = 0 and no bound_ctrl it kept them apart: 96 of the wave's instructions)
    auto shr1 = [](float x) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x111, 0xf, 0xf, true)); };
    auto quad_lane = [](float x, auto sel) {          // lane `sel` of this lane's quad, in every lane of the quad
        return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), decltype(sel)::value * 0x55, 0xf, 0xf, true));
    };
    const int j = i16 & 3;
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) {
        const ActPair &xp = cb < 2 ? v0 : v1;
        const float4 val = cb % 2 == 0 ? quad_value(xp.hi.x, xp.hi.y, xp.lo.x, xp.lo.y) : quad_value(xp.hi.z, xp.hi.w, xp.lo.z, xp.lo.w);
        const float l[4] = {logit[cb].x, logit[cb].y, logit[cb].z, logit[cb].w}, v[4] = {val.x, val.y, val.z, val.w};
        float den[4], acc[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float x = mk == 1.0f ? l[e] : -1e10f;
            int m = __float_as_int(x);               // (logits >= 0 or exactly -1e10: integer order = float order)
            m = dpp_imax_all<0xb1>(m);
            m = dpp_imax_all<0x4e>(m);
            const float ex = exp_hw(x - __int_as_float(m)), ev = ex * v[e];
            float cd = ex, ca = ev;
#pragma unroll
            for (int step = 0; step < 3; ++step) { cd = shr1(cd) + ex; ca = shr1(ca) + ev; }
            // (softmax_chunk8's merge into (-inf, 0, 0): den = 0 * 0 + cd * 1, acc = 0 * 0 + ca * 1)
            den[e] = cd; acc[e] = ca;
        }
        // lane 3 of the quad holds the four (acc, den) pairs; a division is ~11 instructions in every lane whoever needs it, so
        // the quad shares them: lane j divides pair j (read from lane 3), lane 3 collects the quotients (same division, same bits)
        typedef std::integral_constant<int, 3> L3;
        float a4[4], d4[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) { a4[e] = quad_lane(acc[e], L3{}); d4[e] = quad_lane(den[e], L3{}); }
        const float aj = j == 0 ? a4[0] : j == 1 ? a4[1] : j == 2 ? a4[2] : a4[3], dj = j == 0 ? d4[0] : j == 1 ? d4[1] : j == 2 ? d4[2] : d4[3];
        const float qj = aj / dj;
        const float4 o{quad_lane(qj, std::integral_constant<int, 0>{}), quad_lane(qj, std::integral_constant<int, 1>{}),
                       quad_lane(qj, std::integral_constant<int, 2>{}), quad_lane(qj, L3{})};
        if (writer) feat_store4(out, p * 64 + cb * 16 + 4 * kq, o, F16);
    }
}

// The biases of a chain's layers, concatenated, into the workgroup's LDS table (words RR_BIAS_OFF ...: behind the ring and the
// set-conv grouping scratch, inside the pooling scratch -- dead, like the ring, by the time the pooling writes).  One float
// per thread (<= 512 in all), requested at kernel start and written before the first rr_barrier().
constexpr int RR_BIAS_OFF = 8192;
template <int NL>
__device__ __forceinline__ float rr_bias_request(const float *const (&b)[NL], const int (&n)[NL], int tid)
{
    int at = tid;
    const float *src = b[0];                        // (clamped: every thread loads something)
    bool found = false;
#pragma unroll
    for (int l = 0; l < NL; ++l) {
        if (!found && at < n[l]) { src = b[l] + at; found = true; }
        at -= n[l];
    }
    return *src;
}
__device__ __forceinline__ void rr_bias_store(float *lds, int tid, int total, float v) { if (tid < total) lds[RR_BIAS_OFF + tid] = v; }

template <int FP> struct Cv1Plan {   // CV_0 (FP pairs + geometry tail -> 128), CV_1, CV_2, CV_xyz (tail -> 64), sum_CV_0, sum_CV_1
    static constexpr int NL = 6;
    static constexpr int KP[6] = {FP, 4, 2, 0, 4, 4}, TL[6] = {1, 0, 0, 1, 0, 0}, NCB[6] = {8, 4, 4, 4, 8, 4};
};

// `block` of `nblocks`: blockIdx.x / gridDim.x of a plain launch, or the workgroup's share of a heterogeneous one (cv1_setconv_rr_kernel)
template <int C, bool F16, int MODE = MODE_SPLIT>
__device__ __forceinline__ void cv1_rr_body(const elo_cv1_args &a, const unsigned block, const unsigned nblocks)
{
    extern __shared__ __align__(16) float lds[];
    constexpr int FP = C / 16;                       // 16-channel blocks per feature tensor; CV_0's pairs: FP (feat1 | feat2 blocks paired up)
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i16 = lane & 15, kq = lane >> 4;
    uint4 *ring = reinterpret_cast<uint4 *>(lds);    // [2 slots][8 chunks][64 lanes] x 16 bytes
    const int K = a.K, P = RR_ROWS / K;              // points per workgroup: rows r = point * K + slot, r < P * K <= 128
    const long total_points = (long)a.batch * a.npoints;
    const long first_point = (long)xcd_tile(block, nblocks) * P;
    unsigned bad = 0;
    RR_STAMP(0);
    // ---- the W stream.  Steps per layer: CV_0 4*(FP+1), CV_1 2*4, CV_2 2*2, CV_xyz 2*1, sum_CV_0 4*4, sum_CV_1 2*4
    typedef RrStream<Cv1Plan<FP>, MODE> Stream;
    Stream ws{{a.cv0.w_packed, a.cv1.w_packed, a.cv2.w_packed, a.cv_xyz.w_packed, a.sum_cv0.w_packed, a.sum_cv1.w_packed}, ring, wave, lane, {}};
    constexpr int E0 = Stream::first(1), E1 = Stream::first(2), E2 = Stream::first(3), E3 = Stream::first(4), E4 = Stream::first(5);
    auto fetch = [&](int g, RrW &dst) { ws.fetch(g, dst); };
    auto advance = [&](int S) { ws.advance(S); };
    ws.start();
    float4 bias[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) bias[t] = *reinterpret_cast<const float4 *>(a.cv0.bias + t * 16 + 4 * kq);
    const float *const bsrc[6] = {a.cv0.bias, a.cv1.bias, a.cv2.bias, a.cv_xyz.bias, a.sum_cv0.bias, a.sum_cv1.bias};
    constexpr int BN[6] = {128, 64, 64, 64, 128, 64}, B1 = 128, B2 = 192, B3 = 256, B4 = 320, B5 = 448;     // table offsets
    const float bias_word = rr_bias_request(bsrc, BN, (int)threadIdx.x);
    const float *bt = lds + RR_BIAS_OFF;
    // ---- row metadata + gather, every load requested before the first use (this lane's row: r = wave * 16 + i16)
    const int r = wave * 16 + i16, pi = small_div(r, K);
    long pt = first_point + pi;
    const bool used = pi < P && pt < total_points;
    if (!used) pt = first_point;
    const long gr = used ? pt * K + (r - pi * K) : first_point * K;
    const int id0 = a.idx[gr * 3 + 0], id1 = a.idx[gr * 3 + 1], id2 = a.idx[gr * 3 + 2];
    float mk = a.mask[gr];
    const int cell = (id0 * a.H2 + id1) * a.W2 + id2;
    if (!used) mk = 0.0f;
    const bool keep2 = used && mk != 0.0f;
    ActPair in0[FP];                                 // CV_0's input pairs: [feat1 | feat2] blocks, two per pair
    uint4 geo;                                       // its tail: quad kq of [p, g*m, g*m - p, |g*m - p|, 0 ...]
    {
        typedef typename std::conditional<F16, uint2, float4>::type Item;     // 4 channels
        Item f1[FP], f2[FP];
        const Item *r1 = reinterpret_cast<const Item *>(a.feat1) + (pt * C >> 2) + kq;
        const Item *r2 = reinterpret_cast<const Item *>(a.feat2) + ((long)cell * C >> 2) + kq;
#pragma unroll
        for (int j = 0; j < FP; ++j) { f1[j] = r1[j * 4]; f2[j] = r2[j * 4]; }
        const float *c = a.xyz1 + pt * 3, *g = a.xyz2 + (long)cell * 3;
        const float pc0 = c[0], pc1 = c[1], pc2 = c[2], pg0 = g[0], pg1 = g[1], pg2 = g[2];
        ws.prime();                                  // superstep 0 into the ring (its load went out first)
        auto quad = [&](const Item &v, bool keep) {
            const uint4 z{0u, 0u, 0u, 0u};
            if constexpr (F16) return keep ? quad_of_halves(v) : z;
            else return keep ? pack_quad<MODE>(v, bad) : z;
        };
        // CV_0's k-blocks in order: feat1 blocks 0..FP-1, feat2 blocks 0..FP-1, geometry; block b sits in pair b / 2, half b % 2
#pragma unroll
        for (int j = 0; j < FP; ++j) put_quad(in0[j / 2], j % 2, quad(f1[j], used));
#pragma unroll
        for (int j = 0; j < FP; ++j) put_quad(in0[(FP + j) / 2], (FP + j) % 2, quad(f2[j], keep2));
        const float g0 = pg0 * mk, g1 = pg1 * mk, g2 = pg2 * mk;
        const float d0 = g0 - pc0, d1 = g1 - pc1, d2 = g2 - pc2;
        const float e = sqrtf(d0 * d0 + d1 * d1 + d2 * d2 + 1e-20f);
        const float4 q4 = kq == 0 ? float4{pc0, pc1, pc2, g0} : kq == 1 ? float4{g1, g2, d0, d1}
                        : kq == 2 ? float4{d2, e, 0.0f, 0.0f} : float4{0.0f, 0.0f, 0.0f, 0.0f};
        geo = used ? pack_quad<MODE>(q4, bad) : uint4{0u, 0u, 0u, 0u};
    }
    rr_bias_store(lds, (int)threadIdx.x, 512, bias_word);
    RR_STAMP(1);
    rr_barrier();                                    // superstep 0 is in the ring
    RR_LAYER_STAMP(2);
    // ---- the chain
    const uint4 none{0u, 0u, 0u, 0u};
    ActPair h128[4];                                 // CV_0's output (128 channels = 4 pairs)
    auto emit0 = [&](int pass, int t, const f32x4 acc) { put_quad(h128[pass], t, pack_quad<MODE>(relu4(acc, a.cv0.relu), bad)); };
    rr_layer<FP, true, 8, 0, MODE, (F16 ? ~0u : 0u)>(in0, geo, bt, bt + B1, bias, fetch, advance, emit0);
    RR_LAYER_STAMP(3);
    ActPair h64[2];
    rr_layer<4, false, 4, E0, MODE>(h128, none, bt + B1, bt + B2, bias, fetch, advance, [&](int pass, int t, const f32x4 acc) {
        put_quad(h64[pass], t, pack_quad<MODE>(relu4(acc, a.cv1.relu), bad));
    });
    RR_LAYER_STAMP(4);
    ActPair xe[4];                                   // [x | enc]: sum_CV_0's input; x stays alive: it is the pooling's value
    rr_layer<2, false, 4, E1, MODE>(h64, none, bt + B2, bt + B3, bias, fetch, advance, [&](int pass, int t, const f32x4 acc) {
        put_quad(xe[pass], t, pack_quad<MODE>(relu4(acc, a.cv2.relu), bad));
    });
    RR_LAYER_STAMP(5);
    {
        ActPair unused[1];
        rr_layer<0, true, 4, E2, MODE>(unused, geo, bt + B3, bt + B4, bias, fetch, advance, [&](int pass, int t, const f32x4 acc) {
            put_quad(xe[2 + pass], t, pack_quad<MODE>(relu4(acc, a.cv_xyz.relu), bad));
        });
    }
    RR_LAYER_STAMP(6);
    rr_layer<4, false, 8, E3, MODE>(xe, none, bt + B4, bt + B5, bias, fetch, advance, [&](int pass, int t, const f32x4 acc) {
        put_quad(h128[pass], t, pack_quad<MODE>(relu4(acc, a.sum_cv0.relu), bad));
    });
    RR_LAYER_STAMP(7);
    float4 logit[4];                                 // plain fp32, held until the ring is dead
    rr_layer<4, false, 4, E4, MODE>(h128, none, bt + B5, nullptr, bias, fetch, advance, [&](int pass, int t, const f32x4 acc) {
        logit[2 * pass + t] = relu4(acc, a.sum_cv1.relu);
    });
    RR_STAMP(8);
    rr_pool<F16>(lds, logit, xe[0], xe[1], mk, r, wave, lane, K, P, first_point, total_points, a.out);
    RR_STAMP(9);
    report_violations<MODE>(bad, a.range_counter);
}

template <int C, bool F16, int MODE = MODE_SPLIT>
__global__ __launch_bounds__(RR_WAVES * 64, 4) void cv1_rr_kernel(const elo_cv1_args a)
{
    cv1_rr_body<C, F16, MODE>(a, blockIdx.x, gridDim.x);
}
// ---- cost volume, stage 2, register-resident (see cv1_rr_kernel): xyz-encoding (geometry tail -> 64), sum_cost_volume_0
// ([grouped cost (64) | encoding (64) | feat1 (C)] -> 128), sum_cost_volume_1 (-> 64 logits), masked softmax over the K
// rows of a point weighting the grouped cost (utils/pointnet_util.py:110-146).  Neighbours from a.idx / a.mask.
template <int C> struct Cv2Plan {
    static constexpr int NL = 3;
    static constexpr int KP[3] = {0, 4 + C / 32, 4}, TL[3] = {1, C == 16 ? 1 : 0, 0}, NCB[3] = {4, 8, 4};
};

template <int C, bool F16, int MODE = MODE_SPLIT>
__global__ __launch_bounds__(RR_WAVES * 64, 4) void cv2_rr_kernel(const elo_cv2_args a)
{
    extern __shared__ __align__(16) float lds[];
    constexpr int FP = C / 16, KP1 = 4 + C / 32;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i16 = lane & 15, kq = lane >> 4;
    const int K = a.K, P = RR_ROWS / K;
    const long total_points = (long)a.batch * a.npoints;
    const long first_point = (long)xcd_tile(blockIdx.x, gridDim.x) * P;
    unsigned bad = 0;
    typedef RrStream<Cv2Plan<C>, MODE> Stream;
    Stream ws{{a.xyz_enc.w_packed, a.sum_cost0.w_packed, a.sum_cost1.w_packed}, reinterpret_cast<uint4 *>(lds), wave, lane, {}};
    constexpr int E0 = Stream::first(1), E1 = Stream::first(2);
    auto fetch = [&](int g, RrW &dst) { ws.fetch(g, dst); };
    auto advance = [&](int S) { ws.advance(S); };
    ws.start();
    float4 bias[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) bias[t] = *reinterpret_cast<const float4 *>(a.xyz_enc.bias + t * 16 + 4 * kq);
    const float *const bsrc[3] = {a.xyz_enc.bias, a.sum_cost0.bias, a.sum_cost1.bias};
    constexpr int BN[3] = {64, 128, 64}, B1 = 64, B2 = 192;
    const float bias_word = rr_bias_request(bsrc, BN, (int)threadIdx.x);
    const float *bt = lds + RR_BIAS_OFF;
    // ---- row metadata + gather (this lane's row: r = wave * 16 + i16 = point * K + slot)
    const int r = wave * 16 + i16, pi = small_div(r, K);
    long pt = first_point + pi;
    const bool used = pi < P && pt < total_points;
    if (!used) pt = first_point;
    const long gr = used ? pt * K + (r - pi * K) : first_point * K;
    const int id0 = a.idx[gr * 3 + 0], id1 = a.idx[gr * 3 + 1], id2 = a.idx[gr * 3 + 2];
    float mk = a.mask[gr];
    const int cell = (id0 * a.H + id1) * a.W + id2;
    if (!used) mk = 0.0f;
    const bool keep2 = used && mk != 0.0f;
    ActPair in1[KP1];                                // sum_cost_volume_0's pairs: [grouped cost | encoding | feat1]
    uint4 geo, ftail = uint4{0u, 0u, 0u, 0u};        // the geometry quad (xyz-encoding's input); feat1's quad when C == 16 (a tail block)
    typedef typename std::conditional<F16, uint2, float4>::type Item;         // 4 channels
    Item fc[4];                                      // the grouped cost rows
    const Item *rc = reinterpret_cast<const Item *>(a.cost) + ((long)cell * 64 >> 2) + kq;
    auto quad = [&](const Item &v, bool keep) {
        const uint4 z{0u, 0u, 0u, 0u};
        if constexpr (F16) return keep ? quad_of_halves(v) : z;
        else return keep ? pack_quad<MODE>(v, bad) : z;
    };
    {
        Item f1[FP];
        const Item *r1 = reinterpret_cast<const Item *>(a.feat1) + (pt * C >> 2) + kq;
#pragma unroll
        for (int j = 0; j < 4; ++j) fc[j] = rc[j * 4];
#pragma unroll
        for (int j = 0; j < FP; ++j) f1[j] = r1[j * 4];
        const float *c = a.xyz1 + pt * 3, *g = a.xyz1 + (long)cell * 3;
        const float pc0 = c[0], pc1 = c[1], pc2 = c[2], pg0 = g[0], pg1 = g[1], pg2 = g[2];
        ws.prime();
#pragma unroll
        for (int j = 0; j < 4; ++j) put_quad(in1[j / 2], j % 2, quad(fc[j], keep2));       // cost[idx] * mask   :110
        if constexpr (C == 16) ftail = quad(f1[0], used);
        else {
#pragma unroll
            for (int j = 0; j < FP; ++j) put_quad(in1[4 + j / 2], j % 2, quad(f1[j], used));   // centre features    :115
        }
        const float g0 = pg0 * mk, g1 = pg1 * mk, g2 = pg2 * mk;
        const float d0 = g0 - pc0, d1 = g1 - pc1, d2 = g2 - pc2;
        const float e = sqrtf(d0 * d0 + d1 * d1 + d2 * d2 + 1e-20f);
        const float4 q4 = kq == 0 ? float4{pc0, pc1, pc2, g0} : kq == 1 ? float4{g1, g2, d0, d1}
                        : kq == 2 ? float4{d2, e, 0.0f, 0.0f} : float4{0.0f, 0.0f, 0.0f, 0.0f};
        geo = used ? pack_quad<MODE>(q4, bad) : uint4{0u, 0u, 0u, 0u};
    }
    rr_bias_store(lds, (int)threadIdx.x, 256, bias_word);
    rr_barrier();
    const uint4 none{0u, 0u, 0u, 0u};
    {
        ActPair unused[1];
        rr_layer<0, true, 4, 0, MODE>(unused, geo, bt, bt + B1, bias, fetch, advance, [&](int pass, int t, const f32x4 acc) {
            put_quad(in1[2 + pass], t, pack_quad<MODE>(relu4(acc, a.xyz_enc.relu), bad));      // encoding: pairs 2, 3
        });
    }
    ActPair h128[4];
    auto emit1 = [&](int pass, int t, const f32x4 acc) { put_quad(h128[pass], t, pack_quad<MODE>(relu4(acc, a.sum_cost0.relu), bad)); };
    rr_layer<KP1, C == 16, 8, E0, MODE, (F16 ? ~0xcu : 0u), F16>(in1, ftail, bt + B1, bt + B2, bias, fetch, advance, emit1);
    float4 logit[4];
    rr_layer<4, false, 4, E1, MODE>(h128, none, bt + B2, nullptr, bias, fetch, advance, [&](int pass, int t, const f32x4 acc) {
        logit[2 * pass + t] = relu4(acc, a.sum_cost1.relu);
    });
    if (K == 4 && a.sum_cost1.relu) rr_pool_softmax4_inwave<F16>(logit, in1[0], in1[1], mk, r, lane, first_point, total_points, a.out);
    else rr_pool<F16>(lds, logit, in1[0], in1[1], mk, r, wave, lane, K, P, first_point, total_points, a.out);
    report_violations<MODE>(bad, a.range_counter);
}
// ---- two chained row-wise MLPs, register-resident (see cv1_rr_kernel): set-upconv stage 2 and the flow predictor it feeds
// (mlp_kernel's two-stage form: utils/pointnet_util.py:303-311, :161-175) on the model's widths -- stage 1 [pooled (64) |
// points (C)] -> 128 -> 64 = `out` (stored), stage 2 [out (64) | before (C) | after (64)] -> 128 -> 64 = `out2`.  No grouping
// and no pooling: a lane's row is a global row.  blockIdx.y selects one of two jobs.  With fp16 storage `out` continues AS
// STORED (store_quad's rule), so the chain computes exactly what mlp_kernel does.
template <int C> struct Mlp2Plan {
    static constexpr int NL = 4, B1 = 4 + C / 16, B3 = 8 + C / 16;     // 16-k blocks of the two first layers
    static constexpr int KP[4] = {B1 / 2, 4, B3 / 2, 4}, TL[4] = {B1 % 2, 0, B3 % 2, 0}, NCB[4] = {8, 4, 8, 4};
};

template <int C, bool F16, int MODE = MODE_SPLIT>
__global__ __launch_bounds__(RR_WAVES * 64, 4) void mlp2_rr_kernel(const JobPair<elo_mlp_args> jobs)
{
    extern __shared__ __align__(16) float lds[];
    {   // side job (elo_mlp_args.clear_*, job 0's), as in mlp_kernel
        const elo_mlp_args &j0 = jobs.job[0];
        clear_projection(ProjectionClear{j0.clear_scratch, j0.clear_xyz, (unsigned *)j0.clear_feat, j0.clear_cells,
                                         j0.feat_dtype == ELO_F16 ? j0.clear_C / 2 : j0.clear_C, j0.clear_images});
    }
    const elo_mlp_args &a = jobs.job[blockIdx.y];
    typedef Mlp2Plan<C> PL;
    constexpr int FP = C / 16, KP1 = PL::KP[0], KP3 = PL::KP[2];
    constexpr bool T1 = PL::TL[0] != 0, T3 = PL::TL[2] != 0;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i16 = lane & 15, kq = lane >> 4;
    const long first = (long)xcd_tile(blockIdx.x, gridDim.x) * RR_ROWS;
    unsigned bad = 0;
    typedef RrStream<PL, MODE> Stream;
    Stream ws{{a.layers[0].w_packed, a.layers[1].w_packed, a.layers2[0].w_packed, a.layers2[1].w_packed}, reinterpret_cast<uint4 *>(lds), wave, lane, {}};
    constexpr int E0 = Stream::first(1), E1 = Stream::first(2), E2 = Stream::first(3);
    auto fetch = [&](int g, RrW &dst) { ws.fetch(g, dst); };
    auto advance = [&](int S) { ws.advance(S); };
    ws.start();
    float4 bias[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) bias[t] = *reinterpret_cast<const float4 *>(a.layers[0].bias + t * 16 + 4 * kq);
    const float *const bsrc[4] = {a.layers[0].bias, a.layers[1].bias, a.layers2[0].bias, a.layers2[1].bias};
    constexpr int BN[4] = {128, 64, 128, 64}, O1 = 128, O2 = 192, O3 = 320;
    const float bias_word = rr_bias_request(bsrc, BN, (int)threadIdx.x);
    const float *bt = lds + RR_BIAS_OFF;
    // ---- this lane's row
    const int r = wave * 16 + i16;
    const bool used = first + r < a.rows;
    const long gr = used ? first + r : a.rows - 1;
    typedef typename std::conditional<F16, uint2, float4>::type Item;         // 4 channels
    auto quad = [&](const Item &v) {
        const uint4 z{0u, 0u, 0u, 0u};
        if constexpr (F16) return used ? quad_of_halves(v) : z;
        else return used ? pack_quad<MODE>(v, bad) : z;
    };
    // block b of a concatenation goes to half b % 2 of pair b / 2, the last block of an odd count to the tail
    ActPair in1[KP1];
    uint4 tail1{0u, 0u, 0u, 0u};
    {
        Item s0[4], s1[FP];
        const Item *p0 = reinterpret_cast<const Item *>(a.src[0]) + (gr * 64 >> 2) + kq, *p1 = reinterpret_cast<const Item *>(a.src[1]) + (gr * C >> 2) + kq;
#pragma unroll
        for (int j = 0; j < 4; ++j) s0[j] = p0[j * 4];
#pragma unroll
        for (int j = 0; j < FP; ++j) s1[j] = p1[j * 4];
        ws.prime();
#pragma unroll
        for (int j = 0; j < 4; ++j) put_quad(in1[j / 2], j % 2, quad(s0[j]));
#pragma unroll
        for (int j = 0; j < FP; ++j) {
            if (T1 && j == FP - 1) tail1 = quad(s1[j]);
            else put_quad(in1[(4 + j) / 2], (4 + j) % 2, quad(s1[j]));
        }
    }
    rr_bias_store(lds, (int)threadIdx.x, 384, bias_word);
    rr_barrier();
    const uint4 none{0u, 0u, 0u, 0u};
    ActPair h128[4];
    rr_layer<KP1, T1, 8, 0, MODE, (F16 ? ~0u : 0u), F16>(in1, tail1, bt, bt + O1, bias, fetch, advance, [&](int pass, int t, const f32x4 acc) {
        put_quad(h128[pass], t, pack_quad<MODE>(relu4(acc, a.layers[0].relu), bad));
    });
    // the second stage's other inputs: requested now, a layer ahead of their first use
    Item sb[FP], sa[4];
    {
        const Item *pb = reinterpret_cast<const Item *>(a.before) + (gr * C >> 2) + kq, *pa = reinterpret_cast<const Item *>(a.after) + (gr * 64 >> 2) + kq;
#pragma unroll
        for (int j = 0; j < FP; ++j) sb[j] = pb[j * 4];
#pragma unroll
        for (int j = 0; j < 4; ++j) sa[j] = pa[j * 4];
    }
    ActPair in3[KP3];
    uint4 tail3{0u, 0u, 0u, 0u};
    rr_layer<4, false, 4, E0, MODE>(h128, none, bt + O1, bt + O2, bias, fetch, advance, [&](int pass, int t, const f32x4 acc) {
        const float4 v = relu4(acc, a.layers[1].relu);
        const long at = (first + r) * 64 + (2 * pass + t) * 16 + 4 * kq;
        if constexpr (F16) {
            const half2v h0 = half2v{(_Float16)v.x, (_Float16)v.y}, h1 = half2v{(_Float16)v.z, (_Float16)v.w};
            const uint2 stored{__builtin_bit_cast(unsigned, h0), __builtin_bit_cast(unsigned, h1)};
            if (used) *reinterpret_cast<uint2 *>(reinterpret_cast<_Float16 *>(a.out) + at) = stored;
            put_quad(in3[pass], t, quad_of_halves(stored));
        } else {
            if (used) *reinterpret_cast<float4 *>(reinterpret_cast<float *>(a.out) + at) = v;
            put_quad(in3[pass], t, pack_quad<MODE>(v, bad));
        }
    });
    // [out (blocks 0..3) | before (FP blocks) | after (4 blocks)]
#pragma unroll
    for (int j = 0; j < FP; ++j) put_quad(in3[(4 + j) / 2], (4 + j) % 2, quad(sb[j]));
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        constexpr int B0 = 4 + FP;
        if (T3 && j == 3) tail3 = quad(sa[j]);
        else put_quad(in3[(B0 + j) / 2], (B0 + j) % 2, quad(sa[j]));
    }
    rr_layer<KP3, T3, 8, E1, MODE, (F16 ? ~0u : 0u), F16>(in3, tail3, bt + O2, bt + O3, bias, fetch, advance, [&](int pass, int t, const f32x4 acc) {
        put_quad(h128[pass], t, pack_quad<MODE>(relu4(acc, a.layers2[0].relu), bad));
    });
    rr_layer<4, false, 4, E2, MODE>(h128, none, bt + O3, nullptr, bias, fetch, advance, [&](int pass, int t, const f32x4 acc) {
        if (used) feat_store4(a.out2, (first + r) * 64 + (2 * pass + t) * 16 + 4 * kq, relu4(acc, a.layers2[1].relu), F16);
    });
    report_violations<MODE>(bad, a.range_counter);
}
// ---- set-conv / set-upconv stage 1, register-resident (see cv1_rr_kernel): in-kernel random-k grouping (a wave groups
// the points its 16 rows belong to), gather [features (16 FPB) | xyz difference], two or three layers of compile-time
// widths, masked max over the K rows of a point (utils/pointnet_util.py:197-230, :272-298).  Instantiated for the
// model's shapes whose layers all have an even number of k-steps: the set-upconvs (64 + 3 -> 128 -> 64), sa1/layer3
// (64 + 3 -> 64 -> 64 -> 128) and new_layer3 (64 + 3 -> 128 -> 64 -> 64); blockIdx.y selects one of two jobs.
template <int FPB, int N1, int N2, int N3> struct ScPlan {
    static constexpr int NL = N3 ? 3 : 2;
    static constexpr int KP[3] = {FPB / 2, N1 / 32, N2 / 32}, TL[3] = {1, 0, 0}, NCB[3] = {N1 / 16, N2 / 16, (N3 ? N3 : 32) / 16};
};

// masked max over the K rows of a point of the NOUT channels the lane quads hold (`last[cb]`: channels cb*16 + 4kq ..),
// 32 channels at a time through LDS (the scratch overwrites the dead W ring): pool_masked_max's expressions
template <int NOUT, bool F16>
__device__ __forceinline__ void rr_pool_max(float *lds, const float4 (&last)[NOUT / 16], float mk, int r, int wave, int lane, int K,
                                            int P, long first_point, long total_points, void *out)
{
    const int kq = lane >> 4;
    float *vv = lds, *mrow = lds + RR_ROWS * RR_PITCH;
    if (kq == 0) mrow[r] = mk;
#pragma unroll
    for (int part = 0; part < NOUT / 32; ++part) {
        if (part) rr_barrier();
#pragma unroll
        for (int c2 = 0; c2 < 2; ++c2) *reinterpret_cast<float4 *>(vv + r * RR_PITCH + c2 * 16 + 4 * kq) = last[2 * part + c2];
        rr_barrier();
        const int ch = lane & 31;
        for (int pp = wave * 2 + (lane >> 5); pp < P; pp += 2 * RR_WAVES) {
            const long p = first_point + pp;
            if (p >= total_points) break;
            const float *col = vv + (pp * K) * RR_PITCH + ch, *mkp = mrow + pp * K;
            float best = -INFINITY;
            for (int k0 = 0; k0 < K; k0 += 8) {
                float v[8], w[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int k = min(k0 + u, K - 1);
                    v[u] = col[k * RR_PITCH];
                    w[u] = mkp[k];
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) best = fmaxf(best, v[u] * w[u]);
            }
            feat_store(out, p * NOUT + 32 * part + ch, best, F16);
        }
    }
}

template <int FPB, int N1, int N2, int N3, bool F16, int MODE = MODE_SPLIT>
__device__ __forceinline__ void setconv_rr_body(const elo_setconv_args &a, const unsigned block, const unsigned nblocks)
{
    extern __shared__ __align__(16) float lds[];
    constexpr int C = 16 * FPB, NOUT = N3 ? N3 : N2;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i16 = lane & 15, kq = lane >> 4, tid = threadIdx.x;
    const int K = a.K, P = RR_ROWS / K;
    const long total_points = (long)a.batch * a.npoints;
    const long first_point = (long)xcd_tile(block, nblocks) * P;
    unsigned bad = 0;
    RR_STAMP(0);
    typedef RrStream<ScPlan<FPB, N1, N2, N3>, MODE> Stream;
    Stream ws{};
    ws.w[0] = a.layers[0].w_packed; ws.w[1] = a.layers[1].w_packed;
    if constexpr (N3 != 0) ws.w[2] = a.layers[2].w_packed;
    ws.ring = reinterpret_cast<uint4 *>(lds); ws.wave = wave; ws.lane = lane;
    constexpr int E0 = Stream::first(1), E1 = N3 ? Stream::first(N3 ? 2 : 1) : 0;
    auto fetch = [&](int g, RrW &dst) { ws.fetch(g, dst); };
    auto advance = [&](int S) { ws.advance(S); };
    ws.start();
    float4 bias[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) bias[t] = *reinterpret_cast<const float4 *>(a.layers[0].bias + t * 16 + 4 * kq);
    const float *const bsrc[3] = {a.layers[0].bias, a.layers[1].bias, N3 ? a.layers[2].bias : a.layers[1].bias};
    constexpr int BN[3] = {N1, N2, N3}, B1 = N1, B2 = N1 + N2;
    const float bias_word = rr_bias_request(bsrc, BN, tid);
    const float *bt = lds + RR_BIAS_OFF;
    // ---- grouping.  Scratch behind the 16 KB ring: [KT] visiting order | [128] cell | [128] mask | [P] centres (hw, xyz)
    int *lds_off = reinterpret_cast<int *>(lds + 4096);
    const int KT = a.group.kernel_h * a.group.kernel_w;
    int *cell_row = lds_off + ((KT + 3) & ~3);
    float *mask_row = reinterpret_cast<float *>(cell_row + RR_ROWS);
    int *chw = reinterpret_cast<int *>(mask_row + RR_ROWS);                  // [128] (h << 16) | w of the centres
    float *cxyz = reinterpret_cast<float *>(chw + RR_ROWS);                  // [128 * 3]
    if (tid < RR_ROWS) { cell_row[tid] = -1; mask_row[tid] = 0.0f; }
    if (wave == 0) {            // (uniform) the P <= 64 centres of the workgroup are the first wave's lanes: the other seven waves
                                // used to run the same ~60 instructions (division, clamped loads) for nothing
        const long ptq = first_point + tid;
        const bool mine = tid < P && ptq < total_points;
        const long pq = mine ? ptq : first_point;
        const int b = (int)((unsigned)pq / (unsigned)a.npoints), n = (int)((unsigned)pq - (unsigned)b * (unsigned)a.npoints);
        int hc, wc;
        if (a.centre_hw) { hc = a.centre_hw[pq * 2 + 0]; wc = a.centre_hw[pq * 2 + 1]; }
        else { hc = n / a.W; wc = n - hc * a.W; }
        const float *c = a.xyz1_grid + (((long)b * a.H + hc) * a.W + wc) * 3;
        const float cx = c[0], cy = c[1], cz = c[2];
        if (mine) {
            chw[tid] = (hc << 16) | wc;
            cxyz[tid * 3 + 0] = cx; cxyz[tid * 3 + 1] = cy; cxyz[tid * 3 + 2] = cz;
            if (a.new_xyz) { a.new_xyz[ptq * 3 + 0] = cx; a.new_xyz[ptq * 3 + 1] = cy; a.new_xyz[ptq * 3 + 2] = cz; }     // :206
        }
    }
    stage_offsets(lds_off, a.group.random_hw, a.group.kernel_h, a.group.kernel_w, a.group.decoded_hw);      // ends with __syncthreads()
    RR_STAMP(1);
    {
        const float r2 = a.group.distance * a.group.distance;
        const int per_wave = 16 / K > 0 ? 16 / K : 1;                        // whole points in this wave's 16 rows (K = 8: 2, K = 16: 1)
        // the first window step of BOTH points is requested before either is judged: two round trips in flight, not in sequence
        auto uniform = [](float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); };
        int pb[2], ph[2], pw[2];
        float pc[2][3];
        bool live[2];
        RawSlot first[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int pp = (wave * 16) / K + j;                              // (K divides 16 here: see the launcher)
            const long ptw = first_point + pp;
            live[j] = j < per_wave && pp < P && ptw < total_points;          // (uniform)
            if (!live[j]) continue;                                          // (K = 16: the wave has ONE point -- no dead second fetch)
            pb[j] = (int)((unsigned)ptw / (unsigned)a.npoints);
            const int hwc = __builtin_amdgcn_readfirstlane(chw[pp]);
            ph[j] = div_stride(hwc >> 16, a.group.stride_h); pw[j] = div_stride(hwc & 0xffff, a.group.stride_w);
#pragma unroll
            for (int e = 0; e < 3; ++e) pc[j][e] = uniform(cxyz[pp * 3 + e]);
            first[j] = fetch_slot(grid_buffer(a.src_xyz + (size_t)pb[j] * a.H2 * a.W2 * 3), a.H2, a.W2, lds_off[lane < KT ? lane : 0], ph[j], pw[j],
                                  lane < KT);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if (!live[j]) continue;
            const int pp = (wave * 16) / K + j, b = pb[j];
            const float cx = pc[j][0], cy = pc[j][1], cz = pc[j][2];
            auto emit = [&](int slot, int hw) {
                cell_row[pp * K + slot] = (b * a.H2 + (hw >> 16)) * a.W2 + (hw & 0xffff);
                mask_row[pp * K + slot] = 1.0f;
            };
            int count = 0;
            if (!(pick_max(sq3(cx, cy, cz), ELO_EPS) <= ELO_EPS)) {          // valid centre (:62-70)
                const GridBuf gb = grid_buffer(a.src_xyz + (size_t)b * a.H2 * a.W2 * 3);
                count = wave_random_k_prefetched(gb, a.H2, a.W2, KT, K, lds_off, ph[j], pw[j], cx, cy, cz, r2, first[j], emit);
            }
            if (count + lane < K) { cell_row[pp * K + count + lane] = 0; mask_row[pp * K + count + lane] = 0.0f; }   // index (0,0,0), mask 0 (K <= 32: one trip)
        }
    }
    RR_STAMP(2);
    __syncthreads();
    RR_STAMP(3);
    // ---- this lane's row: gather [features | xyz difference]
    const int r = wave * 16 + i16, pi = small_div(r, K);
    const int cell = cell_row[r];
    const bool used = cell >= 0;                     // (rows of points beyond the end keep -1)
    const float mk = mask_row[r];
    const float c0 = cxyz[(pi < P ? pi : 0) * 3 + 0], c1 = cxyz[(pi < P ? pi : 0) * 3 + 1], c2 = cxyz[(pi < P ? pi : 0) * 3 + 2];
    ActPair in0[FPB / 2];
    uint4 dxyz;
    const bool keep = used && mk != 0.0f;
    {
        typedef typename std::conditional<F16, uint2, float4>::type Item;
        Item f[FPB];
        const long cc = used ? cell : 0;
        const Item *rf = reinterpret_cast<const Item *>(a.src_feat) + (cc * C >> 2) + kq;
#pragma unroll
        for (int j = 0; j < FPB; ++j) f[j] = rf[j * 4];
        const float *g = a.src_xyz + cc * 3;
        const float x = g[0], y = g[1], z = g[2];
        ws.prime();
        auto quad = [&](const Item &v) {
            const uint4 zq{0u, 0u, 0u, 0u};
            if constexpr (F16) return keep ? quad_of_halves(v) : zq;
            else return keep ? pack_quad<MODE>(v, bad) : zq;
        };
#pragma unroll
        for (int j = 0; j < FPB; ++j) put_quad(in0[j / 2], j % 2, quad(f[j]));
        const float4 d = kq == 0 ? float4{x * mk - c0, y * mk - c1, z * mk - c2, 0.0f} : float4{0.0f, 0.0f, 0.0f, 0.0f};
        dxyz = used ? pack_quad<MODE>(d, bad) : uint4{0u, 0u, 0u, 0u};
    }
    rr_bias_store(lds, tid, N1 + N2 + N3, bias_word);
    RR_STAMP(4);
    rr_barrier();
    RR_STAMP(5);
    const uint4 none{0u, 0u, 0u, 0u};
    ActPair h1[N1 / 32];
    auto emit0 = [&](int pass, int t, const f32x4 acc) { put_quad(h1[pass], t, pack_quad<MODE>(relu4(acc, a.layers[0].relu), bad)); };
    rr_layer<FPB / 2, true, N1 / 16, 0, MODE, (F16 ? ~0u : 0u)>(in0, dxyz, bt, bt + B1, bias, fetch, advance, emit0);
    RR_STAMP(6);
    float4 last[NOUT / 16];
    if constexpr (N3 == 0) {
        rr_layer<N1 / 32, false, N2 / 16, E0, MODE>(h1, none, bt + B1, nullptr, bias, fetch, advance, [&](int pass, int t, const f32x4 acc) {
            last[2 * pass + t] = relu4(acc, a.layers[1].relu);
        });
    } else {
        ActPair h2[N2 / 32];
        rr_layer<N1 / 32, false, N2 / 16, E0, MODE>(h1, none, bt + B1, bt + B2, bias, fetch, advance, [&](int pass, int t, const f32x4 acc) {
            put_quad(h2[pass], t, pack_quad<MODE>(relu4(acc, a.layers[1].relu), bad));
        });
        rr_layer<N2 / 32, false, NOUT / 16, E1, MODE>(h2, none, bt + B2, nullptr, bias, fetch, advance, [&](int pass, int t, const f32x4 acc) {
            last[2 * pass + t] = relu4(acc, a.layers[2].relu);
        });
    }
    RR_STAMP(7);
    if ((K == 8 || K == 16) && a.layers[N3 ? 2 : 1].relu) rr_pool_max_inwave<NOUT, F16>(last, mk, r, lane, K, first_point, total_points, a.out);
    else rr_pool_max<NOUT, F16>(lds, last, mk, r, wave, lane, K, P, first_point, total_points, a.out);
    RR_STAMP(8);
    report_violations<MODE>(bad, a.range_counter);
}

template <int FPB, int N1, int N2, int N3, bool F16, int MODE = MODE_SPLIT>
__global__ __launch_bounds__(RR_WAVES * 64, 4) void setconv_rr_kernel(const JobPair<elo_setconv_args> jobs)
{
    setconv_rr_body<FPB, N1, N2, N3, F16, MODE>(jobs.job[blockIdx.y], blockIdx.x, gridDim.x);
}

// Cost-volume stage 1 (from the select-k pre-pass's idx / mask) and stage 1 of the level's two set-upconvs (128 -> 64, in-kernel
// random-k) only share inputs (pwclo_model.py:242-250): ONE launch of register-resident workgroups -- the first n_cv run
// cv1_rr_body, the next n_sc set-conv job 0, the rest job 1 -- instead of two dependent launches that each leave most of the
// GPU idle at batch 1-3 (cv1_setconv_kernel is the same move for the tile kernels).  The bodies are the plain kernels': same bits.
template <int C, bool F16, int MODE>
__global__ __launch_bounds__(RR_WAVES * 64, 4) void cv1_setconv_rr_kernel(const elo_cv1_args a, const JobPair<elo_setconv_args> jobs,
                                                                         const unsigned n_cv, const unsigned n_sc)
{
    if (blockIdx.x < n_cv) {
        cv1_rr_body<C, F16, MODE>(a, blockIdx.x, n_cv);
    } else {
        const unsigned r = blockIdx.x - n_cv, j = r >= n_sc ? 1u : 0u;
        setconv_rr_body<4, 128, 64, 0, F16, MODE>(jobs.job[j], j ? r - n_sc : r, n_sc);
    }
}
// ================================================================ the same narrow set-conv, MLP on the matrix cores (round 4)
// setconv_small_kernel spends 1507 (6 -> 8 -> 8 -> 16) / 2756 (19 -> 16 -> 16 -> 32) vector instructions per wave, three
// quarters of them the row-per-lane FMAs of the MLP, and is issue-bound once the GPU is full (batch 8: 37 + 27 us, the two
// largest VALU kernels of a forward).  Here the grouping is unchanged -- half a wave per centre, a lane per window slot,
// ballot + popcount -- and leaves the 32 neighbour slots of the workgroup's 8 centres in LDS; then the wave's 64 rows
// (2 centres x 32 neighbours) run the MLP as FOUR 16-row blocks in the chain kernels' transposed form, D^T = W^T X^T:
// lane (j, kq) of a block gathers the channel quad kq of row j itself (a float4 / four halves of the neighbour's feature
// row; the geometry quad is [dx dy dz 0] in lane group 0), which IS the matrix cores' B operand; the layer's accumulator
// leaves the lane with four consecutive output channels of its row -- the next layer's operand after bias (the
// accumulator's initial value), ReLU and the hi / lo split -- and the masked max over the 32 rows of a centre is four DPP
// steps inside a 16-lane row plus one elementwise maximum of the centre's two row blocks.  No LDS tile, no barrier after
// the grouping, no W stream: the three layers' weights are <= 1.6 K floats, each lane builds its W^T fragments (<= 20
// registers) ONCE from the plain row-major weights (elo_dense.w_plain, already in [features | xyz difference] row order) and
// keeps them.  12 (6 -> 8 -> 8 -> 16: 9) matrix instructions per row block instead of ~270 (60) FMAs per row.
// fp32-class products whatever the launch's products mode (three products of hi / lo operands; the VALU kernel was plain
// fp32 in either mode).  The fp32-MFMA comparison build keeps setconv_small_kernel.
template <int CIN, int N1, int N2, int N3, int MODE>
// (the second launch bound caps the kernel at 256 registers: allowed 512 -- VGPRs + AGPRs -- the compiler selects the AGPR form
//  of the matrix instructions and pays 4 v_accvgpr_write + 4 v_accvgpr_read around every short chain: 128 of 1.66 k vector
//  instructions per wave here.  The kernel needs 103.)
__global__ __launch_bounds__(ELO_BLOCK, 2) void setconv_narrow_kernel(const elo_setconv_args a)
{
    constexpr int G = 32, PER_BLOCK = ELO_BLOCK / G, C = CIN - 3, CB3 = N3 / 16;
    static_assert((C == 16 || C == 3) && N1 <= 16 && N2 <= 16 && (N3 == 16 || N3 == 32), "the pyramid's two narrow set-conv layers");
    __shared__ int slot_hw[PER_BLOCK][G];
    const int tid = threadIdx.x;
    const int l64 = tid & 63, i16 = l64 & 15, kq = l64 >> 4;
    unsigned bad = 0, wbad = 0;
    // ---- (1) W^T fragments and bias quads, requested first (consumed after the grouping): lane (i16, kq) of a fragment holds
    // W[4kq + e][cb * 16 + i16], e = 0..3 (and, for the 32-k pair of the 19-channel input, W[16 + 4kq + e][..] as well)
    // (LK, LN: the layer's K and N, the template's widths -- the launcher picked the instance by them --, so the clamps and the
    //  zero tests fold wherever a whole quad is inside or outside)
    auto wq = [&](const elo_dense &L, const int LK, const int LN, int k0, int n) {   // rows k0 .. k0 + 3 of column n; 0 beyond (K, N); unconditional loads
        const int nc = n < LN ? n : LN - 1;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int k = k0 + e < LK ? k0 + e : LK - 1;
            v[e] = L.w_plain[k * LN + nc];
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (k0 + e < LK && n < LN) ? v[e] : 0.0f;
        return float4{v[0], v[1], v[2], v[3]};
    };
    auto bq = [&](const elo_dense &L, const int LN, int n0) {            // bias[n0 .. n0 + 3]
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = L.bias[n0 + e < LN ? n0 + e : LN - 1];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = n0 + e < LN ? v[e] : 0.0f;
        return float4{v[0], v[1], v[2], v[3]};
    };
    const float4 w1a = wq(a.layers[0], CIN, N1, 4 * kq, i16), w1b = C == 16 ? wq(a.layers[0], CIN, N1, 16 + 4 * kq, i16) : float4{0.0f, 0.0f, 0.0f, 0.0f};
    const float4 w2f = wq(a.layers[1], N1, N2, 4 * kq, i16);
    float4 w3f[CB3];
#pragma unroll
    for (int cb = 0; cb < CB3; ++cb) w3f[cb] = wq(a.layers[2], N2, N3, 4 * kq, cb * 16 + i16);
    const float4 b1 = bq(a.layers[0], N1, 4 * kq), b2 = bq(a.layers[1], N2, 4 * kq);
    float4 b3[CB3];
#pragma unroll
    for (int cb = 0; cb < CB3; ++cb) b3[cb] = bq(a.layers[2], N3, cb * 16 + 4 * kq);

    // ---- (2) grouping: half a wave per centre, as setconv_small_kernel
    const int g = tid / G, lane = tid % G, shift = (tid & 63) / G * G;
    const elo_group_spec &gs = a.group;
    const int KT = gs.kernel_h * gs.kernel_w, kW = gs.kernel_w, hh = gs.kernel_h / 2, hw2 = gs.kernel_w / 2;
    int off[SMALL_STEPS];
    const int *order = gs.decoded_hw ? gs.decoded_hw : gs.random_hw;
#pragma unroll
    for (int st = 0; st < SMALL_STEPS; ++st) off[st] = order[st * G + lane < KT ? st * G + lane : 0];
    const long total = (long)a.batch * a.npoints;
    const long first = (long)xcd_tile(blockIdx.x, gridDim.x) * PER_BLOCK;
    const long pt = first + g;
    const bool live = pt < total;
    const long ptc = live ? pt : total - 1;                 // dead groups shadow the last point and store nothing
    int b, n, hc, wc;
    split_point(ptc, a.npoints, b, n);
    if (a.centre_hw) {
        const int2 c2 = reinterpret_cast<const int2 *>(a.centre_hw)[ptc];
        hc = c2.x; wc = c2.y;
    } else { hc = n / a.W; wc = n - hc * a.W; }
    if (!gs.decoded_hw) {
#pragma unroll
        for (int st = 0; st < SMALL_STEPS; ++st) off[st] = ((off[st] / kW - hh) << 16) | ((off[st] % kW - hw2) & 0xffff);
    }
    const float *cp = a.xyz1_grid + (((long)b * a.H + hc) * a.W + wc) * 3;
    const float cx = cp[0], cy = cp[1], cz = cp[2];
    const float *grid2 = a.src_xyz + (size_t)b * a.H2 * a.W2 * 3;
    const int base_h = div_stride(hc, gs.stride_h), base_w = div_stride(wc, gs.stride_w);
    RawSlot raw[SMALL_STEPS];
#pragma unroll
    for (int st = 0; st < SMALL_STEPS; ++st) raw[st] = fetch_slot(grid2, a.H2, a.W2, off[st], base_h, base_w, st * G + lane < KT);
    if (live && lane < 3 && a.new_xyz) a.new_xyz[pt * 3 + lane] = lane == 0 ? cx : lane == 1 ? cy : cz;
    const float r2 = gs.distance * gs.distance;
    slot_hw[g][lane] = -1;
    int taken = 0;
    const bool centre_ok = !(pick_max(sq3(cx, cy, cz), ELO_EPS) <= ELO_EPS);
#pragma unroll
    for (int st = 0; st < SMALL_STEPS; ++st) {
        if (st > 0 && (st * G >= KT || __builtin_amdgcn_ballot_w64(taken < G) == 0)) break;     // as in setconv_small_kernel
        const Probe pr = judge(raw[st], cx, cy, cz, r2);
        const bool hit = pr.hit && centre_ok && taken < G;
        const unsigned long long mh = group_ballot<G>(hit, shift);
        const int slot = taken + __popcll(mh & ((1ull << lane) - 1ull));
        if (hit && slot < G) slot_hw[g][slot] = pr.hw;
        taken += __popcll(mh);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (gs.idx_out || gs.mask_out) {                         // (parity tests: the op's index / mask outputs)
        const int hw = slot_hw[g][lane];
        if (live && gs.idx_out) {
            int *o = gs.idx_out + (pt * G + lane) * 3;
            o[0] = hw >= 0 ? b : 0; o[1] = hw >= 0 ? hw >> 16 : 0; o[2] = hw >= 0 ? hw & 0xffff : 0;
        }
        if (live && gs.mask_out) gs.mask_out[pt * G + lane] = hw >= 0 ? 1.0f : 0.0f;
    }

    // ---- (3) the wave's four row blocks: rows (rb & 1) * 16 + i16 of centre 2 * wave + (rb >> 1); every gather goes out first
    const int wv = tid >> 6, f16 = a.feat_dtype == ELO_F16;
    auto uni = [](float v, int src) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src)); };
    float mrow[4], sx[4][3];
    uint4 ff[4];                                             // bits of the gathered features: the quad kq (C = 16: four floats, or four halves in .x / .y) or channels 0..2 (C = 3, as floats)
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) {
        const int hw = slot_hw[wv * 2 + (rb >> 1)][(rb & 1) * 16 + i16];
        const int bb = __builtin_amdgcn_readlane(b, (rb >> 1) * 32);
        mrow[rb] = hw >= 0 ? 1.0f : 0.0f;
        const long cell = hw >= 0 ? ((long)bb * a.H2 + (hw >> 16)) * a.W2 + (hw & 0xffff) : 0;       // empty slot: index (0,0,0)
        const float *sp = a.src_xyz + cell * 3;
        sx[rb][0] = sp[0]; sx[rb][1] = sp[1]; sx[rb][2] = sp[2];
        if constexpr (C == 16) {
            if (f16) {
                const uint2 h = reinterpret_cast<const uint2 *>(a.src_feat)[cell * 4 + kq];
                ff[rb] = uint4{h.x, h.y, 0u, 0u};
            } else ff[rb] = reinterpret_cast<const uint4 *>(a.src_feat)[cell * 4 + kq];
        } else {
            float v[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) v[c] = feat_load(a.src_feat, cell * 3 + c, f16);
            ff[rb] = uint4{__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), 0u};
        }
    }
    // the fragments as matrix-core operands (weights were vetted when they were packed: no range count for them)
    const uint4 q1a = pack_quad<MODE_SPLIT>(w1a, wbad), q1b = pack_quad<MODE_SPLIT>(w1b, wbad), w2t = pack_quad<MODE_SPLIT>(w2f, wbad);
    const WPair w1p{uint4{q1a.x, q1a.y, q1b.x, q1b.y}, uint4{q1a.z, q1a.w, q1b.z, q1b.w}};
    uint4 w3t[CB3];
#pragma unroll
    for (int cb = 0; cb < CB3; ++cb) w3t[cb] = pack_quad<MODE_SPLIT>(w3f[cb], wbad);
    float4 best[CB3];
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) {
        const float m = mrow[rb];
        const float ccx = uni(cx, (rb >> 1) * 32), ccy = uni(cy, (rb >> 1) * 32), ccz = uni(cz, (rb >> 1) * 32);
        const float dx = sx[rb][0] * m - ccx, dy = sx[rb][1] * m - ccy, dz = sx[rb][2] * m - ccz;
        f32x4 acc1[1][1] = {{f32x4{b1.x, b1.y, b1.z, b1.w}}};
        if constexpr (C == 16) {
            // k-blocks [features 0..15 | dx dy dz 0 ...]: one 32-k pair
            uint4 fq;
            const float4 fv = __builtin_bit_cast(float4, ff[rb]);
            if (f16) fq = m != 0.0f ? quad_of_halves(uint2{ff[rb].x, ff[rb].y}) : uint4{0u, 0u, 0u, 0u};
            else fq = pack_quad<MODE>(float4{fv.x * m, fv.y * m, fv.z * m, fv.w * m}, bad);
            const uint4 gq = pack_quad<MODE>(kq == 0 ? float4{dx, dy, dz, 0.0f} : float4{0.0f, 0.0f, 0.0f, 0.0f}, bad);
            const ActPair in[1] = {ActPair{uint4{fq.x, fq.y, gq.x, gq.y}, uint4{fq.z, fq.w, gq.z, gq.w}}};
            const WPair w[1] = {w1p};
            mma_pair<MODE, 1, 1, 1>(acc1, 0, in, w);
        } else {
            // one 16-k block: [f0 f1 f2 dx | dy dz 0 0 | 0 ...]
            const float4 fv = __builtin_bit_cast(float4, ff[rb]);
            const float4 v = kq == 0 ? float4{fv.x * m, fv.y * m, fv.z * m, dx} : kq == 1 ? float4{dy, dz, 0.0f, 0.0f}
                                                                                            : float4{0.0f, 0.0f, 0.0f, 0.0f};
            const uint4 in[1] = {pack_quad<MODE>(v, bad)}, w[1] = {q1a};
            mma_tail<MODE, 1, 1, 1, false>(acc1, 0, in, w);
        }
        f32x4 acc2[1][1] = {{f32x4{b2.x, b2.y, b2.z, b2.w}}};
        {
            const uint4 in[1] = {pack_quad<MODE>(relu4(acc1[0][0], a.layers[0].relu), bad)}, w[1] = {w2t};
            mma_tail<MODE, 1, 1, 1, false>(acc2, 0, in, w);
        }
        const uint4 h2[1] = {pack_quad<MODE>(relu4(acc2[0][0], a.layers[1].relu), bad)};
#pragma unroll
        for (int cb = 0; cb < CB3; ++cb) {
            f32x4 acc3[1][1] = {{f32x4{b3[cb].x, b3[cb].y, b3[cb].z, b3[cb].w}}};
            const uint4 w[1] = {w3t[cb]};
            mma_tail<MODE, 1, 1, 1, false>(acc3, 0, h2, w);
            // masked max over the block's 16 rows (:224-230): the last layer has a ReLU (launcher), so values >= 0 and the
            // maximum can be taken on the bit patterns; every lane of the 16-lane row ends up with it
            const float4 y = relu4(acc3[0][0], a.layers[2].relu);
            const float q[4] = {y.x * m, y.y * m, y.z * m, y.w * m};
            float o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                int v = __float_as_int(q[e]);
                v = dpp_imax_all<0xb1>(v);
                v = dpp_imax_all<0x4e>(v);
                v = dpp_imax_all<0x141>(v);
                v = dpp_imax_all<0x140>(v);
                o[e] = __int_as_float(v);
            }
            if (rb & 1) {                                    // the centre's second half: combine and store (lane group kq: channels cb*16 + 4kq ..)
                const long ptg = first + wv * 2 + (rb >> 1);
                const float4 r4{fmaxf(best[cb].x, o[0]), fmaxf(best[cb].y, o[1]), fmaxf(best[cb].z, o[2]), fmaxf(best[cb].w, o[3])};
                if (i16 == 0 && ptg < total) feat_store4(a.out, ptg * N3 + cb * 16 + 4 * kq, r4, f16);
            } else best[cb] = float4{o[0], o[1], o[2], o[3]};
        }
    }
    report_violations<MODE>(bad, a.range_counter);
}

#endif   // !ELO_DENSE_F32

// ================================================================ cost volume, stage 2
// LDS columns: [0,64) grouped cost, [64,128) xyz-encoding, [128,128+C) feat1, [192,208) xyz_cat;
// sum_cost0 reads [0,128+C) and writes [64,192); sum_cost1 maps [64,192) -> [64,128).
template <int TILE, int MODE>
__global__ __launch_bounds__(FUSED_BLOCK, TILE == 32 ? ELO_TILE32_WAVES : ELO_TILE_WAVES) void cv2_kernel(const elo_cv2_args a, const int S)   // (32 rows at 6 waves: 4 VGPRs spilled to scratch)
{
    extern __shared__ __align__(16) float lds[];
    float *act = lds;
    const TileMeta meta = tile_meta(lds, TILE, S);
    const int tid = threadIdx.x;
    const int K = a.K, P = TILE / K, C = a.C, f16 = a.feat_dtype == ELO_F16;
    unsigned bad = 0;
    const long total_points = (long)a.batch * a.npoints;
    const long first_point = (long)xcd_tile(blockIdx.x, gridDim.x) * P;
    if (first_point >= total_points) return;
    if (a.group.random_hw) {                          // random-k of the warped cloud on itself (:106-108)
        int *lds_off = reinterpret_cast<int *>(meta.cxyz + 128);
        group_tile<TILE, false>(meta, lds_off, nullptr, a.group, first_point, total_points, a.npoints, P, K, a.xyz1, a.H,
                                a.W, nullptr, a.xyz1, a.H, a.W, nullptr);
    } else {
        load_meta<TILE>(meta, first_point, total_points, P, K, a.idx, a.mask, a.H, a.W);
        __syncthreads();
    }
    const int Cp = ceil16(C), XYZ = 192;
    {
        auto cell_of = [&](int row) { return (long)meta.cell[row]; };
        auto centre_of = [&](int row) { return meta.cell[row] >= 0 ? first_point + small_div(row, K) : -1L; };
        uint4 rc[SEG_ITEMS], rf[SEG_ITEMS];
        seg_load<TILE>(rc, a.cost, 64, f16, cell_of);                                 // grouped cost * mask  :110
        seg_load<TILE>(rf, a.feat1, C, f16, centre_of);                               // centre features      :115
        GeoRow gr;
        const bool grow = tid < TILE;
        const int gcell = grow ? meta.cell[tid] : -1;
        if (grow) gr = geo_load(a.xyz1 + (gcell >= 0 ? first_point + small_div(tid, K) : 0) * 3, a.xyz1 + (long)(gcell >= 0 ? gcell : 0) * 3);
        seg_store<TILE, MODE>(act, S, 0, 64, f16, rc, cell_of, [&](int row) { return meta.mask[row] != 0.0f; }, bad);
        seg_store<TILE, MODE>(act, S, 128, C, f16, rf, centre_of, [](int) { return true; }, bad);
        zero_cols<TILE, MODE>(act, S, 128 + C, 128 + Cp);
        if (grow) geo_store<MODE>(act, tid * S + XYZ, gr, meta.mask[tid], gcell >= 0, bad);   // 10-channel geometry + 6 zeros  :111-120
    }
    Pre<Sub<TILE, 128>::TPW> p128;
    Pre<Sub<TILE, 64>::TPW> p64;
    prefetch<TILE, Sub<TILE, 64>::TPW, MODE>(a.xyz_enc, p64);
    __syncthreads();
    dense_pf<TILE, 64, 128, MODE>(act, S, XYZ, to_tile(64), a.xyz_enc, p64, &a.sum_cost0, &p128, bad);   // -> enc at [64,128)       :123-126
    dense_pf<TILE, 128, 64, MODE>(act, S, 0, to_tile(64), a.sum_cost0, p128, &a.sum_cost1, &p64, bad);   // [grouped | enc | feat1] -> [64,192)   :129-135
    dense_pf<TILE, 64, 0, MODE>(act, S, 64, to_pool(64), a.sum_cost1, p64, nullptr, nullptr, bad);       // -> 64 logits at [64,128)
    pool_masked_softmax(act, S, 64, 0, meta, P, K, first_point, total_points, a.out, f16);   // :137-146
    report_violations<MODE>(bad, a.range_counter);
}

// A 16-row tile halves the serial work per workgroup and doubles their number: take it whenever the
// K rows of a point fit and 32-row tiles would leave most of the 256 CUs without a workgroup.
inline long small_tile_threshold() { return tuning().small_tile_units; }
inline bool small_tile(long units_at_32, int K) { return K <= 16 && units_at_32 < small_tile_threshold(); }

int check_dense(const elo_dense &L, int K, int N, const char *who, const char *name)
{
    if (!L.w_packed || !L.bias) return fail(ELO_ERR_ARG, "%s: layer %s has null weights", who, name);
    if (L.K != K || (N > 0 && L.N != N))
        return fail(ELO_ERR_ARG, "%s: layer %s is %dx%d, expected %dx%d", who, name, L.K, L.N, K, N);
    if (L.N <= 0 || L.N > 128) return fail(ELO_ERR_LIMIT, "%s: layer %s width %d outside 1..128", who, name, L.N);
    if (L.products != ELO_PRODUCTS_SPLIT && L.products != ELO_PRODUCTS_HALF)
        return fail(ELO_ERR_ARG, "%s: layer %s: products must be ELO_PRODUCTS_SPLIT or ELO_PRODUCTS_HALF", who, name);
#ifdef ELO_DENSE_F32
    if (L.products != ELO_PRODUCTS_SPLIT) return fail(ELO_ERR_ARG, "%s: this library was built with -DELO_DENSE_F32 (fp32 MFMA only)", who);
#endif
    return ELO_OK;
}

int check_dtype(int feat_dtype, const char *who)
{
    if (feat_dtype != ELO_F32 && feat_dtype != ELO_F16) return fail(ELO_ERR_ARG, "%s: feat_dtype must be ELO_F32 or ELO_F16", who);
    return ELO_OK;
}

// the range check is a process-wide switch (elo_tuning.range_check / elo_range_check)
int &range_check_flag() { return tuning().range_check; }

// the products mode is a property of the LAUNCH (the kernels are instantiated per mode): every layer must agree
template <typename... Rest>
int products_mode(const char *who, int *mode, const elo_dense &first, const Rest &...rest)
{
    const int m = first.products;
    for (const elo_dense *L : {&rest...})
        if (L->products != m) return fail(ELO_ERR_ARG, "%s: the layers of one launch must share one products mode", who);
    *mode = m == ELO_PRODUCTS_HALF ? MODE_HALF : range_check_flag() ? MODE_CHECKED : MODE_SPLIT;
    return ELO_OK;
}

int products_mode(const char *who, int *mode, const elo_dense *layers, int n, const elo_dense *layers2 = nullptr, int n2 = 0)
{
    const int m = layers[0].products;
    for (int l = 0; l < n; ++l)
        if (layers[l].products != m) return fail(ELO_ERR_ARG, "%s: the layers of one launch must share one products mode", who);
    for (int l = 0; l < n2; ++l)
        if (layers2[l].products != m) return fail(ELO_ERR_ARG, "%s: the layers of one launch must share one products mode", who);
    *mode = m == ELO_PRODUCTS_HALF ? MODE_HALF : range_check_flag() ? MODE_CHECKED : MODE_SPLIT;
    return ELO_OK;
}

size_t tile_lds_bytes(int rows, int S, int KT = 0, bool select = false, int K = 0)
{
    // select-k: per wave the two [KT] arrays of its LDS form, or 128 words for the register form's small-K rank path
    return sizeof(float) * ((size_t)rows * S + 64 + 96 + 32 + ((KT + 3) & ~3) + (select ? (size_t)FUSED_WAVES * select_scratch_words(KT, K) : 0));
}

// in-kernel grouping: validate the spec the way elo_fused_conv_*_k validates its attributes
int check_group(const elo_group_spec &g, int H2, int W2, size_t lds_bytes, const char *who)
{
    if (!g.random_hw) return ELO_OK;
    if (g.kernel_h <= 0 || g.kernel_w <= 0 || !(g.distance > 0.0f) || g.stride_h <= 0 || g.stride_w <= 0)
        return fail(ELO_ERR_ARG, "%s: bad grouping attributes", who);
    if (g.kernel_w / 2 > W2)
        return fail(ELO_ERR_LIMIT, "%s: kernel_size_W/2 = %d exceeds the queried width %d (single wrap)", who, g.kernel_w / 2, W2);
    if (H2 >= 32768 || W2 >= 65536) return fail(ELO_ERR_LIMIT, "%s: queried grid larger than 32767 x 65535", who);
    if ((long)H2 * W2 * 12 >= 0x7fffffffL) return fail(ELO_ERR_LIMIT, "%s: queried grid beyond 2 GB per batch element (buffer addressing)", who);
    if (lds_bytes > 64 * 1024)
        return fail(ELO_ERR_LIMIT, "%s: window %dx%d needs %zu bytes of LDS for in-kernel grouping", who, g.kernel_h, g.kernel_w, lds_bytes);
    return ELO_OK;
}

#define ELO_REQUIRE(cond, who, what) \
    do { if (!(cond)) return fail(ELO_ERR_ARG, "%s: %s", who, what); } while (0)

}  // namespace
}  // namespace elo

using namespace elo;

static inline int pad16(int x) { return (x + 15) & ~15; }

// KERNEL<TILE, MODE> for TILE in {32, 16}, picked by (tile16, mode); a -DELO_DENSE_F32 library has no fp16-product kernels
#ifdef ELO_DENSE_F32
#define ELO_PICK(KERNEL, tile16, mode, CALL)                                                              \
    do {                                                                                                  \
        if ((mode) == MODE_CHECKED) { if (tile16) CALL((KERNEL<16, MODE_CHECKED>)); else CALL((KERNEL<32, MODE_CHECKED>)); } \
        else { if (tile16) CALL((KERNEL<16, MODE_SPLIT>)); else CALL((KERNEL<32, MODE_SPLIT>)); }         \
    } while (0)
#else
#define ELO_PICK(KERNEL, tile16, mode, CALL)                                                              \
    do {                                                                                                  \
        if ((mode) == MODE_HALF) { if (tile16) CALL((KERNEL<16, MODE_HALF>)); else CALL((KERNEL<32, MODE_HALF>)); } \
        else if ((mode) == MODE_CHECKED) { if (tile16) CALL((KERNEL<16, MODE_CHECKED>)); else CALL((KERNEL<32, MODE_CHECKED>)); } \
        else { if (tile16) CALL((KERNEL<16, MODE_SPLIT>)); else CALL((KERNEL<32, MODE_SPLIT>)); }         \
    } while (0)
#endif

// CALL_T(MODE) for the products mode of a launch (the heterogeneous launch, mlp_sv_kernel)
#ifdef ELO_DENSE_F32
#define ELO_PICK_MODE(M_, CALL_T) do { if ((M_) == MODE_CHECKED) CALL_T(MODE_CHECKED); else CALL_T(MODE_SPLIT); } while (0)
#else
#define ELO_PICK_MODE(M_, CALL_T) do { if ((M_) == MODE_HALF) CALL_T(MODE_HALF); else if ((M_) == MODE_CHECKED) CALL_T(MODE_CHECKED); else CALL_T(MODE_SPLIT); } while (0)
#endif

// column budget of an in-place chain starting from `width` input columns
static int chain_cols(const elo_dense *layers, int n_layers, int width)
{
    int cols = pad16(width);
    for (int l = 0; l < n_layers; ++l) cols = cols > pad16(layers[l].N) ? cols : pad16(layers[l].N);
    return cols;
}

static int check_setconv(const elo_setconv_args *a, const char *who)
{
    ELO_REQUIRE(a, who, "null argument block");
    ELO_REQUIRE(a->batch >= 0 && a->npoints > 0 && a->K > 0 && a->H2 > 0 && a->W2 > 0 && a->C >= 0, who, "bad sizes");
    if (a->K > 32) return fail(ELO_ERR_LIMIT, "%s: K = %d exceeds the 32-row tile", who, a->K);
    ELO_REQUIRE(a->n_layers >= 1 && a->n_layers <= ELO_MAX_CHAIN, who, "1..3 layers");
    ELO_REQUIRE(a->src_xyz && (a->src_feat || a->C == 0) && a->out, who, "null tensor pointer");
    ELO_REQUIRE(a->group.random_hw || (a->idx && a->mask), who, "neither idx/mask nor a grouping spec");
    ELO_REQUIRE((a->xyz1_grid && a->H > 0 && a->W > 0 && (a->centre_hw || a->group.random_hw)) || a->centre_xyz, who, "no centre source");
    ELO_REQUIRE(!a->group.random_hw || (a->xyz1_grid && a->H > 0 && a->W > 0), who, "in-kernel grouping needs xyz1_grid");
    // the window walk maps a centre (hc, wc) of the H x W grid to (hc / stride_h, wc / stride_w) of the queried H2 x W2 grid and
    // wraps a column ONCE: a queried grid smaller than ceil(H / stride_h) x ceil(W / stride_w) would send the raw-pointer
    // kernels out of bounds (fetch_slot has no final clamp).  centre_hw CONTENTS stay a documented precondition (h < H, w < W).
    if (a->group.random_hw && a->group.stride_h > 0 && a->group.stride_w > 0 &&
        ((a->H - 1) / a->group.stride_h >= a->H2 || (a->W - 1) / a->group.stride_w >= a->W2))
        return fail(ELO_ERR_ARG, "%s: the queried grid %dx%d is smaller than the centres' grid %dx%d over the strides %dx%d", who, a->H2, a->W2,
                    a->H, a->W, a->group.stride_h, a->group.stride_w);
    if (int rc = check_dtype(a->feat_dtype, who)) return rc;
    int width = 3 + a->C;
    for (int l = 0; l < a->n_layers; ++l) {
        if (int rc = check_dense(a->layers[l], width, 0, who, "mlp")) return rc;
        width = a->layers[l].N;
    }
    return ELO_OK;
}

// how a tile kernel is launched: LDS row stride, tile height, workgroups (per job), LDS bytes, products mode
struct TilePlan { int S; bool t16; long units; size_t lds; int mode; };

static bool same_shape(const elo_setconv_args *a, const elo_setconv_args *b)
{
    if (a->batch != b->batch || a->npoints != b->npoints || a->K != b->K || a->C != b->C || a->n_layers != b->n_layers ||
        a->H2 != b->H2 || a->W2 != b->W2 || a->group.kernel_h != b->group.kernel_h || a->group.kernel_w != b->group.kernel_w ||
        (a->group.random_hw == nullptr) != (b->group.random_hw == nullptr) || a->feat_dtype != b->feat_dtype)
        return false;
    for (int l = 0; l < a->n_layers; ++l)
        if (a->layers[l].K != b->layers[l].K || a->layers[l].N != b->layers[l].N) return false;
    return true;
}

// the tile-kernel launch of one or two (same-shape, already checked) set-conv jobs
static int plan_setconv(const elo_setconv_args *a, const elo_setconv_args *b, TilePlan *p, const char *who)
{
    const long points = (long)a->batch * a->npoints;
    if (points >= 0x7fffffffL) return fail(ELO_ERR_LIMIT, "%s: batch * npoints beyond 2^31", who);
    p->S = row_stride(chain_cols(a->layers, a->n_layers, 3 + a->C));
    const int P32 = 32 / a->K, P16 = a->K <= 16 ? 16 / a->K : 1;
    const long u32 = (points + P32 - 1) / P32, u16 = (points + P16 - 1) / P16;
    int mode_b = 0;
    if (int rc = products_mode(who, &p->mode, a->layers, a->n_layers)) return rc;
    if (b) {
        if (int rc = products_mode(who, &mode_b, b->layers, b->n_layers)) return rc;
        if (p->mode != mode_b) return fail(ELO_ERR_ARG, "%s: the two jobs of a paired launch must share one products mode", who);
    }
    p->t16 = small_tile(u32 * (b ? 2 : 1), a->K);
    p->units = p->t16 ? u16 : u32;
    const int KT = a->group.random_hw ? a->group.kernel_h * a->group.kernel_w : 0;
    p->lds = tile_lds_bytes(p->t16 ? 16 : 32, p->S, KT, false);
    return check_group(a->group, a->H2, a->W2, p->lds, who);
}

extern "C" int elo_dense_f32(void)
{
#ifdef ELO_DENSE_F32
    return 1;
#else
    return 0;
#endif
}

#ifdef ELO_CV1_CLOCK
extern "C" int elo_debug_cv1_clock(unsigned long long *out16)
{
    return hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_cv1_clock), 24 * sizeof(unsigned long long)) == hipSuccess ? 0 : -3;
}
#endif

extern "C" int elo_range_check(int enable)
{
    const int prev = range_check_flag();
    if (enable >= 0) range_check_flag() = enable ? 1 : 0;
    return prev;
}

extern "C" int elo_range_violations(unsigned long long *count, elo_stream_t stream)
{
    const char *who = "elo_range_violations";
    ELO_REQUIRE(count, who, "null count");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(range_take_kernel, dim3(1), dim3(1), 0, s);
    if (hipMemcpyFromSymbolAsync(count, HIP_SYMBOL(g_range_snapshot), sizeof(*count), 0, hipMemcpyDeviceToHost, s) != hipSuccess ||
        hipStreamSynchronize(s) != hipSuccess)
        return fail(ELO_ERR_LAUNCH, "%s: %s", who, hipGetErrorString(hipGetLastError()));
    return ELO_OK;
}

// Which kernel FORM an entry point takes is elo_tuning (include/elo.h; elo_set_tuning): the hooks below are single-field
// shorthands with a "-1 = back to elo_set_tuning's value" convention for the tests that flip one form and restore it.
extern "C" int elo_debug_cv1_rr(int on)           // (the name is round 3's: the switch covers all four register-resident kernels)
{
    const int prev = tuning().chain_forms;
    tuning().chain_forms = on >= 0 ? (on != 0) : tuning_base().chain_forms;
    return prev;
}
static bool cv1_rr_on() { return tuning().chain_forms != 0; }
extern "C" int elo_debug_rr_rows(long setconv_rows, long mlp_rows)
{
    tuning().setconv_chain_rows = setconv_rows >= 0 ? setconv_rows : tuning_base().setconv_chain_rows;
    tuning().mlp_chain_rows = mlp_rows >= 0 ? mlp_rows : tuning_base().mlp_chain_rows;
    return ELO_OK;
}
static long setconv_rr_rows(int batch)
{
    const long t = tuning().setconv_chain_rows;
    (void)batch;                                   // (below ELO_THROUGHPUT_BATCH: 100000 until round 5 -- profiles/r05_batch1_regimes.txt)
    return t >= 0 ? t : 20000L;
}
static long mlp_rr_rows(int batch_hint)
{
    const long t = tuning().mlp_chain_rows;
    return t >= 0 ? t : batch_hint >= ELO_THROUGHPUT_BATCH ? 2048L : 8192L;
}
extern "C" int elo_debug_narrow_mfma(int on)
{
    const int prev = tuning().narrow_mfma;
    tuning().narrow_mfma = on >= 0 ? (on ? 1 : 0) : tuning_base().narrow_mfma;
    return prev;
}
static bool narrow_mfma_on() { return tuning().narrow_mfma != 0; }
// launches of the register-resident kernels since the last reset: [cv1_rr, cv2_rr, setconv_rr, mlp2_rr].  The parity tests
// of the chain regime assert through it that the kernel under test is the one that ran.
static std::atomic<unsigned long long> g_rr_launches[4];
extern "C" int elo_debug_rr_launches(unsigned long long *counts4, int reset)
{
    for (int i = 0; i < 4; ++i) {
        if (counts4) counts4[i] = g_rr_launches[i].load();
        if (reset) g_rr_launches[i].store(0);
    }
    return ELO_OK;
}
static std::atomic<unsigned long long> g_chain_pair_launches; // cv1_setconv_rr_kernel (elo_cv_stage1_setconv_chain)
extern "C" int elo_debug_chain_pair_launches(unsigned long long *count, int reset)
{
    if (count) *count = g_chain_pair_launches.load();
    if (reset) g_chain_pair_launches.store(0);
    return ELO_OK;
}
static std::atomic<unsigned long long> g_sv_ride_launches;    // mlp_sv_kernel (elo_mlp_args.sv_*)
extern "C" int elo_debug_sv_ride_launches(unsigned long long *count, int reset)
{
    if (count) *count = g_sv_ride_launches.load();
    if (reset) g_sv_ride_launches.store(0);
    return ELO_OK;
}
// launches of the narrow set-conv forms since the last reset: [setconv_narrow_kernel (matrix cores), setconv_small_kernel (VALU)]
static std::atomic<unsigned long long> g_narrow_launches[2];
extern "C" int elo_debug_narrow_launches(unsigned long long *counts2, int reset)
{
    for (int i = 0; i < 2; ++i) {
        if (counts2) counts2[i] = g_narrow_launches[i].load();
        if (reset) g_narrow_launches[i].store(0);
    }
    return ELO_OK;
}

// The register-resident form (setconv_rr_kernel) is taken for the model's wide shapes -- 64 feature channels, layers
// 128 -> 64 (the set-upconvs: shape 1), 64 -> 64 -> 128 (2) or 128 -> 64 -> 64 (3) -- with in-kernel grouping, from
// ELO_SETCONV_RR_ROWS rows per launch on (regimes: elo_mlp_fused2).  0 = the tile kernel.  
static int setconv_chain_shape(const elo_setconv_args *a, const elo_setconv_args *b, int mode)
{
#ifdef ELO_DENSE_F32
    return 0;
#else
    const long points = (long)a->batch * a->npoints;
    const int nl = a->n_layers, n1 = a->layers[0].N, n2 = nl >= 2 ? a->layers[1].N : 0, n3 = nl == 3 ? a->layers[2].N : 0;
    const int shape = (nl == 2 && n1 == 128 && n2 == 64) ? 1 : (nl == 3 && n1 == 64 && n2 == 64 && n3 == 128) ? 2
                    : (nl == 3 && n1 == 128 && n2 == 64 && n3 == 64) ? 3 : 0;
    const int KT = a->group.kernel_h * a->group.kernel_w;
    const bool narrow = !b && a->K == 32 && nl == 3 && a->layers[0].w_plain && a->layers[1].w_plain && a->layers[2].w_plain;   // (the VALU / narrow kernels)
    if (!shape || narrow || !a->group.random_hw || !(mode == MODE_SPLIT || mode == MODE_HALF) || !cv1_rr_on() || a->C != 64 ||
        !(a->K == 8 || a->K == 16 || a->K == 32) || KT > 512 || a->group.idx_out || a->group.mask_out ||
        points * a->K * (b ? 2 : 1) < setconv_rr_rows(a->batch))
        return 0;
    return shape;
#endif
}

// the pre-grouped cost-volume calls take the chain kernels (cv1_rr_kernel / cv2_rr_kernel) in the plain products modes
static bool cv_chain(int C, int mode)
{
#ifdef ELO_DENSE_F32
    return false;
#else
    return (mode == MODE_SPLIT || mode == MODE_HALF) && cv1_rr_on() && (C == 16 || C == 32 || C == 64);
#endif
}

extern "C" int elo_setconv_fused2(const elo_setconv_args *a, const elo_setconv_args *b, elo_stream_t stream)
{
    const char *who = "elo_setconv_fused";
    if (int rc = check_setconv(a, who)) return rc;
    if (b) {
        if (int rc = check_setconv(b, who)) return rc;
        if (!same_shape(a, b)) return fail(ELO_ERR_ARG, "%s: the two jobs of a paired launch must have the same shape", who);
    }
    const long points = (long)a->batch * a->npoints;
    if (points >= 0x7fffffffL) return fail(ELO_ERR_LIMIT, "%s: batch * npoints beyond 2^31", who);
    if (points == 0) return ELO_OK;
    hipStream_t s = (hipStream_t)stream;
    if (!b && a->group.random_hw && a->K == 32 && a->n_layers == 3 && a->layers[0].w_plain && a->layers[1].w_plain &&
        a->layers[2].w_plain && a->layers[2].relu && a->group.kernel_h * a->group.kernel_w <= SMALL_STEPS * 32) {   // narrow chains: wave-per-point VALU kernel
        const int cin = 3 + a->C, n1 = a->layers[0].N, n2 = a->layers[1].N, n3 = a->layers[2].N;
        const unsigned grid = (unsigned)((points + 7) / 8);
        if (int rc = check_group(a->group, a->H2, a->W2, 0, who)) return rc;
#ifndef ELO_DENSE_F32
        if (narrow_mfma_on()) {                              // round 4: the same kernel with its MLP on the matrix cores
            const bool checked = range_check_flag();
            // (the 6 -> 8 -> 8 -> 16 layer stays on the VALU kernel: on the matrix cores it measured 64 us against 36.5 us at batch 8 --
            //  240 FMAs per row, the kernel is the latency chain of its grouping times its occupancy: profiles/r04_ab_narrow.txt)
            if (cin == 19 && n1 == 16 && n2 == 16 && n3 == 32) {
                if (checked) hipLaunchKernelGGL((setconv_narrow_kernel<19, 16, 16, 32, MODE_CHECKED>), dim3(grid), dim3(ELO_BLOCK), 0, s, *a);
                else hipLaunchKernelGGL((setconv_narrow_kernel<19, 16, 16, 32, MODE_SPLIT>), dim3(grid), dim3(ELO_BLOCK), 0, s, *a);
                ++g_narrow_launches[0];
                return check_launch(who);
            }
        }
#endif
        if (cin == 6 && n1 == 8 && n2 == 8 && n3 == 16) {
            hipLaunchKernelGGL((setconv_small_kernel<6, 8, 8, 16>), dim3(grid), dim3(ELO_BLOCK), 0, s, *a);
            ++g_narrow_launches[1];
            return check_launch(who);
        }
        if (cin == 19 && n1 == 16 && n2 == 16 && n3 == 32) {
            hipLaunchKernelGGL((setconv_small_kernel<19, 16, 16, 32>), dim3(grid), dim3(ELO_BLOCK), 0, s, *a);
            ++g_narrow_launches[1];
            return check_launch(who);
        }
    }
    TilePlan plan;
    if (int rc = plan_setconv(a, b, &plan, who)) return rc;
    JobPair<elo_setconv_args> pair;
    pair.job[0] = *a;
    pair.job[1] = b ? *b : *a;
#ifndef ELO_DENSE_F32
    if (const int shape = setconv_chain_shape(a, b, plan.mode)) {   // the register-resident form (setconv_rr_kernel)
        const int P = RR_ROWS / a->K;
        const dim3 rgrid((unsigned)((points + P - 1) / P), b ? 2u : 1u);
        const bool f16 = a->feat_dtype == ELO_F16;
#define RRS(N1_, N2_, N3_)                                                                                                              \
        do {                                                                                                                        \
            if (f16 && plan.mode == MODE_HALF) hipLaunchKernelGGL((setconv_rr_kernel<4, N1_, N2_, N3_, true, MODE_HALF>), rgrid, dim3(RR_WAVES * 64), RR_LDS_BYTES, s, pair);  \
            else if (f16) hipLaunchKernelGGL((setconv_rr_kernel<4, N1_, N2_, N3_, true, MODE_SPLIT>), rgrid, dim3(RR_WAVES * 64), RR_LDS_BYTES, s, pair);  \
            else if (plan.mode == MODE_HALF) hipLaunchKernelGGL((setconv_rr_kernel<4, N1_, N2_, N3_, false, MODE_HALF>), rgrid, dim3(RR_WAVES * 64), RR_LDS_BYTES, s, pair);     \
            else hipLaunchKernelGGL((setconv_rr_kernel<4, N1_, N2_, N3_, false, MODE_SPLIT>), rgrid, dim3(RR_WAVES * 64), RR_LDS_BYTES, s, pair);     \
        } while (0)
        if (shape == 1) RRS(128, 64, 0);
        else if (shape == 2) RRS(64, 64, 128);
        else RRS(128, 64, 64);
#undef RRS
        ++g_rr_launches[2];
        return check_launch(who);
    }
#endif
    const dim3 grid((unsigned)plan.units, b ? 2u : 1u);
    const size_t lds = plan.lds;
    const int S = plan.S;
#define CALL(k) hipLaunchKernelGGL(k, grid, dim3(FUSED_BLOCK), lds, s, pair, S)
    ELO_PICK(setconv_kernel, plan.t16, plan.mode, CALL);
#undef CALL
    return check_launch(who);
}

extern "C" int elo_setconv_fused(const elo_setconv_args *a, elo_stream_t stream)
{
    return elo_setconv_fused2(a, nullptr, stream);
}

static int check_mlp(const elo_mlp_args *a, const char *who, int *in_width)
{
    ELO_REQUIRE(a, who, "null argument block");
    ELO_REQUIRE(a->rows >= 0 && a->n_sources >= 1 && a->n_sources <= 3, who, "bad sizes");
    ELO_REQUIRE(a->n_layers >= 1 && a->n_layers <= ELO_MAX_CHAIN && a->out, who, "1..3 layers and an output");
    if (int rc = check_dtype(a->feat_dtype, who)) return rc;
    int width = 0;
    for (int s = 0; s < a->n_sources; ++s) {
        ELO_REQUIRE(a->src[s] && a->src_width[s] > 0, who, "null / empty source");
        width += a->src_width[s];
    }
    *in_width = width;
    for (int l = 0; l < a->n_layers; ++l) {
        if (int rc = check_dense(a->layers[l], width, 0, who, "mlp")) return rc;
        width = a->layers[l].N;
    }
    ELO_REQUIRE(!a->clear_scratch || (a->clear_xyz && a->clear_cells > 0 && a->clear_images > 0 && a->clear_C >= 0 &&
                                      (a->clear_feat || a->clear_C == 0) && (a->feat_dtype != ELO_F16 || a->clear_C % 2 == 0)),
                who, "incomplete clear_* side job");
    ELO_REQUIRE(!a->clear_scratch || a->clear_cells * (a->clear_C > 3 ? a->clear_C : 3) < (1l << 31), who, "clear_* side job: more than 2^31 words");
    ELO_REQUIRE(a->n_layers2 >= 0 && a->n_layers2 <= ELO_MAX_CHAIN, who, "0..3 second-stage layers");
    if (a->n_layers2 > 0) {
        ELO_REQUIRE(a->out2 && a->w_before >= 0 && a->w_after >= 0 && (a->before || a->w_before == 0) &&
                    (a->after || a->w_after == 0), who, "incomplete second stage");
        ELO_REQUIRE(width % 4 == 0, who, "a second stage needs a first-stage output width that is a multiple of 4");
        width += a->w_before + a->w_after;
        for (int l = 0; l < a->n_layers2; ++l) {
            if (int rc = check_dense(a->layers2[l], width, 0, who, "mlp stage 2")) return rc;
            width = a->layers2[l].N;
        }
    }
    return ELO_OK;
}

// column budget of a (possibly two-stage) row-wise MLP tile
static int mlp_cols(const elo_mlp_args *a, int in_width)
{
    int cols = chain_cols(a->layers, a->n_layers, in_width);
    if (a->n_layers2 > 0) {
        const int n1 = a->layers[a->n_layers - 1].N, in2 = a->w_before + n1 + a->w_after;
        const int c2 = chain_cols(a->layers2, a->n_layers2, in2);
        cols = cols > c2 ? cols : c2;
    }
    return cols;
}

// everything elo_mlp_fused2 decides before it launches
struct MlpPlan { int S, mode, C; bool t16, chain; int sv_tiles; };     // sv_tiles: row tiles per batch element of the softmax_valid ride (0: none)

static int plan_mlp(const elo_mlp_args *a, const elo_mlp_args *b, MlpPlan *p, const char *who)
{
    int in_width = 0, in_width_b = 0;
    if (int rc = check_mlp(a, who, &in_width)) return rc;
    if (b) {
        if (int rc = check_mlp(b, who, &in_width_b)) return rc;
        bool same = a->rows == b->rows && in_width == in_width_b && a->n_layers == b->n_layers &&
                    a->n_layers2 == b->n_layers2 && a->w_before == b->w_before && a->w_after == b->w_after &&
                    a->feat_dtype == b->feat_dtype;
        for (int l = 0; same && l < a->n_layers; ++l) same = a->layers[l].N == b->layers[l].N;
        for (int l = 0; same && l < a->n_layers2; ++l) same = a->layers2[l].N == b->layers2[l].N;
        if (!same) return fail(ELO_ERR_ARG, "%s: the two jobs of a paired launch must have the same shape", who);
    }
    if (b && b->clear_scratch) return fail(ELO_ERR_ARG, "%s: the clear_* side job belongs to the first job of a pair", who);
    if (b && (b->sv_scratch || b->sv_feature)) return fail(ELO_ERR_ARG, "%s: the sv_* fields belong to the first job of a pair", who);
    p->S = row_stride(mlp_cols(a, in_width));
    const long u32 = (a->rows + 31) / 32;
    p->mode = 0;
    int mode_b = 0;
    if (int rc = products_mode(who, &p->mode, a->layers, a->n_layers, a->layers2, a->n_layers2)) return rc;
    if (b) {
        if (int rc = products_mode(who, &mode_b, b->layers, b->n_layers, b->layers2, b->n_layers2)) return rc;
        if (p->mode != mode_b) return fail(ELO_ERR_ARG, "%s: the two jobs of a paired launch must share one products mode", who);
    }
    p->t16 = small_tile(u32 * (b ? 2 : 1), 1);
    p->chain = false;
    p->C = a->n_sources == 2 ? a->src_width[1] : 0;
#ifndef ELO_DENSE_F32
    {   // the register-resident form (mlp2_rr_kernel) for the model's two-stage shape, from ELO_MLP_RR_ROWS rows per launch on
        // two regimes (fused._prepass_rows): batch >= 4 keeps the GPU full -- 24.0 k -> 25.5 k pairs/s at batch 8 with the chain at
        // every level; at batch 1 (7200 rows at l0) the tile kernel is faster: 10.2 k vs 9.8 k
        const long min_rows = mlp_rr_rows(a->batch_hint);
        const int C = p->C;
        const bool aligned = ((uintptr_t)a->src[0] | (uintptr_t)a->src[1] | (uintptr_t)a->before | (uintptr_t)a->after | (uintptr_t)a->out |
                              (uintptr_t)a->out2) % 16 == 0 &&
                             (!b || ((uintptr_t)b->src[0] | (uintptr_t)b->src[1] | (uintptr_t)b->before | (uintptr_t)b->after | (uintptr_t)b->out |
                                     (uintptr_t)b->out2) % 16 == 0);
        p->chain = (p->mode == MODE_SPLIT || p->mode == MODE_HALF) && cv1_rr_on() && a->n_sources == 2 && a->src_width[0] == 64 &&
                   (C == 16 || C == 32 || C == 64) && a->n_layers == 2 && a->layers[0].N == 128 && a->layers[1].N == 64 && a->n_layers2 == 2 &&
                   a->layers2[0].N == 128 && a->layers2[1].N == 64 && a->w_before == C && a->w_after == 64 && aligned &&
                   a->rows * (b ? 2 : 1) >= min_rows;
    }
#endif
    // the softmax_valid ride (elo_mlp_args.sv_*): tile kernel, 64-wide final output, whole tiles per batch element
    p->sv_tiles = 0;
    const int final_n = a->n_layers2 > 0 ? a->layers2[a->n_layers2 - 1].N : a->layers[a->n_layers - 1].N;
    if (!p->chain && final_n == 64 && a->sv_npoints > 0 && a->rows > 0 && a->rows % a->sv_npoints == 0) {
        // the ride launches (batch x row tiles per element) workgroups -- a pair is ONE 512-thread workgroup, and an element's last tile
        // may be ragged -- so the 16- / 32-row choice is made on THAT grid, not on the plain launch's
        const bool t16 = small_tile((a->rows / a->sv_npoints) * (((long)a->sv_npoints + 31) / 32), 1);
        const int tile = t16 ? 16 : 32, groups = b ? 2 : 1;
        const long tiles = ((long)a->sv_npoints + tile - 1) / tile;
        const size_t lds = sizeof(float) * ((size_t)groups * tile * p->S + (b ? 0 : (size_t)tile * 64));   // (mlp_sv_kernel's dynamic LDS)
        if (tiles <= ELO_SV_MAX_PARTS && lds <= 64 * 1024) {      // a wider generic MLP falls back on the separate partial-sums launch
            p->sv_tiles = (int)tiles;
            p->t16 = t16;
        }
    }
    return ELO_OK;
}

extern "C" int elo_mlp_sv_parts(const elo_mlp_args *a, const elo_mlp_args *b)
{
    MlpPlan p;
    if (plan_mlp(a, b, &p, "elo_mlp_sv_parts")) return 0;
    return p.sv_tiles;
}

extern "C" int elo_mlp_fused2(const elo_mlp_args *a, const elo_mlp_args *b, elo_stream_t stream)
{
    const char *who = "elo_mlp_fused";
    MlpPlan plan;
    if (int rc = plan_mlp(a, b, &plan, who)) return rc;
    if (a->rows == 0)
        return a->clear_scratch || a->sv_scratch ? fail(ELO_ERR_ARG, "%s: a clear_* / sv_* side job needs rows to ride on", who) : ELO_OK;
    const int S = plan.S, mode = plan.mode;
    const bool t16 = plan.t16;
    const long u32 = (a->rows + 31) / 32, u16 = (a->rows + 15) / 16;
    JobPair<elo_mlp_args> pair;
    pair.job[0] = *a;
    pair.job[1] = b ? *b : *a;
    if (a->sv_scratch) {               // the launch also reduces its final rows to softmax_valid's partial sums (mlp_sv_kernel)
        if (!plan.sv_tiles)
            return fail(ELO_ERR_ARG, "%s: sv_scratch is set but this launch cannot compute softmax_valid's partial sums "
                                     "(elo_mlp_sv_parts returns 0: chain-kernel regime, final width != 64, rows %% sv_npoints != 0 or too many tiles)", who);
        if (!a->sv_xyz) return fail(ELO_ERR_ARG, "%s: sv_scratch without sv_xyz", who);
        if ((b != nullptr) == (a->sv_feature != nullptr))
            return fail(ELO_ERR_ARG, "%s: a paired launch takes its features from job b (sv_feature NULL), a single launch from sv_feature", who);
        if (!b && ((uintptr_t)a->sv_feature & 15)) return fail(ELO_ERR_ARG, "%s: unaligned sv_feature", who);
        const int tile = t16 ? 16 : 32, groups = b ? 2 : 1;
        const size_t lds = sizeof(float) * ((size_t)groups * tile * S + (b ? 0 : (size_t)tile * 64));
        const dim3 grid((unsigned)(a->rows / a->sv_npoints) * (unsigned)plan.sv_tiles);
        hipStream_t s = (hipStream_t)stream;
#define CALL_SV(T_, M_)                                                                                                         \
        do {                                                                                                                    \
            if (b) hipLaunchKernelGGL((mlp_sv_kernel<T_, M_, true>), grid, dim3(2 * FUSED_BLOCK), lds, s, pair, S, plan.sv_tiles); \
            else hipLaunchKernelGGL((mlp_sv_kernel<T_, M_, false>), grid, dim3(FUSED_BLOCK), lds, s, pair, S, plan.sv_tiles);     \
        } while (0)
#define CALL_SV_T(M_) do { if (t16) CALL_SV(16, M_); else CALL_SV(32, M_); } while (0)
        ELO_PICK_MODE(mode, CALL_SV_T);
#undef CALL_SV_T
#undef CALL_SV
        ++g_sv_ride_launches;
        return check_launch(who);
    }
    const size_t lds = tile_lds_bytes(t16 ? 16 : 32, S);
#ifndef ELO_DENSE_F32
    if (plan.chain) {
        const int C = plan.C;
        const dim3 rgrid((unsigned)((a->rows + RR_ROWS - 1) / RR_ROWS), b ? 2u : 1u);
        const bool f16 = a->feat_dtype == ELO_F16;
        hipStream_t rs = (hipStream_t)stream;
#define RRM(C_)                                                                                                                      \
        do {                                                                                                                    \
            if (f16 && mode == MODE_HALF) hipLaunchKernelGGL((mlp2_rr_kernel<C_, true, MODE_HALF>), rgrid, dim3(RR_WAVES * 64), RR_LDS_BYTES, rs, pair);             \
            else if (f16) hipLaunchKernelGGL((mlp2_rr_kernel<C_, true, MODE_SPLIT>), rgrid, dim3(RR_WAVES * 64), RR_LDS_BYTES, rs, pair);             \
            else if (mode == MODE_HALF) hipLaunchKernelGGL((mlp2_rr_kernel<C_, false, MODE_HALF>), rgrid, dim3(RR_WAVES * 64), RR_LDS_BYTES, rs, pair);                \
            else hipLaunchKernelGGL((mlp2_rr_kernel<C_, false, MODE_SPLIT>), rgrid, dim3(RR_WAVES * 64), RR_LDS_BYTES, rs, pair);                \
        } while (0)
        if (C == 16) RRM(16);
        else if (C == 32) RRM(32);
        else RRM(64);
#undef RRM
        ++g_rr_launches[3];
        return check_launch(who);
    }
#endif
    const dim3 grid((unsigned)(t16 ? u16 : u32), b ? 2u : 1u);
    hipStream_t s = (hipStream_t)stream;
#define CALL(k) hipLaunchKernelGGL(k, grid, dim3(FUSED_BLOCK), lds, s, pair, S)
    ELO_PICK(mlp_kernel, t16, mode, CALL);
#undef CALL
    return check_launch(who);
}

extern "C" int elo_mlp_fused(const elo_mlp_args *a, elo_stream_t stream) { return elo_mlp_fused2(a, nullptr, stream); }

// vector gathers only: channel counts must be whole 16-byte items, tensors 16-byte aligned
static int check_cv_features(const char *who, int C, int f16, std::initializer_list<const void *> tensors)
{
    const int per = f16 ? 8 : 4;
    if (C % per != 0) return fail(ELO_ERR_LIMIT, "%s: C = %d must be a multiple of %d (%s feature storage)", who, C, per, f16 ? "fp16" : "fp32");
    if (32 * (C / per) > SEG_ITEMS * FUSED_BLOCK) return fail(ELO_ERR_LIMIT, "%s: C = %d too wide for the tile gather", who, C);
    for (const void *p : tensors)
        if ((uintptr_t)p & 15) return fail(ELO_ERR_ARG, "%s: feature tensors must be 16-byte aligned", who);
    return ELO_OK;
}

static int plan_cv1(const elo_cv1_args *a, TilePlan *p, const char *who)
{
    ELO_REQUIRE(a, who, "null argument block");
    ELO_REQUIRE(a->batch >= 0 && a->npoints > 0 && a->K > 0 && a->H2 > 0 && a->W2 > 0 && a->C > 0, who, "bad sizes");
    if (a->K > 32) return fail(ELO_ERR_LIMIT, "%s: K = %d exceeds the 32-row tile", who, a->K);
    ELO_REQUIRE(a->xyz1 && a->feat1 && a->xyz2 && a->feat2 && a->out, who, "null tensor pointer");
    ELO_REQUIRE(a->group.random_hw || (a->idx && a->mask), who, "neither idx/mask nor a grouping spec");
    ELO_REQUIRE(!a->group.random_hw || a->npoints == a->H2 * a->W2, who, "in-kernel grouping needs npoints == H2*W2");
    if (int rc = check_dtype(a->feat_dtype, who)) return rc;
    if (int rc = check_cv_features(who, a->C, a->feat_dtype == ELO_F16, {a->feat1, a->feat2})) return rc;
    const int CT = 10 + 2 * a->C;
    if (int rc = check_dense(a->cv0, CT, 128, who, "CV_0")) return rc;
    if (int rc = check_dense(a->cv1, 128, 64, who, "CV_1")) return rc;
    if (int rc = check_dense(a->cv2, 64, 64, who, "CV_2")) return rc;
    if (int rc = check_dense(a->cv_xyz, 10, 64, who, "CV_xyz")) return rc;
    if (int rc = check_dense(a->sum_cv0, 128, 128, who, "sum_CV_0")) return rc;
    if (int rc = check_dense(a->sum_cv1, 128, 64, who, "sum_CV_1")) return rc;
    const long points = (long)a->batch * a->npoints;
    if (points >= 0x7fffffffL) return fail(ELO_ERR_LIMIT, "%s: batch * npoints beyond 2^31", who);
    const int cols = 128 + cv1_feat_cols(a->C);
    p->S = row_stride(cols > 192 ? cols : 192);
    const int P32 = 32 / a->K, P16 = a->K <= 16 ? 16 / a->K : 1;
    const long u32 = (points + P32 - 1) / P32, u16 = (points + P16 - 1) / P16;
    if (int rc = products_mode(who, &p->mode, a->cv0, a->cv1, a->cv2, a->cv_xyz, a->sum_cv0, a->sum_cv1)) return rc;
    p->t16 = small_tile(u32, a->K);
    p->units = p->t16 ? u16 : u32;
    const int KT = a->group.random_hw ? a->group.kernel_h * a->group.kernel_w : 0;
    p->lds = tile_lds_bytes(p->t16 ? 16 : 32, p->S);                 // (order + select-k scratch alias the tile: cv1_tile)
    if (sizeof(float) * (((size_t)KT + 3) / 4 * 4 + (size_t)FUSED_WAVES * select_scratch_words(KT, a->K)) > sizeof(float) * (p->t16 ? 16 : 32) * p->S)
        return fail(ELO_ERR_LIMIT, "%s: window %dx%d does not fit the tile for in-kernel grouping", who, a->group.kernel_h, a->group.kernel_w);
    return check_group(a->group, a->H2, a->W2, p->lds, who);
}

extern "C" int elo_cv_stage1_fused(const elo_cv1_args *a, elo_stream_t stream)
{
    const char *who = "elo_cv_stage1_fused";
    TilePlan plan;
    if (int rc = plan_cv1(a, &plan, who)) return rc;
    if ((long)a->batch * a->npoints == 0) return ELO_OK;
#ifndef ELO_DENSE_F32
    if (!a->group.random_hw && cv_chain(a->C, plan.mode)) {
        const int P = RR_ROWS / a->K;                              // points per workgroup (128 rows)
        const dim3 grid((unsigned)(((long)a->batch * a->npoints + P - 1) / P));
        const size_t lds = RR_LDS_BYTES;
        hipStream_t s = (hipStream_t)stream;
        const bool f16 = a->feat_dtype == ELO_F16;
        static_assert(RR_LDS_BYTES <= 64 * 1024, "dynamic LDS within the default limit");
#define RR(CC)                                                                                                      \
        do {                                                                                                        \
            if (f16 && plan.mode == MODE_HALF) hipLaunchKernelGGL((cv1_rr_kernel<CC, true, MODE_HALF>), grid, dim3(RR_WAVES * 64), lds, s, *a);          \
            else if (f16) hipLaunchKernelGGL((cv1_rr_kernel<CC, true, MODE_SPLIT>), grid, dim3(RR_WAVES * 64), lds, s, *a);          \
            else if (plan.mode == MODE_HALF) hipLaunchKernelGGL((cv1_rr_kernel<CC, false, MODE_HALF>), grid, dim3(RR_WAVES * 64), lds, s, *a);             \
            else hipLaunchKernelGGL((cv1_rr_kernel<CC, false, MODE_SPLIT>), grid, dim3(RR_WAVES * 64), lds, s, *a);             \
        } while (0)
        if (a->C == 16) RR(16);
        else if (a->C == 32) RR(32);
        else RR(64);
#undef RR
        ++g_rr_launches[0];
        return check_launch(who);
    }
#endif
    const dim3 grid((unsigned)plan.units);
    const size_t lds = plan.lds;
    const int S = plan.S;
    hipStream_t s = (hipStream_t)stream;
#define CALL(k) hipLaunchKernelGGL(k, grid, dim3(FUSED_BLOCK), lds, s, *a, S)
    if (a->group.random_hw) ELO_PICK(cv1_kernel, plan.t16, plan.mode, CALL);
    else ELO_PICK(cv1_meta_kernel, plan.t16, plan.mode, CALL);
#undef CALL
    return check_launch(who);
}


extern "C" int elo_cv_stage1_setconv_fused(const elo_cv1_args *a, const elo_setconv_args *ja, const elo_setconv_args *jb,
                                           elo_stream_t stream)
{
    const char *who = "elo_cv_stage1_setconv_fused";
    TilePlan pc, ps;
    if (int rc = plan_cv1(a, &pc, who)) return rc;
    if (int rc = check_setconv(ja, who)) return rc;
    if (jb) {
        if (int rc = check_setconv(jb, who)) return rc;
        if (!same_shape(ja, jb)) return fail(ELO_ERR_ARG, "%s: the two set-conv jobs must have the same shape", who);
    }
    if (int rc = plan_setconv(ja, jb, &ps, who)) return rc;
    if (ps.mode != pc.mode) return fail(ELO_ERR_ARG, "%s: the cost volume and the set-conv jobs must share one products mode", who);
    if ((long)a->batch * a->npoints == 0 || (long)ja->batch * ja->npoints == 0)
        return fail(ELO_ERR_ARG, "%s: empty batch (call the separate entry points)", who);
    SideJobs side;
    side.job[0] = *ja;
    side.job[1] = jb ? *jb : *ja;
    side.njobs = jb ? 2 : 1;
    side.S = ps.S;
    side.blocks_per_job = (int)ps.units;
    const unsigned n_cv = (unsigned)pc.units;
    const dim3 grid(n_cv + (unsigned)(ps.units * side.njobs));
    const size_t lds = pc.lds > ps.lds ? pc.lds : ps.lds;
    const int S = pc.S;
    hipStream_t s = (hipStream_t)stream;
#define CALL_T(M) do {                                                                                                  \
        if (pc.t16 && ps.t16) hipLaunchKernelGGL((cv1_setconv_kernel<16, 16, M>), grid, dim3(FUSED_BLOCK), lds, s, *a, S, n_cv, side);      \
        else if (pc.t16) hipLaunchKernelGGL((cv1_setconv_kernel<16, 32, M>), grid, dim3(FUSED_BLOCK), lds, s, *a, S, n_cv, side);           \
        else if (ps.t16) hipLaunchKernelGGL((cv1_setconv_kernel<32, 16, M>), grid, dim3(FUSED_BLOCK), lds, s, *a, S, n_cv, side);           \
        else hipLaunchKernelGGL((cv1_setconv_kernel<32, 32, M>), grid, dim3(FUSED_BLOCK), lds, s, *a, S, n_cv, side);                        \
    } while (0)
    ELO_PICK_MODE(pc.mode, CALL_T);
#undef CALL_T
    return check_launch(who);
}

// The chain-kernel twin of elo_cv_stage1_setconv_fused (cv1_setconv_rr_kernel): the cost volume comes PRE-GROUPED (idx / mask
// of the select-k pre-pass), the two set-conv jobs are the level's set-upconv stage 1 (in-kernel random-k, 64 + 3 -> 128 -> 64).
// elo_cv_stage1_setconv_chain_form: 1 when (C, the layers' products mode, the jobs' shape and size) take it -- the host asks
// before it runs the pre-pass.
static int chain_pair_form(const elo_cv1_args *a, const elo_setconv_args *ja, const elo_setconv_args *jb, int *mode_out, const char *who)
{
#ifdef ELO_DENSE_F32
    (void)a; (void)ja; (void)jb; (void)mode_out; (void)who;
    return 0;
#else
    if (!a || !ja || !jb) return 0;
    int mode = 0, mode_sc = 0, mode_sb = 0;
    if (products_mode(who, &mode, a->cv0, a->cv1, a->cv2, a->cv_xyz, a->sum_cv0, a->sum_cv1)) return 0;
    if (ja->n_layers != 2 || jb->n_layers != 2 || products_mode(who, &mode_sc, ja->layers, 2) || products_mode(who, &mode_sb, jb->layers, 2)) return 0;
    if (mode != mode_sc || mode != mode_sb || !same_shape(ja, jb) || ja->feat_dtype != a->feat_dtype) return 0;
    if (!cv_chain(a->C, mode) || setconv_chain_shape(ja, jb, mode) != 1) return 0;
    if (a->K <= 0 || a->K > 32) return 0;
    *mode_out = mode;
    return 1;
#endif
}

extern "C" int elo_cv_stage1_setconv_chain_form(const elo_cv1_args *a, const elo_setconv_args *ja, const elo_setconv_args *jb)
{
    int mode = 0;
    return chain_pair_form(a, ja, jb, &mode, "elo_cv_stage1_setconv_chain_form");
}

extern "C" int elo_cv_stage1_setconv_chain(const elo_cv1_args *a, const elo_setconv_args *ja, const elo_setconv_args *jb,
                                           elo_stream_t stream)
{
    const char *who = "elo_cv_stage1_setconv_chain";
#ifdef ELO_DENSE_F32
    (void)a; (void)ja; (void)jb; (void)stream;
    return fail(ELO_ERR_LIMIT, "%s: the fp32-MFMA comparison build has no register-resident kernels", who);
#else
    TilePlan pc, ps;
    if (int rc = plan_cv1(a, &pc, who)) return rc;
    ELO_REQUIRE(jb, who, "two set-conv jobs");
    if (int rc = check_setconv(ja, who)) return rc;
    if (int rc = check_setconv(jb, who)) return rc;
    if (!same_shape(ja, jb)) return fail(ELO_ERR_ARG, "%s: the two set-conv jobs must have the same shape", who);
    if (int rc = plan_setconv(ja, jb, &ps, who)) return rc;
    if (a->group.random_hw || !a->idx || !a->mask) return fail(ELO_ERR_ARG, "%s: the cost volume comes pre-grouped (idx / mask of the select-k pre-pass)", who);
    int mode = 0;
    if (!chain_pair_form(a, ja, jb, &mode, who) || mode != pc.mode || mode != ps.mode)
        return fail(ELO_ERR_ARG, "%s: not a chain-form launch (elo_cv_stage1_setconv_chain_form returns 0)", who);
    const long pts_cv = (long)a->batch * a->npoints, pts_sc = (long)ja->batch * ja->npoints;
    if (pts_cv == 0 || pts_sc == 0) return fail(ELO_ERR_ARG, "%s: empty batch (call the separate entry points)", who);
    const int Pc = RR_ROWS / a->K, Ps = RR_ROWS / ja->K;
    const unsigned n_cv = (unsigned)((pts_cv + Pc - 1) / Pc), n_sc = (unsigned)((pts_sc + Ps - 1) / Ps);
    const dim3 grid(n_cv + 2u * n_sc);
    JobPair<elo_setconv_args> pair;
    pair.job[0] = *ja;
    pair.job[1] = *jb;
    hipStream_t s = (hipStream_t)stream;
    const bool f16 = a->feat_dtype == ELO_F16;
#define RRP(CC)                                                                                                                              \
    do {                                                                                                                                    \
        if (f16 && mode == MODE_HALF) hipLaunchKernelGGL((cv1_setconv_rr_kernel<CC, true, MODE_HALF>), grid, dim3(RR_WAVES * 64), RR_LDS_BYTES, s, *a, pair, n_cv, n_sc);   \
        else if (f16) hipLaunchKernelGGL((cv1_setconv_rr_kernel<CC, true, MODE_SPLIT>), grid, dim3(RR_WAVES * 64), RR_LDS_BYTES, s, *a, pair, n_cv, n_sc);                 \
        else if (mode == MODE_HALF) hipLaunchKernelGGL((cv1_setconv_rr_kernel<CC, false, MODE_HALF>), grid, dim3(RR_WAVES * 64), RR_LDS_BYTES, s, *a, pair, n_cv, n_sc);   \
        else hipLaunchKernelGGL((cv1_setconv_rr_kernel<CC, false, MODE_SPLIT>), grid, dim3(RR_WAVES * 64), RR_LDS_BYTES, s, *a, pair, n_cv, n_sc);                          \
    } while (0)
    if (a->C == 16) RRP(16);
    else if (a->C == 32) RRP(32);
    else RRP(64);
#undef RRP
    ++g_rr_launches[0];
    ++g_rr_launches[2];
    ++g_chain_pair_launches;
    return check_launch(who);
#endif
}

extern "C" int elo_cv_stage2_fused(const elo_cv2_args *a, elo_stream_t stream)
{
    const char *who = "elo_cv_stage2_fused";
    ELO_REQUIRE(a, who, "null argument block");
    ELO_REQUIRE(a->batch >= 0 && a->npoints > 0 && a->K > 0 && a->H > 0 && a->W > 0 && a->C > 0, who, "bad sizes");
    ELO_REQUIRE(a->npoints == a->H * a->W, who, "npoints must equal H*W (every pixel is a centre)");
    if (a->K > 32) return fail(ELO_ERR_LIMIT, "%s: K = %d exceeds the 32-row tile", who, a->K);
    if (a->C > 64) return fail(ELO_ERR_LIMIT, "%s: C = %d exceeds 64", who, a->C);
    ELO_REQUIRE(a->xyz1 && a->feat1 && a->cost && a->out, who, "null tensor pointer");
    ELO_REQUIRE(a->group.random_hw || (a->idx && a->mask), who, "neither idx/mask nor a grouping spec");
    if (int rc = check_dtype(a->feat_dtype, who)) return rc;
    if (int rc = check_cv_features(who, a->C, a->feat_dtype == ELO_F16, {a->feat1, a->cost})) return rc;
    if (int rc = check_dense(a->xyz_enc, 10, 64, who, "sum_xyz_encoding")) return rc;
    if (int rc = check_dense(a->sum_cost0, 128 + a->C, 128, who, "sum_cost_volume_0")) return rc;
    if (int rc = check_dense(a->sum_cost1, 128, 64, who, "sum_cost_volume_1")) return rc;
    const long points = (long)a->batch * a->npoints;
    if (points >= 0x7fffffffL) return fail(ELO_ERR_LIMIT, "%s: batch * npoints beyond 2^31", who);
    if (points == 0) return ELO_OK;
    const int S = row_stride(208);
    const int P32 = 32 / a->K, P16 = a->K <= 16 ? 16 / a->K : 1;
    const long u32 = (points + P32 - 1) / P32, u16 = (points + P16 - 1) / P16;
    int mode = 0;
    if (int rc = products_mode(who, &mode, a->xyz_enc, a->sum_cost0, a->sum_cost1)) return rc;
    const bool t16 = small_tile(u32, a->K);
    const int KT = a->group.random_hw ? a->group.kernel_h * a->group.kernel_w : 0;
    const size_t lds = tile_lds_bytes(t16 ? 16 : 32, S, KT, false);
    if (int rc = check_group(a->group, a->H, a->W, lds, who)) return rc;
    hipStream_t s = (hipStream_t)stream;
#ifndef ELO_DENSE_F32
    if (!a->group.random_hw && cv_chain(a->C, mode)) {
        const int P = RR_ROWS / a->K;                              // points per workgroup (128 rows)
        const dim3 rgrid((unsigned)((points + P - 1) / P));
        const bool f16 = a->feat_dtype == ELO_F16;
#define RR(CC)                                                                                                           \
        do {                                                                                                             \
            if (f16 && mode == MODE_HALF) hipLaunchKernelGGL((cv2_rr_kernel<CC, true, MODE_HALF>), rgrid, dim3(RR_WAVES * 64), RR_LDS_BYTES, s, *a);     \
            else if (f16) hipLaunchKernelGGL((cv2_rr_kernel<CC, true, MODE_SPLIT>), rgrid, dim3(RR_WAVES * 64), RR_LDS_BYTES, s, *a);     \
            else if (mode == MODE_HALF) hipLaunchKernelGGL((cv2_rr_kernel<CC, false, MODE_HALF>), rgrid, dim3(RR_WAVES * 64), RR_LDS_BYTES, s, *a);        \
            else hipLaunchKernelGGL((cv2_rr_kernel<CC, false, MODE_SPLIT>), rgrid, dim3(RR_WAVES * 64), RR_LDS_BYTES, s, *a);        \
        } while (0)
        if (a->C == 16) RR(16);
        else if (a->C == 32) RR(32);
        else RR(64);
#undef RR
        ++g_rr_launches[1];
        return check_launch(who);
    }
#endif
    const dim3 grid((unsigned)(t16 ? u16 : u32));
#define CALL(k) hipLaunchKernelGGL(k, grid, dim3(FUSED_BLOCK), lds, s, *a, S)
    ELO_PICK(cv2_kernel, t16, mode, CALL);
#undef CALL
    return check_launch(who);
}
