// elo_common.h -- shared device/host helpers for libelo_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cstdint>

#include "../../include/elo.h"

#define ELO_WAVE 64            // CDNA4 wavefront (hard-coded on purpose)
#define ELO_BLOCK 256          // 4 waves, one per SIMD
#define ELO_MAX_WINDOW 5000    // tf_ops/2d_conv_*_k/fused_conv_g.cu:42-43

namespace elo {

// ---- host: thread-local error text -------------------------------------
char *err_buf();
int fail(int code, const char *fmt, ...);
int check_launch(const char *what);
elo_tuning &tuning();           // the process-wide tuning (elo_set_tuning); the library reads no environment variable
elo_tuning &tuning_base();      // ... without the elo_debug_* overrides

// ---- device: arithmetic contract -----------------------------------------
// Squared norms are ((x*x + y*y) + z*z) in fp32 with no fused multiply-add, so
// that the GPU result equals the CPU oracle bit for bit (DESIGN.md "numerics").
__device__ __forceinline__ float sq3(float x, float y, float z)
{
    return __fadd_rn(__fadd_rn(__fmul_rn(x, x), __fmul_rn(y, y)), __fmul_rn(z, z));
}
// ---- integer divisions of the per-point prologues.  A 64-bit `pt / npoints` is an ~80-instruction routine (+ 75 scalar
// ones), `h / stride` a 24-instruction one; tools/isa_by_line.py showed them as 130 of the ~720 vector instructions a
// setconv_small wave executes, 126 of the 294 of a select-k wave.  The fused launchers bound batch * npoints by 2^31: there the
// point split is a 32-bit unsigned division (split_point); elsewhere the 32-bit form sits on a branch (point_batch).  Strides
// are 1, 2, 4 or 8 in this model: a shift behind a uniform branch (a division can trap in the abstract machine, so the
// compiler keeps the branch instead of computing both and selecting).
__device__ __forceinline__ void split_point(long pt, int npoints, int &b, int &n)
{
    const unsigned p = (unsigned)pt, q = p / (unsigned)npoints;
    b = (int)q; n = (int)(p - q * (unsigned)npoints);
}
__device__ __forceinline__ int div_stride(int v, int stride)
{
    if ((stride & (stride - 1)) == 0) return v >> (31 - __builtin_clz(stride));
    return v / stride;
}
__device__ __forceinline__ int point_batch(long pt, int npoints)      // pt / npoints for any pt
{
    if ((unsigned long)pt >> 31 == 0) return (int)((unsigned)pt / (unsigned)npoints);
    return (int)(pt / npoints);
}

// n / d for TILE-LOCAL indices (0 <= n < 2^20, 0 < d: rows of a tile, quads of a row, slots of a point): a float
// reciprocal, 4 instructions, instead of the ~35-instruction integer routine the compiler expands a 32-bit division into
// (tools/isa_by_line.py: 860 of mlp_kernel's 2757 vector instructions were seg_split's divisions, on the batch-1 critical
// path).  Exact: (n + 0.5) / d is at least 0.5 / d away from an integer and the float error is below (n / d) * 2^-22.
__device__ __forceinline__ int small_div(int n, int d) { return (int)(((float)n + 0.5f) * __builtin_amdgcn_rcpf((float)d)); }

// the reference's max(a,b) with a NaN first operand returns b
__device__ __forceinline__ float pick_max(float a, float b) { return a > b ? a : b; }

// e^x to ~2 ulp in 7 instructions on the hardware exp2: x * log2(e) in two pieces (the rounding of the product, up to
// |x| * 2^-24 in the exponent -- 1e-6 relative at x = -20 --, is carried as a first-order correction).  The argument is
// clamped at -104 (e^-104 = 2^-150 is 0 in fp32), which also makes exp_acc(-inf) = 0 without a NaN from the correction.
// Used where a softmax runs over thousands of points (softmax_valid): the plain v_exp_f32(x * 1.4427) form moved a
// refinement level's pose by enough (1e-7) to flip projection cells on one seed in ~10 ("What's weak" 2).
__device__ __forceinline__ float exp_acc(float x)
{
    const float L2E_HI = 1.44269502162933349609375f, L2E_LO = 1.925963033500011e-8f;
    x = fmaxf(x, -104.0f);
    const float t = __fmul_rn(x, L2E_HI);
    const float r = __fmaf_rn(x, L2E_LO, __fmaf_rn(x, L2E_HI, -t));
    return __builtin_amdgcn_exp2f(t) * __fmaf_rn(r, 0.693147180559945f, 1.0f);
}

// ---- device: the next pooled set of window visiting orders into a forward's order buffers (include/elo.h
// elo_perm_refresh_args): one workgroup; copies version (*cursor % versions), decodes it, advances the cursor
__device__ __forceinline__ void perm_refresh_block(const elo_perm_refresh_args &a)
{
    const unsigned cur = (unsigned)*a.cursor;           // the stored cursor is kept in [0, versions): no overflow however long a lane replays
    const int r = (int)(cur % (unsigned)a.versions);
    const int *src = a.pool + (size_t)r * a.total;
    for (int i = threadIdx.x; i < a.total; i += blockDim.x) {
        const int p = src[i];
        const int *e = a.table + 4 * a.entry_of[i];
        const int kH = e[2], kW = e[3];
        a.flat[i] = p;
        a.decoded[i] = ((p / kW - kH / 2) << 16) | ((p % kW - kW / 2) & 0xffff);
    }
    __syncthreads();
    if (threadIdx.x == 0) *a.cursor = (r + 1) % a.versions;
}

// ---- device: XCD-aware tile order ------------------------------------------
// The dispatcher places block b on XCD b % 8 (MI355X_MICROARCH.md, workgroup
// dispatch).  Give each XCD one contiguous run of tiles so that neighbouring
// centres, which read overlapping windows, share that XCD's L2.  Pure speed
// choice: any placement gives the same result.
__device__ __forceinline__ unsigned xcd_tile(unsigned bid, unsigned nblocks)
{
    const unsigned q = nblocks >> 3, r = nblocks & 7u;
    const unsigned xcd = bid & 7u, k = bid >> 3;
    return xcd * q + (xcd < r ? xcd : r) + k;
}

// ---- device: sub-wave ballots ----------------------------------------------
// G lanes (16/32/64) cooperate on one centre; a wave holds 64/G groups.
template <int G>
__device__ __forceinline__ unsigned long long group_ballot(bool p, int group_shift)
{
    unsigned long long m = __ballot(p);
    if (G == 64) return m;
    return (m >> group_shift) & ((1ull << (G & 63)) - 1ull);
}

// Wave-wide minimum of a u32 on the DPP data path (no LDS crossbar traffic, unlike __shfl_xor/ds_bpermute):
// quad swaps, row_shr:4/8, row_bcast:15/31 leave the result in lane 63; readlane broadcasts it.
template <int CTRL>
__device__ __forceinline__ unsigned dpp_min_step(unsigned v)
{
    const unsigned o = (unsigned)__builtin_amdgcn_update_dpp((int)0xffffffffu, (int)v, CTRL, 0xf, 0xf, false);
    return o < v ? o : v;
}

__device__ __forceinline__ unsigned wave_min_u32(unsigned v)
{
    v = dpp_min_step<0xb1>(v);     // quad_perm:[1,0,3,2]
    v = dpp_min_step<0x4e>(v);     // quad_perm:[2,3,0,1]
    v = dpp_min_step<0x114>(v);    // row_shr:4
    v = dpp_min_step<0x118>(v);    // row_shr:8
    v = dpp_min_step<0x142>(v);    // row_bcast:15
    v = dpp_min_step<0x143>(v);    // row_bcast:31
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

template <int CTRL>
__device__ __forceinline__ unsigned dpp_max_step(unsigned v)
{
    const unsigned o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, false);
    return o > v ? o : v;
}

template <int CTRL>
__device__ __forceinline__ unsigned dpp_add_step(unsigned v)
{
    return v + (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, false);
}

// minimum over each aligned group of 8 lanes, in every lane of the group (quad swaps + row_half_mirror)
__device__ __forceinline__ unsigned group8_min_u32(unsigned v)
{
    v = dpp_min_step<0xb1>(v);     // quad_perm:[1,0,3,2]
    v = dpp_min_step<0x4e>(v);     // quad_perm:[2,3,0,1]
    v = dpp_min_step<0x141>(v);    // row_half_mirror: the other quad of the 8
    return v;
}

// maximum over each aligned group of 8 lanes, in every lane of the group
__device__ __forceinline__ unsigned group8_max_u32(unsigned v)
{
    v = dpp_max_step<0xb1>(v);
    v = dpp_max_step<0x4e>(v);
    v = dpp_max_step<0x141>(v);
    return v;
}

// wave-wide maximum of a value that is uniform inside every group of 8 lanes
__device__ __forceinline__ unsigned wave_max_of_group8(unsigned v)
{
    v = dpp_max_step<0x140>(v);    // row_mirror: the other half of the row
    v = dpp_max_step<0x142>(v);    // row_bcast:15
    v = dpp_max_step<0x143>(v);    // row_bcast:31
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

// wave-wide sum (lane 63 holds it after the standard DPP reduction)
__device__ __forceinline__ unsigned wave_sum_u32(unsigned v)
{
    v = dpp_add_step<0xb1>(v);
    v = dpp_add_step<0x4e>(v);
    v = dpp_add_step<0x114>(v);    // row_shr:4
    v = dpp_add_step<0x118>(v);    // row_shr:8
    v = dpp_add_step<0x142>(v);    // row_bcast:15
    v = dpp_add_step<0x143>(v);    // row_bcast:31
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

// lexicographic (hi, lo) minimum over the wave, as one u64 key
__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long v)
{
    const unsigned hi = (unsigned)(v >> 32), lo = (unsigned)v;
    const unsigned mhi = wave_min_u32(hi);
    const unsigned mlo = wave_min_u32(hi == mhi ? lo : 0xffffffffu);
    return ((unsigned long long)mhi << 32) | mlo;
}

}  // namespace elo
