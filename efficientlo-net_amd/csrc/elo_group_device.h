// elo_group_device.h -- device-side pieces of the neighbour grouping shared by the stand-alone grouping
// kernels (elo_grouping.hip) and the fused kernels that group in-kernel (elo_fused.hip).
// Reference semantics: tf_ops/2d_conv_random_k/fused_conv_g.cu:13-156, tf_ops/2d_conv_select_k/fused_conv_g.cu:11-209.
#pragma once
#include "elo_common.h"

namespace elo {

// -DELO_CV1_CLOCK (debugging build, tools/cv1_clock.sh): s_memtime stamps of wave 0 of workgroup 0
#ifdef ELO_CV1_CLOCK
__device__ unsigned long long g_cv1_clock[24];
#define ELO_GROUP_STAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_cv1_clock[i] = __builtin_readcyclecounter(); } while (0)
#else
#define ELO_GROUP_STAMP(i) do { } while (0)
#endif

#define ELO_EPS 1e-10f
#define ELO_FAR 1e10f

struct Probe {
    bool valid;   // in-grid, non-empty pixel
    bool hit;     // valid and within the radius
    float d;      // clamped squared distance (hit or not)
    int hw;       // (h << 16) | w of the probed pixel
};

// The queried grid is read either through a plain pointer (stand-alone kernels) or through a buffer resource (fused
// kernels: the base is wave-uniform, so it lives in scalar registers and a probe's address is one 32-bit pixel*12
// instead of a 64-bit vector multiply-add per load -- three of those per probe in the ISA of the pointer form).
struct GridBuf { __amdgpu_buffer_rsrc_t rsrc; };
struct Xyz { float x, y, z; };

__device__ __forceinline__ GridBuf grid_buffer(const float *base)
{
    return GridBuf{__builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(base), 0, 0x7fffffff, 0x00020000)};
}

__device__ __forceinline__ Xyz pixel3(const float *__restrict__ grid, int pixel)
{
    const float *q = grid + (size_t)pixel * 3;
    return Xyz{q[0], q[1], q[2]};
}

__device__ __forceinline__ Xyz pixel3(const GridBuf &grid, int pixel)
{
    typedef float f32x3 __attribute__((ext_vector_type(3)));
    const f32x3 v = __builtin_bit_cast(f32x3, __builtin_amdgcn_raw_buffer_load_b96(grid.rsrc, pixel * 12, 0, 0));
    return Xyz{v.x, v.y, v.z};
}

// One window slot of one centre. fused_conv_g.cu:80-123.
template <class Grid>
__device__ __forceinline__ Probe probe_slot(const Grid &grid2, int H2, int W2, int off,
                                            int base_h, int base_w, float cx, float cy, float cz, float r2)
{
    Probe p{false, false, ELO_FAR, 0};
    int h = base_h + (off >> 16);
    int w = base_w + (int)(short)(off & 0xffff);
    if (h < 0 || h >= H2) return p;
    if (w < 0) w += W2;
    if (w >= W2) w -= W2;
    const Xyz q = pixel3(grid2, h * W2 + w);
    const float qx = q.x, qy = q.y, qz = q.z;
    if (sq3(qx, qy, qz) <= ELO_EPS) return p;
    p.valid = true;
    p.d = pick_max(sq3(__fsub_rn(cx, qx), __fsub_rn(cy, qy), __fsub_rn(cz, qz)), ELO_EPS);
    p.hit = !(p.d > r2);
    p.hw = (h << 16) | w;
    return p;
}

// Decode the visiting order once per block: LDS[i] = (dh << 16) | (dw & 0xffff).
__device__ __forceinline__ void stage_offsets(int *lds_off, const int *__restrict__ perm, int kH, int kW,
                                              const int *__restrict__ decoded = nullptr)
{
    const int KT = kH * kW, hh = kH / 2, hw = kW / 2;
    if (decoded) {                                   // the host decoded the order once (elo_group_spec.decoded_hw)
        for (int i = threadIdx.x; i < KT; i += blockDim.x) lds_off[i] = decoded[i];
        __syncthreads();
        return;
    }
    for (int i = threadIdx.x; i < KT; i += blockDim.x) {
        const int p = perm[i];
        const int dh = p / kW - hh, dw = p % kW - hw;
        lds_off[i] = (dh << 16) | (dw & 0xffff);
    }
    __syncthreads();
}


// ---- wave-per-centre forms used inside the fused kernels (flag_copy == 0) ----------------------------
// Both call emit(slot, hw) for slots 0..count-1 (hw = (h << 16) | w of the neighbour) and return count;
// slots count..K-1 are the reference's zero-filled slots (index (0,0,0), mask 0).

// first K in-range neighbours in visiting order
template <class Grid, class Emit>
__device__ __forceinline__ int wave_random_k(const Grid &grid2, int H2, int W2, int KT, int K,
                                             const int *lds_off, int base_h, int base_w, float cx, float cy, float cz,
                                             float r2, Emit emit)
{
    const int lane = threadIdx.x & 63;
    int taken = 0;
    for (int base = 0; base < KT; base += ELO_WAVE) {
        const int i = base + lane;
        Probe p{false, false, ELO_FAR, 0};
        if (i < KT) p = probe_slot(grid2, H2, W2, lds_off[i], base_h, base_w, cx, cy, cz, r2);
        const unsigned long long mh = __ballot(p.hit);
        if (mh == 0) continue;
        const int slot = taken + __popcll(mh & ((1ull << lane) - 1ull));
        if (p.hit && slot < K) emit(slot, p.hw);
        taken += __popcll(mh);
        if (taken >= K) return K;
    }
    return taken;
}

// Raw fetch of one window slot with an UNCONDITIONAL load (out-of-grid rows are clamped and flagged), so that
// several probes can have their loads in flight together; judge() then applies fused_conv_g.cu:83-123.
struct RawSlot { float x, y, z; int hw; bool in_grid; };

template <class Grid>
__device__ __forceinline__ RawSlot fetch_slot(const Grid &grid2, int H2, int W2, int off, int base_h,
                                              int base_w, bool active)
{
    int h = base_h + (off >> 16);
    int w = base_w + (int)(short)(off & 0xffff);
    const bool in_grid = active && h >= 0 && h < H2;
    h = h < 0 ? 0 : h >= H2 ? H2 - 1 : h;
    if (w < 0) w += W2;
    if (w >= W2) w -= W2;                              // in [0, W2): |dw| <= kernel_w / 2 <= W2 (launchers), and inactive lanes are
                                                        // handed a real slot's offset by every caller -- no clamp needed behind the wrap
    const Xyz q = pixel3(grid2, h * W2 + w);
    return RawSlot{q.x, q.y, q.z, (h << 16) | w, in_grid};
}

__device__ __forceinline__ Probe judge(const RawSlot &r, float cx, float cy, float cz, float r2)
{
    // straight-line: an early return here became a branch per probe (exec save / restore, 17 of them around 8 probes)
    const bool valid = r.in_grid && !(sq3(r.x, r.y, r.z) <= ELO_EPS);
    const float d = pick_max(sq3(__fsub_rn(cx, r.x), __fsub_rn(cy, r.y), __fsub_rn(cz, r.z)), ELO_EPS);
    return Probe{valid, valid && !(d > r2), valid ? d : ELO_FAR, valid ? r.hw : 0};
}

// The register form keeps the window in J = 2, 3 or 8 registers per lane (position j*64 + lane in register j), so a
// round costs J compares: K <= 8 with any window up to 512 slots (the refinement cost volumes: 11x41 -> J = 8,
// 7x25 -> 3, 5x15 -> 2), and K <= 32 when the window fits three registers (the l2 cost volume's 5x35, K = 32).
// Measured at 64x1800 (tools/grouping_micro.py) with J fixed at 8: K = 6 register form 158 vs ~190 us (11x41); K = 32
// register form 439 vs 379 us (5x35) -- eight compares per round for a three-register window; hence J by window size.
__host__ __device__ __forceinline__ bool select_in_registers(int KT, int K)
{
    return (KT <= 512 && K <= 8) || (KT <= 192 && K <= 32);
}

// wave-private LDS words select-k needs per wave: the two [KT] arrays of the LDS form, or the 2 x 64 candidate slots of
// the register form's small-K rank path
__host__ __device__ __forceinline__ int select_scratch_words(int KT, int K)
{
    return select_in_registers(KT, K) ? 128 : (2 * KT + 3) & ~3;            // (multiples of 4 words: 16-byte aligned per wave)
}

// ---- select-k without the K dependent rounds ------------------------------------------------------------------------
// The reference's selection sort (fused_conv_g.cu:148-204) outputs the K smallest distances in increasing order; WHICH of
// two EQUAL distances comes first depends on array positions as the swaps left them, and only then.  A round is a
// ~150-instruction dependent chain (measured: ~1070 cycles per round for the wave that runs it alone: tools/cv1_clock.sh),
// so K = 6 rounds were 47 % of a cost-volume tile at batch 1 and K = 32 rounds nearly all of the l2_origin one.  The two
// forms below compute every element's RANK (number of strictly smaller distances) instead -- slot s takes the element
// of rank s -- and verify that the ranks they hand out are 0 .. count-1, each exactly once: any exact tie that involves
// a selected element (or the first one left out) breaks that, and only then the swap rounds run (on the same registers).
// Real scans essentially never tie; the exact-tie lattices of the tests take the fallback and stay bit-exact.

// Shared tail of the two threshold forms.  Candidates = the in-range elements <= T (T' = min(T, FAR - 1): one compare);
// they are compacted one per lane into cand[0..63] (distance) / cand[64..127] (payload) -- slot = running total +
// v_mbcnt of the ballot, no bounds check: more than 64 candidates only scribble over slots of an attempt that is given up
// -- and each is ranked against the candidate list read back as broadcast 16-byte items (the list is padded with
// 0xffffffff, which is below nothing).  count = min(total, K): fewer than K candidates means T was "no bound" (FAR), i.e.
// every in-range element is a candidate.
template <int J, class Emit>
__device__ __forceinline__ bool rank_candidates(const unsigned (&d)[J], const int (&pw)[J], unsigned T, int K, unsigned *cand,
                                                Emit emit, int &count)
{
    const unsigned FAR_BITS = __float_as_uint(ELO_FAR);
    const int lane = threadIdx.x & 63;
    const unsigned Tp = T < FAR_BITS ? T : FAR_BITS - 1u;
    cand[lane] = 0xffffffffu;
    int total = 0;
#pragma unroll
    for (int j = 0; j < J; ++j) {
        const bool c = d[j] <= Tp;
        const unsigned long long m = __ballot(c);
        const int slot = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, (unsigned)total)) & 63;
        if (c) { cand[slot] = d[j]; cand[64 + slot] = (unsigned)pw[j]; }
        total += __popcll(m);
    }
    count = total < K ? total : K;
    if (total > 64) return false;                                  // (uniform) does not fit one element per lane
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const unsigned dc = cand[lane];                                 // 0xffffffff beyond `total`
    const int pc = (int)cand[64 + lane];
    const uint4 *list = reinterpret_cast<const uint4 *>(cand);      // (callers keep cand 16-byte aligned)
    int rank = 0;
    {
        uint4 x[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) x[i] = list[i];                 // the first 16 in one go: usually all there are
#pragma unroll
        for (int i = 0; i < 4; ++i) rank += (x[i].x < dc) + (x[i].y < dc) + (x[i].z < dc) + (x[i].w < dc);
    }
    for (int i = 4; i * 4 < total; ++i) {
        const uint4 x = list[i];
        rank += (x.x < dc) + (x.y < dc) + (x.z < dc) + (x.w < dc);
    }
    const bool sel = rank < count && dc != 0xffffffffu;
    if (__popcll(__ballot(sel)) != count) return false;
    if (wave_sum_u32(sel ? (unsigned)rank : 0u) != (unsigned)(count * (count - 1) / 2)) return false;   // 0 .. count-1, each once
    if (sel) emit(rank, pc);
    return true;
}

// K <= 7: the maximum over 8 lane-groups of the group minimum bounds the 8th smallest distance from above, so the K + 1
// smallest are among the (typically 10-20) elements <= that bound.
template <int J, class Emit>
__device__ __forceinline__ bool select_small_k(const unsigned (&d)[J], const int (&pw)[J], int K, unsigned *cand, Emit emit,
                                               int &count)
{
    unsigned lmin = d[0];
#pragma unroll
    for (int j = 1; j < J; ++j) lmin = d[j] < lmin ? d[j] : lmin;
    const unsigned T = wave_max_of_group8(group8_min_u32(lmin));
    return rank_candidates<J>(d, pw, T, K, cand, emit, count);
}

// K < 64 on a window of more than 128 slots: the same idea with the threshold taken from the 64 PER-LANE minima -- their
// (K+1)-th smallest bounds the (K+1)-th smallest distance from above -- found by ranking the lane minima against each
// other (through the same LDS list), after which the candidates (typically K + 10..20; more than 64: give up) are
// compacted and ranked.  ~400 instructions where ranking all of 175 slots against each other takes ~1200.
template <int J, class Emit>
__device__ __forceinline__ bool select_mid_k(const unsigned (&d)[J], const int (&pw)[J], int K, unsigned *cand, Emit emit,
                                             int &count)
{
    const int lane = threadIdx.x & 63;
    unsigned lmin = d[0];
#pragma unroll
    for (int j = 1; j < J; ++j) lmin = d[j] < lmin ? d[j] : lmin;
    cand[lane] = lmin;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const uint4 *list = reinterpret_cast<const uint4 *>(cand);
    int below = 0;                                                  // lane minima strictly below this lane's
#pragma unroll
    for (int i0 = 0; i0 < 16; i0 += 8) {
        uint4 x[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i] = list[i0 + i];
#pragma unroll
        for (int i = 0; i < 8; ++i) below += (x[i].x < lmin) + (x[i].y < lmin) + (x[i].z < lmin) + (x[i].w < lmin);
    }
    const unsigned T = wave_max_of_group8(group8_max_u32(below <= K ? lmin : 0u));     // = the (K+1)-th smallest lane minimum
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    return rank_candidates<J>(d, pw, T, K, cand, emit, count);
}

// windows of at most 192 slots (J <= 3), any K: every element is ranked against the whole window.
template <int J, class Emit>
__device__ __forceinline__ bool select_by_rank(const unsigned (&d)[J], const int (&pw)[J], int KT, int K, Emit emit, int &count)
{
    const unsigned FAR_BITS = __float_as_uint(ELO_FAR);
    int rank[J];
#pragma unroll
    for (int j = 0; j < J; ++j) rank[j] = 0;
#pragma unroll
    for (int jj = 0; jj < J; ++jj) {
        const int n = KT - jj * 64 < 64 ? KT - jj * 64 : 64;        // (uniform) slots held in register jj
        for (int l = 0; l < n; ++l) {
            const unsigned di = (unsigned)__builtin_amdgcn_readlane((int)d[jj], l);
#pragma unroll
            for (int j = 0; j < J; ++j) rank[j] += di < d[j];
        }
    }
    int nvalid = 0, nsel = 0;
    unsigned part = 0;
#pragma unroll
    for (int j = 0; j < J; ++j) {
        const bool hit = d[j] < FAR_BITS, sel = hit && rank[j] < K;
        nvalid += __popcll(__ballot(hit));
        nsel += __popcll(__ballot(sel));
        part += sel ? (unsigned)rank[j] : 0u;
    }
    count = nvalid < K ? nvalid : K;
    if (nsel != count) return false;
    if (wave_sum_u32(part) != (unsigned)(count * (count - 1) / 2)) return false;    // ranks 0 .. count-1, each once
#pragma unroll
    for (int j = 0; j < J; ++j)
        if (d[j] < FAR_BITS && rank[j] < K) emit(rank[j], pw[j]);
    return true;
}

// The window of J*64 slots in registers; the rank forms above where they apply, else (or on an exact tie) K rounds of the
// reference's selection sort.  `cand`: 128 words of wave-private LDS (may be null: no small-K form).
template <int J, class Grid, class Emit>
__device__ __forceinline__ int select_rounds_in_registers(const Grid &grid2, int H2, int W2, int KT, int rounds,
                                                          const int *lds_off, int base_h, int base_w, float cx, float cy,
                                                          float cz, float r2, unsigned *cand, Emit emit, int &seen,
                                                          int &taken)
{
    const int lane = threadIdx.x & 63;
    const unsigned FAR_BITS = __float_as_uint(ELO_FAR);
    constexpr int BATCH = J < 4 ? J : 4;                   // probes in flight per lane (8: measured, no faster)
    unsigned d[J];
    int pw[J];
    ELO_GROUP_STAMP(12);
#pragma unroll
    for (int b0 = 0; b0 < J; b0 += BATCH) {
        RawSlot raw[BATCH];
#pragma unroll
        for (int u = 0; u < BATCH; ++u) {
            const int i = (b0 + u) * 64 + lane;
            raw[u] = fetch_slot(grid2, H2, W2, lds_off[i < KT ? i : 0], base_h, base_w, i < KT);
        }
#pragma unroll
        for (int u = 0; u < BATCH; ++u) {
            const Probe p = judge(raw[u], cx, cy, cz, r2);
            seen += __popcll(__ballot(p.valid));                         // (dropped by the compiler where the caller ignores them)
            taken += __popcll(__ballot(p.hit));
            d[b0 + u] = p.hit ? __float_as_uint(p.d) : FAR_BITS;        // d > 0: bit order == float order
            pw[b0 + u] = p.hit ? p.hw : 0;
        }
        if (J == 8 && KT <= 256 && b0 == 0) {               // the second batch would be all padding
#pragma unroll
            for (int u = BATCH; u < J; ++u) { d[u] = FAR_BITS; pw[u] = 0; }
            break;
        }
    }
    ELO_GROUP_STAMP(13);
    {
        int count = 0;
        bool done = false;
        if (rounds <= 7 && cand) done = select_small_k<J>(d, pw, rounds, cand, emit, count);
        else if (J <= 3 && rounds < 64 && cand) {
            done = select_mid_k<J>(d, pw, rounds, cand, emit, count);
            if (!done) done = select_by_rank<J>(d, pw, KT, rounds, emit, count);       // more than 64 candidates (or a tie)
        }
        else if (J <= 3) done = select_by_rank<J>(d, pw, KT, rounds, emit, count);
        if (done) { ELO_GROUP_STAMP(15); return count; }              // (uniform)
    }
    for (int s = 0; s < rounds; ++s) {
        if (s == 1) ELO_GROUP_STAMP(14);
        unsigned best = 0xffffffffu;
        int where = 0x7fffffff;
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const bool open = j > 0 || lane >= s;            // positions < s are already placed (s < 64)
            if (open && d[j] < best) { best = d[j]; where = j * 64 + lane; }
        }
        const unsigned long long key = wave_min_u64(((unsigned long long)best << 32) | (unsigned)where);
        if ((unsigned)(key >> 32) >= FAR_BITS) return s;                 // sorted: everything left is empty
        const int m = (int)(key & 0xffffffffu), mj = m >> 6, ml = m & 63;
        int sel_pw = pw[0];
#pragma unroll
        for (int j = 1; j < J; ++j) sel_pw = mj == j ? pw[j] : sel_pw;
        const int pm = __shfl(sel_pw, ml, ELO_WAVE);
        const unsigned ds = (unsigned)__shfl((int)d[0], s, ELO_WAVE);       // element at position s (lane s, register 0)
        const int ps = __shfl(pw[0], s, ELO_WAVE);
        if (m != s && lane == ml) {                                      // the reference's swap: s moves to m
#pragma unroll
            for (int j = 0; j < J; ++j)
                if (mj == j) { d[j] = ds; pw[j] = ps; }
        }
        if (lane == 0) emit(s, pm);
    }
    ELO_GROUP_STAMP(15);
    return rounds;
}

// wave_random_k whose first window step (slots 0..63) was fetched by the caller (fetch_slot: a plain load, so that the first
// steps of SEVERAL centres can be in flight together -- the walk of a centre is otherwise one L2 round trip per step, and
// most centres are done after the first)
template <class Grid, class Emit>
__device__ __forceinline__ int wave_random_k_prefetched(const Grid &grid2, int H2, int W2, int KT, int K, const int *lds_off, int base_h,
                                                        int base_w, float cx, float cy, float cz, float r2, const RawSlot &first, Emit emit)
{
    const int lane = threadIdx.x & 63;
    int taken = 0;
    {
        const Probe p = judge(first, cx, cy, cz, r2);
        const unsigned long long mh = __ballot(p.hit);
        const int slot = __popcll(mh & ((1ull << lane) - 1ull));
        if (p.hit && slot < K) emit(slot, p.hw);
        taken = __popcll(mh);
        if (taken >= K) return K;
    }
    for (int base = ELO_WAVE; base < KT; base += ELO_WAVE) {
        const int i = base + lane;
        Probe p{false, false, ELO_FAR, 0};
        if (i < KT) p = probe_slot(grid2, H2, W2, lds_off[i], base_h, base_w, cx, cy, cz, r2);
        const unsigned long long mh = __ballot(p.hit);
        if (mh == 0) continue;
        const int slot = taken + __popcll(mh & ((1ull << lane) - 1ull));
        if (p.hit && slot < K) emit(slot, p.hw);
        taken += __popcll(mh);
        if (taken >= K) return K;
    }
    return taken;
}

// K nearest in-range neighbours, reference tie order (selection sort with swaps): register form (above) where
// select_in_registers() says so, otherwise the LDS arrays dist/pay ([KT] wave-private each).  seen / taken: the numbers
// of existing and of in-range window slots (the reference's valid_idx / valid_in_dis_idx prefix lengths) are ADDED to them.
template <class Grid, class Emit>
__device__ __forceinline__ int wave_select_k(const Grid &grid2, int H2, int W2, int KT, int K,
                                             const int *lds_off, int base_h, int base_w, float cx, float cy, float cz,
                                             float r2, unsigned *dist, int *pay, Emit emit, int &seen, int &taken)
{
    const int lane = threadIdx.x & 63;
    const unsigned FAR_BITS = __float_as_uint(ELO_FAR);
    const int rounds = K < KT ? K : KT;
    if (select_in_registers(KT, K)) {
        unsigned *cand = dist;                          // wave-private scratch of the caller: >= 128 words (or null)
        if (KT <= 128)
            return select_rounds_in_registers<2>(grid2, H2, W2, KT, rounds, lds_off, base_h, base_w, cx, cy, cz, r2, cand, emit, seen, taken);
        if (KT <= 192)
            return select_rounds_in_registers<3>(grid2, H2, W2, KT, rounds, lds_off, base_h, base_w, cx, cy, cz, r2, cand, emit, seen, taken);
        return select_rounds_in_registers<8>(grid2, H2, W2, KT, rounds, lds_off, base_h, base_w, cx, cy, cz, r2, cand, emit, seen, taken);
    }
    for (int base = 0; base < KT; base += ELO_WAVE) {
        const int i = base + lane;
        if (i < KT) {
            const Probe p = probe_slot(grid2, H2, W2, lds_off[i], base_h, base_w, cx, cy, cz, r2);
            dist[i] = p.hit ? __float_as_uint(p.d) : FAR_BITS;
            pay[i] = p.hit ? p.hw : 0;
            seen += __popcll(__ballot(p.valid));
            taken += __popcll(__ballot(p.hit));
        }
    }
    for (int s = 0; s < rounds; ++s) {
        unsigned best = 0xffffffffu;
        int where = 0x7fffffff;
        for (int t = s + lane; t < KT; t += ELO_WAVE) {
            const unsigned d = dist[t];
            if (d < best) { best = d; where = t; }
        }
        const unsigned long long key = wave_min_u64(((unsigned long long)best << 32) | (unsigned)where);
        const int m = (int)(key & 0xffffffffu);
        if ((unsigned)(key >> 32) >= FAR_BITS) return s;          // sorted: everything left is empty
        const int pm = pay[m];
        if (m != s && lane == 0) { dist[m] = dist[s]; pay[m] = pay[s]; }
        if (lane == 0) emit(s, pm);
    }
    return rounds;
}

template <class Grid, class Emit>
__device__ __forceinline__ int wave_select_k(const Grid &grid2, int H2, int W2, int KT, int K,
                                             const int *lds_off, int base_h, int base_w, float cx, float cy, float cz,
                                             float r2, unsigned *dist, int *pay, Emit emit)
{
    int seen = 0, taken = 0;
    return wave_select_k(grid2, H2, W2, KT, K, lds_off, base_h, base_w, cx, cy, cz, r2, dist, pay, emit, seen, taken);
}

}  // namespace elo
