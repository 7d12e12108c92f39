// elo_grouping.hip -- projection-aware neighbour grouping on the range image
// for MI355X (gfx950, wave64).
//
// What it computes is defined by the reference kernels
//   tf_ops/2d_conv_random_k/fused_conv_g.cu:13-156   (first K hits in caller order)
//   tf_ops/2d_conv_select_k/fused_conv_g.cu:11-209   (K nearest, selection sort with swaps)
// How it computes it is not: the reference runs ONE thread per centre in
// `batch` blocks and walks the window serially out of global memory.  Here
//   * a group of G lanes (16/32, picked from the window size and K) owns a centre
//     and examines G*U window slots per step (U loads in flight per lane);
//   * the (dh,dw) of every slot in visiting order is decoded once per block
//     into LDS (no per-candidate integer division);
//   * output slots are assigned with __ballot + popcount prefix sums, so the
//     "first K in visiting order" rule needs no serial walk and stops as soon
//     as K hits exist;
//   * select-k keeps (distance, packed hw) of the whole window in LDS and runs
//     the reference's K selection rounds as wave-wide (distance, position)
//     arg-mins followed by the same swap, which reproduces its tie order;
//   * every output element is written exactly once (zero-fill fused; the op
//     glue's four cudaMemset calls, fused_conv.cpp:154-166, disappear).
// Both are memory-latency bound gathers out of an L2-resident grid: no MFMA.
#include "elo_group_device.h"
#include <cstdlib>

namespace elo {
namespace {

struct Centre {
    bool ok;
    int b, base_h, base_w;
    float x, y, z;
};

__device__ __forceinline__ Centre load_centre(const elo_group_args &a, long u)
{
    Centre c;
    c.b = point_batch(u, a.npoints);
    const int hc = a.idx_n2[u * 2 + 0], wc = a.idx_n2[u * 2 + 1];
    const float *p = a.xyz1 + (((size_t)c.b * a.H + hc) * a.W + wc) * 3;
    c.x = p[0]; c.y = p[1]; c.z = p[2];
    c.ok = !(pick_max(sq3(c.x, c.y, c.z), ELO_EPS) <= ELO_EPS);      // :62-70
    c.base_h = div_stride(hc, a.stride_h);
    c.base_w = div_stride(wc, a.stride_w);
    return c;
}

// ---------------------------------------------------------------- random-k
// G lanes own a centre and examine G*U window slots per step: every lane requests its U slots (unconditional,
// clamped loads) before the first one is judged, so a step is one L2 round trip whatever U is, and a wave carries
// 64/G centres.  The kernel is a chain of dependent round trips per centre (visiting order, centre index, centre,
// window), so centres in flight are what it is bound by: for windows above 32 slots G follows K (16 lanes up to
// K = 16, else 32) with U = 64/G instead of one wave per centre -- 60 -> 40 us on BASELINE configs[0].
template <int G, int U>
__global__ __launch_bounds__(ELO_BLOCK) void group_random_k(const elo_group_args a, const long total)
{
    extern __shared__ int lds_off[];
    constexpr int PER_BLOCK = ELO_BLOCK / G;
    const int lane = threadIdx.x % G;
    const int shift = (threadIdx.x % ELO_WAVE) / G * G;      // group's first lane in its wave
    const long u0 = (long)xcd_tile(blockIdx.x, gridDim.x) * PER_BLOCK + threadIdx.x / G;
    const long u = u0 < total ? u0 : total - 1;              // a group past the end shadows the last centre, stores nothing
    const Centre c = load_centre(a, u);                      // requested before the order is staged: overlaps it
    stage_offsets(lds_off, a.random_hw, a.kernel_h, a.kernel_w);
    if (u0 >= total) return;

    const int KT = a.kernel_h * a.kernel_w, K = a.K;
    const float r2 = a.distance * a.distance;
    const float *grid2 = a.xyz2 + (size_t)c.b * a.H2 * a.W2 * 3;
    int *o_sel = a.selected_bhw_idx + u * K * 3;
    float *o_mask = a.selected_mask + u * K;

    int taken = 0, seen = 0, first_hw = -1;
    bool full = false;
    if (c.ok) {
        for (int base = 0; base < KT && !full; base += G * U) {
            RawSlot raw[U];
#pragma unroll
            for (int v = 0; v < U; ++v) {
                const int i = base + v * G + lane;
                raw[v] = fetch_slot(grid2, a.H2, a.W2, lds_off[i < KT ? i : 0], c.base_h, c.base_w, i < KT);
            }
#pragma unroll
            for (int v = 0; v < U; ++v) {
                if (full || base + v * G >= KT) break;
                const Probe p = judge(raw[v], c.x, c.y, c.z, r2);
                const unsigned long long mv = group_ballot<G>(p.valid, shift);
                const unsigned long long mh = group_ballot<G>(p.hit, shift);
                if (mh == 0) { seen += __popcll(mv); continue; }
                const int before = __popcll(mh & ((1ull << lane) - 1ull));
                const int slot = taken + before;
                if (p.hit && slot < K) {
                    o_sel[slot * 3 + 0] = c.b;
                    o_sel[slot * 3 + 1] = p.hw >> 16;
                    o_sel[slot * 3 + 2] = p.hw & 0xffff;
                    o_mask[slot] = 1.0f;
                }
                if (taken == 0) first_hw = __shfl(p.hw, shift + __ffsll((long long)mh) - 1, ELO_WAVE);
                const int nh = __popcll(mh);
                if (taken + nh >= K) {
                    // the walk stops AT the K-th hit: valid pixels after it are never counted (:149-150)
                    const unsigned long long kth = group_ballot<G>(p.hit && before == K - taken - 1, shift);
                    const int kl = __ffsll((long long)kth) - 1;
                    seen += __popcll(mv & ((2ull << kl) - 1ull));
                    taken = K;
                    full = true;
                    break;
                }
                seen += __popcll(mv);
                taken += nh;
            }
        }
    }
    // slots that never got a hit: zeros, or copies of the first hit (flag_copy, :126-138)
    const bool copy = a.flag_copy == 1 && first_hw >= 0;
    for (int k = taken + lane; k < K; k += G) {
        o_sel[k * 3 + 0] = copy ? c.b : 0;
        o_sel[k * 3 + 1] = copy ? first_hw >> 16 : 0;
        o_sel[k * 3 + 2] = copy ? first_hw & 0xffff : 0;
        o_mask[k] = copy ? 1.0f : 0.0f;
    }
    if (a.valid_idx) {
        float *o = a.valid_idx + u * KT;
        for (int i = lane; i < KT; i += G) o[i] = i < seen ? 1.0f : 0.0f;
    }
    if (a.valid_in_dis_idx) {
        float *o = a.valid_in_dis_idx + u * KT;
        for (int i = lane; i < KT; i += G) o[i] = i < taken ? 1.0f : 0.0f;
    }
}

// ---------------------------------------------------------------- random-k, every pixel a centre: LDS-staged windows
// When EVERY pixel of xyz1 is a centre in row-major order (idx_n2 = get_hw_idx, utils/pointnet_util.py:23-30: the
// cost-volume and set-upconv calls, BASELINE configs[0]) neighbouring centres share almost their whole window.  A
// workgroup owns a tile of DENSE_ROWS x 64 centres, stages the union of their windows -- (rows + kH - 1) x (64 + kW - 1)
// points of the queried grid for stride 1 -- ONCE into LDS with coalesced row segments (16 bytes per point: x, y, z and
// the "non-empty" test already evaluated; rows outside the grid are staged as empty points: the reference skips both the
// same way, fused_conv_g.cu:83-86 / :106-111, neither counts as valid), and then ONE THREAD PER CENTRE walks the window in
// the caller's visiting order exactly like the reference's loop (:74-152).  All 64 lanes of a wave look at the SAME
// window offset at the same time, so a probe is one conflict-free ds_read_b128 at (lane's base + a wave-uniform offset):
// no address arithmetic, no 12-byte gathers from L2, no ballots.  Hits are collected per centre in LDS and streamed
// out as whole contiguous (64 centres x K) output rows at the end.
constexpr int DENSE_COLS = 64;

struct DenseGeom { int RH, RW; };      // staged region (rows, columns) of the queried grid per tile

__host__ __device__ inline DenseGeom dense_geom(int rows, int kH, int kW, int sh, int sw)
{
    return DenseGeom{(rows - 1) / sh + kH, (DENSE_COLS - 1) / sw + kW};
}

constexpr int DENSE_CHUNK = 8;          // window slots examined per step: their LDS reads are independent and go out together
// DENSE_ROWS x 64 centres per workgroup (one wave per tile row); hit lists are [K][threads + 1]: slot-major with an odd
// pitch -> conflict-free for the per-centre writes and for the row-wise read-out
template <int DENSE_ROWS>
__global__ __launch_bounds__(DENSE_ROWS * 64) void group_random_k_dense(const elo_group_args a)
{
    extern __shared__ int lds[];
    constexpr int THREADS = DENSE_ROWS * 64, DENSE_SEL_PITCH = THREADS + 1;
    const int KT = a.kernel_h * a.kernel_w, K = a.K;
    const int KTp = (KT + DENSE_CHUNK - 1) / DENSE_CHUNK * DENSE_CHUNK;
    const DenseGeom g = dense_geom(DENSE_ROWS, a.kernel_h, a.kernel_w, a.stride_h, a.stride_w);
    const int cells = g.RH * g.RW;
    int *lds_off = lds;                                              // [KTp] visiting order as region offsets (padding: 0)
    float4 *region = reinterpret_cast<float4 *>(lds + KTp);          // [RH * RW] x, y, z, and in .w the bits of (h << 16) | w of
                                                                     // the staged point, -1 for an empty pixel / a row outside the grid
    int *sel = reinterpret_cast<int *>(region + cells);              // [K][THREADS + 1] packed hw of the hits, -1 = none
    int *count = sel + K * DENSE_SEL_PITCH;                          // [2][THREADS] num_valid, num_select of every centre
    const int tid = threadIdx.x, lane = tid & 63, wrow = tid >> 6;
    const int b = blockIdx.z, r0 = blockIdx.y * DENSE_ROWS, c0 = blockIdx.x * DENSE_COLS;
    const int h0 = div_stride(r0, a.stride_h) - a.kernel_h / 2, w0 = div_stride(c0, a.stride_w) - a.kernel_w / 2;    // region origin (unwrapped)
    // centre of this thread (requested first: overlaps the staging)
    const int hc = r0 + wrow, wc = c0 + lane;
    const bool live = hc < a.H && wc < a.W;
    const float *cp = a.xyz1 + (((size_t)b * a.H + (live ? hc : 0)) * a.W + (live ? wc : 0)) * 3;
    const float cx = cp[0], cy = cp[1], cz = cp[2];
    {   // stage the region: consecutive threads along a row of the queried grid; all of a thread's loads go out first
        const float *grid2 = a.xyz2 + (size_t)b * a.H2 * a.W2 * 3;
        constexpr int PER = 8;                                        // points per thread in flight
        for (int e0 = 0; e0 < cells; e0 += PER * THREADS) {
            float x[PER], y[PER], z[PER];
            int hw[PER];
            bool in[PER];
#pragma unroll
            for (int v = 0; v < PER; ++v) {
                const int e = min(e0 + v * THREADS + tid, cells - 1);
                const int rr = e / g.RW, cc = e - rr * g.RW;
                const int h = h0 + rr;
                int w = (w0 + cc) % a.W2;
                if (w < 0) w += a.W2;                                 // the cylindrical wrap (:89-97; kW/2 <= W2: one wrap = modulo)
                in[v] = h >= 0 && h < a.H2;
                hw[v] = (h << 16) | w;
                const float *q = grid2 + ((size_t)(in[v] ? h : 0) * a.W2 + w) * 3;
                x[v] = q[0]; y[v] = q[1]; z[v] = q[2];
            }
#pragma unroll
            for (int v = 0; v < PER; ++v) {
                const int e = e0 + v * THREADS + tid;
                if (e < cells) {
                    const bool ok = in[v] && !(sq3(x[v], y[v], z[v]) <= ELO_EPS);       // :106-111
                    region[e] = float4{x[v], y[v], z[v], __int_as_float(ok ? hw[v] : -1)};
                }
            }
        }
        for (int i = tid; i < KTp; i += THREADS) {                    // visiting order -> offset in the region (float4 units)
            const int p = a.random_hw[i < KT ? i : 0];
            lds_off[i] = i < KT ? (p / a.kernel_w) * g.RW + (p % a.kernel_w) : 0;
        }
        for (int k = 0; k < K; ++k) sel[k * DENSE_SEL_PITCH + tid] = -1;
    }
    (void)h0; (void)w0;
    __syncthreads();
    // this thread's window origin inside the region: slot (dh, dw) of the window is region[(bh + dh) * RW + bw + dw] with
    // dh, dw counted from the window's top-left corner (the -kH/2, -kW/2 of :80-81 is in h0, w0)
    const int bh = div_stride(hc, a.stride_h) - (div_stride(r0, a.stride_h)), bw = div_stride(wc, a.stride_w) - (div_stride(c0, a.stride_w));
    const int origin = bh * g.RW + bw;
    const float r2 = a.distance * a.distance;
    const bool centre_ok = live && !(pick_max(sq3(cx, cy, cz), ELO_EPS) <= ELO_EPS);           // :62-70
    int taken = 0, seen = 0;
    bool done = !centre_ok;
    for (int i0 = 0; i0 < KT; i0 += DENSE_CHUNK) {
        if (__all(done)) break;                                       // wave-uniform: every centre of the row is finished
        int off[DENSE_CHUNK];
        float4 q[DENSE_CHUNK];
#pragma unroll
        for (int v = 0; v < DENSE_CHUNK; ++v) off[v] = lds_off[i0 + v];          // the same word for all lanes: LDS broadcasts
#pragma unroll
        for (int v = 0; v < DENSE_CHUNK; ++v) q[v] = region[origin + off[v]];    // one conflict-free ds_read_b128 each
#pragma unroll
        for (int v = 0; v < DENSE_CHUNK; ++v) {
            if (i0 + v >= KT) break;                                  // (uniform)
            // branch-free up to the hit: a non-empty pixel counts as valid (:115-116), one within the radius is taken
            // (:118-145); the walk of this centre stops at its K-th hit (:149-150)
            const int hw = __float_as_int(q[v].w);
            const bool valid = !done && hw >= 0;
            const float d = pick_max(sq3(__fsub_rn(cx, q[v].x), __fsub_rn(cy, q[v].y), __fsub_rn(cz, q[v].z)), ELO_EPS);
            const bool hit = valid && !(d > r2);
            seen += valid;
            if (hit) {
                sel[taken * DENSE_SEL_PITCH + tid] = hw;
                ++taken;
            }
            done = done || taken >= K;
        }
    }
    if (a.flag_copy == 1 && taken > 0 && taken < K) {                 // :126-138: the first hit fills the empty slots
        const int first = sel[tid];
        for (int k = taken; k < K; ++k) sel[k * DENSE_SEL_PITCH + tid] = first;
    }
    count[tid] = seen;
    count[THREADS + tid] = taken;
    __syncthreads();
    // stream the tile's outputs: the centres of a tile row are consecutive in npoints order, so each output of a row is
    // ONE contiguous span, written as flat 4-byte words by consecutive threads
    const int ncol = min(DENSE_COLS, a.W - c0);
    for (int row = 0; row < DENSE_ROWS; ++row) {
        const int hr = r0 + row;
        if (hr >= a.H) break;
        const long u0 = ((long)b * a.H + hr) * a.W + c0;              // first centre of the row (npoints == H * W)
        // a thread per output slot (c, k): 12 + 4 bytes; with THREADS a multiple of K the slot's k never changes and c steps
        // by THREADS / K: no division in the loop
        int *o_sel = a.selected_bhw_idx + u0 * K * 3;
        float *o_mask = a.selected_mask + u0 * K;
        const bool even = THREADS % K == 0;
        int c = tid / K, k = tid - c * K;
        for (int e = tid; e < ncol * K; e += THREADS) {
            if (!even) { c = e / K; k = e - c * K; }
            const int hw = sel[k * DENSE_SEL_PITCH + row * DENSE_COLS + c];
            int *o = o_sel + (long)e * 3;
            o[0] = hw < 0 ? 0 : b; o[1] = hw < 0 ? 0 : hw >> 16; o[2] = hw < 0 ? 0 : hw & 0xffff;
            o_mask[e] = hw >= 0 ? 1.0f : 0.0f;
            if (even) c += THREADS / K;
        }
        if (a.valid_idx || a.valid_in_dis_idx) {                      // prefix ones of length num_valid / num_select
            for (int e = tid; e < ncol * KT; e += THREADS) {
                const int c = e / KT, i = e - c * KT;
                if (a.valid_idx) a.valid_idx[u0 * KT + e] = i < count[row * DENSE_COLS + c] ? 1.0f : 0.0f;
                if (a.valid_in_dis_idx) a.valid_in_dis_idx[u0 * KT + e] = i < count[THREADS + row * DENSE_COLS + c] ? 1.0f : 0.0f;
            }
        }
    }
}

// ---------------------------------------------------------------- select-k
// One wave per centre. LDS: [KT] decoded offsets (block) + per wave [KT] distance
// bits and [KT] packed hw.
// wave-private LDS words of group_select_k: the [KT] distance and payload arrays of its LDS form; at least the 2 x 64
// candidate slots of the register form's rank paths; whole 16-byte items
__host__ __device__ inline int select_wave_words(int KT) { return 2 * KT > 128 ? (2 * KT + 3) & ~3 : 128; }

__global__ __launch_bounds__(ELO_BLOCK) void group_select_k(const elo_group_args a, const long total,
                                                            const int waves_per_block)
{
    extern __shared__ int lds[];
    const int KT = a.kernel_h * a.kernel_w, K = a.K;
    int *lds_off = lds;
    const int wave = threadIdx.x / ELO_WAVE, lane = threadIdx.x % ELO_WAVE;
    const int per_wave = select_wave_words(KT);              // (the register form's rank path keeps 2 x 64 candidates here)
    unsigned *dist = (unsigned *)(lds + ((KT + 3) & ~3)) + (size_t)wave * per_wave;      // 16-byte aligned
    int *pay = (int *)dist + KT;
    const long u0 = (long)xcd_tile(blockIdx.x, gridDim.x) * waves_per_block + wave;
    const long u = u0 < total ? u0 : total - 1;
    const Centre c = load_centre(a, u);                      // requested before the order is staged: overlaps it
    stage_offsets(lds_off, a.random_hw, a.kernel_h, a.kernel_w);
    if (u0 >= total) return;
    const float r2 = a.distance * a.distance;
    const float *grid2 = a.xyz2 + (size_t)c.b * a.H2 * a.W2 * 3;
    int *o_sel = a.selected_bhw_idx + u * K * 3;
    float *o_mask = a.selected_mask + u * K;
    const unsigned FAR_BITS = __float_as_uint(ELO_FAR);

    int taken = 0, seen = 0;
    if (c.ok && a.flag_copy == 0 && select_in_registers(KT, K)) {
        // the model's call shape (flag_copy 0): the register-resident wave form shared with the fused kernels
        // (elo_group_device.h) -- no LDS scans, probes batched four-deep, ranks instead of dependent rounds
        const int count = wave_select_k(grid2, a.H2, a.W2, KT, K, lds_off, c.base_h, c.base_w, c.x, c.y, c.z, r2, dist, pay,
                                        [&](int slot, int hw) {
                                            o_sel[slot * 3 + 0] = c.b; o_sel[slot * 3 + 1] = hw >> 16; o_sel[slot * 3 + 2] = hw & 0xffff;
                                            o_mask[slot] = 1.0f;
                                        }, seen, taken);
        for (int k = count + lane; k < K; k += ELO_WAVE) {
            o_sel[k * 3 + 0] = 0; o_sel[k * 3 + 1] = 0; o_sel[k * 3 + 2] = 0;
            o_mask[k] = 0.0f;
        }
    } else if (c.ok) {
        // pass 1: slot position == visiting order (:84,:110,:123,:138)
        for (int base = 0; base < KT; base += ELO_WAVE) {
            const int i = base + lane;
            Probe p{false, false, ELO_FAR, 0};
            if (i < KT) {
                p = probe_slot(grid2, a.H2, a.W2, lds_off[i], c.base_h, c.base_w, c.x, c.y, c.z, r2);
                dist[i] = p.hit ? __float_as_uint(p.d) : FAR_BITS;   // d > 0: bit order == float order
                pay[i] = p.hit ? p.hw : 0;
            }
            seen += __popcll(__ballot(p.valid));
            taken += __popcll(__ballot(p.hit));
        }
        // pass 2: K rounds of "lowest position among the minima", then the reference's swap (:148-204)
        const int rounds = K < KT ? K : KT;
        int copy_hw = 0;
        for (int s = 0; s < rounds; ++s) {
            unsigned best = 0xffffffffu;
            int where = 0x7fffffff;
            for (int t = s + lane; t < KT; t += ELO_WAVE) {
                const unsigned d = dist[t];
                if (d < best) { best = d; where = t; }
            }
            const unsigned long long key =
                wave_min_u64(((unsigned long long)best << 32) | (unsigned)where);
            const int m = (int)(key & 0xffffffffu);
            const unsigned dm = (unsigned)(key >> 32);
            const int pm = pay[m];
            if (m != s && lane == 0) {      // element s moves to m; slots <= s are never read again
                dist[m] = dist[s];
                pay[m] = pay[s];
            }
            if (s == 0) copy_hw = pm;
            const bool ok = dm < FAR_BITS;                     // :194
            const bool copy = a.flag_copy == 1;                // :179-191 (fires even on an empty slot 0)
            if (lane < 3) {
                const int v = lane == 0 ? c.b : lane == 1 ? (ok ? pm : copy_hw) >> 16 : (ok ? pm : copy_hw) & 0xffff;
                o_sel[s * 3 + lane] = (ok || copy) ? v : 0;
            } else if (lane == 3) {
                o_mask[s] = (ok || copy) ? 1.0f : 0.0f;
            }
        }
        for (int k = rounds + lane; k < K; k += ELO_WAVE) {    // K > KT: nothing left to select
            const bool copy = a.flag_copy == 1;
            o_sel[k * 3 + 0] = copy ? c.b : 0;
            o_sel[k * 3 + 1] = copy ? copy_hw >> 16 : 0;
            o_sel[k * 3 + 2] = copy ? copy_hw & 0xffff : 0;
            o_mask[k] = copy ? 1.0f : 0.0f;
        }
    } else {
        for (int k = lane; k < K; k += ELO_WAVE) {
            o_sel[k * 3 + 0] = 0; o_sel[k * 3 + 1] = 0; o_sel[k * 3 + 2] = 0;
            o_mask[k] = 0.0f;
        }
    }
    if (a.valid_idx) {
        float *o = a.valid_idx + u * KT;
        for (int i = lane; i < KT; i += ELO_WAVE) o[i] = i < seen ? 1.0f : 0.0f;
    }
    if (a.valid_in_dis_idx) {
        float *o = a.valid_in_dis_idx + u * KT;
        for (int i = lane; i < KT; i += ELO_WAVE) o[i] = i < taken ? 1.0f : 0.0f;
    }
}

// ---------------------------------------------------------------- select-k, every pixel a centre: LDS-staged windows
// The call shape of the refinement cost volumes (utils/pointnet_util.py:49-51: every pixel of the warped frame-1 grid is a
// centre, K = 6 of a 5x15 / 7x25 / 11x41 window, distance 1000): the wave-per-centre form above spends ~870 instructions
// per centre, two thirds of them on per-probe address arithmetic (offset decode, wrap, clamp, 12-byte gather) that is
// the same for neighbouring centres shifted by one pixel.  Here a workgroup owns 64 consecutive centres of one grid
// row, stages the UNION of their windows once (kH x (64 + kW - 1) points, 16 bytes each: x, y, z and the packed (h, w),
// an empty pixel or a row outside the grid staged as x = +inf, so that its distance is +inf and no probe tests
// validity), and A LANE PER CENTRE walks the window: a probe is one conflict-free ds_read_b128 at (lane's origin + a
// wave-uniform offset) and nine arithmetic instructions for 64 centres at once.  The window's slots are dealt over the
// P waves of the workgroup (slot s in raster order -> wave s % P), so every wave samples the whole window.
//
// What the reference outputs (fused_conv_g.cu:148-204) is the K smallest in-range distances in increasing order; WHICH
// of two EQUAL distances comes first depends on the visiting order -- and only then (see elo_group_device.h).  So the
// walk needs no visiting order:
//   pass 1   every wave keeps the minimum of each of its G slot classes (P * G >= 8 disjoint classes in all): the maximum
//            of 8 class minima bounds the 8th smallest distance -- hence the (K+1)-th, K <= 7 -- from above: T;
//   pass 2   the same walk again, appending the in-range slots with d <= T (typically K + 5..15) to the centre's
//            candidate list in LDS;
//   select   L = P lanes per centre pull K + 1 minima out of the list (registers + DPP): slots 0..K-1, and the first one
//            left out.  Two equal minima anywhere among those K + 1 (or a list that overflowed) send the centre through
//            the exact wave-per-centre form (wave_select_k: reference visiting order, swap rounds) after the tile is done.
// Real scans essentially never tie; the integer-lattice tests tie on every centre and stay bit-exact through the fallback.
constexpr int SD_CAP = 32;              // candidate slots per centre

template <int L>
__device__ __forceinline__ unsigned lanes_min_u32(unsigned v)           // minimum over each aligned group of L lanes
{
    v = dpp_min_step<0xb1>(v);                                             // quad_perm:[1,0,3,2]
    v = dpp_min_step<0x4e>(v);                                             // quad_perm:[2,3,0,1]
    if (L >= 8) v = dpp_min_step<0x141>(v);                                // row_half_mirror
    if (L >= 16) v = dpp_min_step<0x140>(v);                               // row_mirror
    return v;
}

__host__ __device__ inline size_t select_dense_lds_words(int P, int kH, int kW, int sh, int sw, bool counts)
{
    const DenseGeom g = dense_geom(1, kH, kW, sh, sw);
    const int NG = P >= 8 ? P : 8, KT = kH * kW;
    return (size_t)4 * g.RH * g.RW + 2 * 64 * SD_CAP + (size_t)NG * 64 + 64 + (counts ? 2 * P * 64 : 0) + 64 +
           ((KT + 3) & ~3) + (size_t)P * 128;
}

template <int P, bool COUNTS>
__global__ __launch_bounds__(P * 64) void group_select_k_dense(const elo_group_args a)
{
    extern __shared__ int lds[];
    constexpr int THREADS = P * 64, G = P >= 8 ? 1 : 8 / P, NG = P * G, L = P, E = SD_CAP / L, CPW = 64 / P;
    const int KT = a.kernel_h * a.kernel_w, K = a.K, kW = a.kernel_w;
    const DenseGeom g = dense_geom(1, a.kernel_h, a.kernel_w, a.stride_h, a.stride_w);
    const int cells = g.RH * g.RW;
    float4 *region = reinterpret_cast<float4 *>(lds);                       // [RH * RW]
    uint2 *list = reinterpret_cast<uint2 *>(region + cells);                // [SD_CAP][64] (distance bits, packed hw), slot-major
    unsigned *gm = reinterpret_cast<unsigned *>(list + 64 * SD_CAP);        // [NG][64] class minima
    int *cnt = reinterpret_cast<int *>(gm + NG * 64);                       // [64] candidates appended
    int *part = cnt + 64;                                                   // [2][P][64] seen / in-range counts per wave (COUNTS)
    int *redo = part + (COUNTS ? 2 * P * 64 : 0);                           // [64] 1: the centre takes the exact wave form
    int *lds_off = redo + 64;                                               // [KT] decoded visiting order (fallback)
    unsigned *scratch = reinterpret_cast<unsigned *>(lds_off + ((KT + 3) & ~3));   // [P][128] wave scratch of the fallback
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.z, hc = blockIdx.y, c0 = blockIdx.x * DENSE_COLS;
    const int h0 = div_stride(hc, a.stride_h) - a.kernel_h / 2, w0 = div_stride(c0, a.stride_w) - a.kernel_w / 2;     // region origin (unwrapped)
    const int wc = c0 + lane;
    const bool live = wc < a.W;
    const float *cp = a.xyz1 + (((size_t)b * a.H + hc) * a.W + (live ? wc : a.W - 1)) * 3;
    const float cx = cp[0], cy = cp[1], cz = cp[2];
    const float *grid2 = a.xyz2 + (size_t)b * a.H2 * a.W2 * 3;
    {   // stage the region (all of a thread's loads go out before the first LDS write), the visiting order, the counters
        constexpr int PER = 4;
        for (int e0 = 0; e0 < cells; e0 += PER * THREADS) {
            float x[PER], y[PER], z[PER];
            int hw[PER];
            bool in[PER];
#pragma unroll
            for (int v = 0; v < PER; ++v) {
                const int e = min(e0 + v * THREADS + tid, cells - 1);
                const int rr = e / g.RW, cc = e - rr * g.RW;
                const int h = h0 + rr;
                int w = (w0 + cc) % a.W2;
                if (w < 0) w += a.W2;                                       // the cylindrical wrap (:89-97; kW/2 <= W2: one wrap = modulo)
                in[v] = h >= 0 && h < a.H2;
                hw[v] = (h << 16) | w;
                const float *q = grid2 + ((size_t)(in[v] ? h : 0) * a.W2 + w) * 3;
                x[v] = q[0]; y[v] = q[1]; z[v] = q[2];
            }
#pragma unroll
            for (int v = 0; v < PER; ++v) {
                const int e = e0 + v * THREADS + tid;
                if (e < cells) {
                    const bool ok = in[v] && !(sq3(x[v], y[v], z[v]) <= ELO_EPS);                 // :106-111
                    region[e] = ok ? float4{x[v], y[v], z[v], __int_as_float(hw[v])} : float4{INFINITY, 0.0f, 0.0f, __int_as_float(-1)};
                }
            }
        }
        const int hh = a.kernel_h / 2, hw2 = kW / 2;
        for (int i = tid; i < KT; i += THREADS) {
            const int p = a.random_hw[i];
            lds_off[i] = ((p / kW - hh) << 16) | ((p % kW - hw2) & 0xffff);
        }
        if (tid < 64) { cnt[tid] = 0; redo[tid] = 0; }
    }
    __syncthreads();
    const int origin = wc / a.stride_w - c0 / a.stride_w;                  // this centre's window: region[dh * RW + origin + dw]
    const float r2 = a.distance * a.distance;
    const bool centre_ok = live && !(pick_max(sq3(cx, cy, cz), ELO_EPS) <= ELO_EPS);              // :62-70
    constexpr int U = 4;                                                    // probes in flight per lane
    // the walk of this wave: slots wave, wave + P, ... in raster order; fn(u, q, active) per probe, u = its position in the batch
    auto walk = [&](auto fn) {
        int dh = wave / kW, dw = wave - dh * kW;                            // (scalar: derived from the wave index)
        for (int s0 = wave; s0 < KT; s0 += P * U) {
            float4 q[U];
            bool act[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                act[u] = s0 + u * P < KT;
                q[u] = region[(act[u] ? dh * g.RW + dw : 0) + origin];
                dw += P;
                while (dw >= kW) { dw -= kW; ++dh; }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) fn(u, q[u], act[u]);
        }
    };
    auto dist_bits = [&](const float4 &q) {                                 // d > 0 (or +inf): bit order == float order
        return __float_as_uint(pick_max(sq3(__fsub_rn(cx, q.x), __fsub_rn(cy, q.y), __fsub_rn(cz, q.z)), ELO_EPS));
    };
    // ---- pass 1: class minima (class = wave, and for P = 4 the probe's parity inside the wave's walk)
    {
        unsigned m[G];
        int seen = 0, taken = 0;
#pragma unroll
        for (int j = 0; j < G; ++j) m[j] = 0xffffffffu;
        walk([&](int u, const float4 &q, bool act) {
            const unsigned d = dist_bits(q);
            const unsigned dd = act ? d : 0xffffffffu;
            m[u % G] = dd < m[u % G] ? dd : m[u % G];
            if (COUNTS) {
                const bool valid = centre_ok && act && __float_as_int(q.w) >= 0;        // (a skipped centre's masks stay 0: :62-70)
                seen += valid;
                taken += valid && !(__uint_as_float(d) > r2);
            }
        });
#pragma unroll
        for (int j = 0; j < G; ++j) gm[(wave * G + j) * 64 + lane] = m[j];
        if (COUNTS) { part[wave * 64 + lane] = seen; part[(P + wave) * 64 + lane] = taken; }
    }
    __syncthreads();
    // ---- the bound T: the maximum of 8 class minima (P = 16: classes merged in pairs first)
    unsigned T = 0;
    if (centre_ok) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            unsigned v = gm[j * 64 + lane];
            if (NG == 16) { const unsigned w2 = gm[(j + 8) * 64 + lane]; v = w2 < v ? w2 : v; }
            T = v > T ? v : T;
        }
    }
    // ---- pass 2: candidates = in-range slots with d <= T (an invalid centre has T = 0: none)
    walk([&](int u, const float4 &q, bool act) {
        const unsigned d = dist_bits(q);
        const bool c = act && d <= T && !(__uint_as_float(d) > r2) && __float_as_int(q.w) >= 0;
        if (c) {
            const int slot = atomicAdd(&cnt[lane], 1);
            if (slot < SD_CAP) list[slot * 64 + lane] = uint2{d, (unsigned)__float_as_int(q.w)};
        }
    });
    __syncthreads();
    // ---- selection: L lanes per centre, wave w takes centres w * CPW .. ; lane j of the group holds entries j, j + L, ...
    {
        const int c = wave * CPW + lane / L, j = lane % L;
        const int total = cnt[c], n = total < SD_CAP ? total : SD_CAP;
        unsigned ed[E];
        int ep[E];
#pragma unroll
        for (int i = 0; i < E; ++i) {
            const int slot = i * L + j;
            const uint2 v = list[(slot < n ? slot : 0) * 64 + c];
            ed[i] = slot < n ? v.x : 0xffffffffu;
            ep[i] = (int)v.y;
        }
        const int cw = c0 + c;
        const bool clive = cw < a.W;
        const long uo = (((long)b * a.H + hc) * a.W + (clive ? cw : 0)) * K;
        int *o_sel = a.selected_bhw_idx + uo * 3;
        float *o_mask = a.selected_mask + uo;
        bool tie = total > SD_CAP;
        unsigned prev = 0xffffffffu;
        for (int s = 0; s <= K; ++s) {
            unsigned bd = 0xffffffffu;
            int bi = 0, bp = 0;
#pragma unroll
            for (int i = 0; i < E; ++i)
                if (ed[i] < bd) { bd = ed[i]; bi = i; bp = ep[i]; }
            const unsigned md = lanes_min_u32<L>(bd);
            const bool some = md != 0xffffffffu;
            const bool mine = some && bd == md;
            const unsigned long long bal = __ballot(mine);
            const unsigned grp = (unsigned)(bal >> (lane & ~(L - 1))) & ((1u << L) - 1u);
            tie = tie || (some && (__popc(grp) > 1 || md == prev));
            prev = md;
            const bool owner = mine && (grp & ((1u << j) - 1u)) == 0;      // the lowest lane holding the minimum
            if (owner) {
#pragma unroll
                for (int i = 0; i < E; ++i)
                    if (i == bi) ed[i] = 0xffffffffu;                       // consumed
            }
            if (s < K && clive) {
                if (owner) {
                    o_sel[s * 3 + 0] = b; o_sel[s * 3 + 1] = bp >> 16; o_sel[s * 3 + 2] = bp & 0xffff;
                    o_mask[s] = 1.0f;
                } else if (!some && j == 0) {
                    o_sel[s * 3 + 0] = 0; o_sel[s * 3 + 1] = 0; o_sel[s * 3 + 2] = 0;
                    o_mask[s] = 0.0f;
                }
            }
        }
        if (tie && j == 0 && clive) redo[c] = 1;
    }
    __syncthreads();
    // ---- the prefix masks (:115-117, :132): ones of length num_valid / num_select per centre
    if (COUNTS && (a.valid_idx || a.valid_in_dis_idx)) {
        const int ncol = min(DENSE_COLS, a.W - c0);
        const long u0 = ((long)b * a.H + hc) * a.W + c0;
        for (int e = tid; e < ncol * KT; e += THREADS) {
            const int c = e / KT, i = e - c * KT;
            int seen = 0, taken = 0;
            for (int w = 0; w < P; ++w) { seen += part[w * 64 + c]; taken += part[(P + w) * 64 + c]; }
            if (a.valid_idx) a.valid_idx[u0 * KT + e] = i < seen ? 1.0f : 0.0f;
            if (a.valid_in_dis_idx) a.valid_in_dis_idx[u0 * KT + e] = i < taken ? 1.0f : 0.0f;
        }
    }
    // ---- exact ties / overflowed lists: the reference's own walk, a wave per centre
    for (int c = wave; c < 64; c += P) {
        if (!__builtin_amdgcn_readfirstlane(redo[c])) continue;
        const int cw = c0 + c;
        const float *pc = a.xyz1 + (((size_t)b * a.H + hc) * a.W + cw) * 3;
        const float x = pc[0], y = pc[1], z = pc[2];
        const long uo = (((long)b * a.H + hc) * a.W + cw) * K;
        int *o_sel = a.selected_bhw_idx + uo * 3;
        float *o_mask = a.selected_mask + uo;
        unsigned *dist = scratch + wave * 128;
        const int count = wave_select_k(grid2, a.H2, a.W2, KT, K, lds_off, div_stride(hc, a.stride_h), div_stride(cw, a.stride_w), x, y, z, r2, dist,
                                        reinterpret_cast<int *>(dist) + 64,
                                        [&](int slot, int hw) {
                                            o_sel[slot * 3 + 0] = b; o_sel[slot * 3 + 1] = hw >> 16; o_sel[slot * 3 + 2] = hw & 0xffff;
                                            o_mask[slot] = 1.0f;
                                        });
        for (int k = count + lane; k < K; k += ELO_WAVE) {
            o_sel[k * 3 + 0] = 0; o_sel[k * 3 + 1] = 0; o_sel[k * 3 + 2] = 0;
            o_mask[k] = 0.0f;
        }
    }
}

int check_args(const elo_group_args *a, const char *who, bool dense = false)
{
    if (!a) return fail(ELO_ERR_ARG, "%s: null argument block", who);
    if (a->batch < 0 || a->H <= 0 || a->W <= 0 || a->H2 <= 0 || a->W2 <= 0)
        return fail(ELO_ERR_ARG, "%s: bad grid sizes", who);
    if (a->npoints <= 0) return fail(ELO_ERR_ARG, "%s: FusedConv expects positive npoints", who);
    if (a->kernel_h <= 0) return fail(ELO_ERR_ARG, "%s: FusedConv expects positive kernel_size_H", who);
    if (a->kernel_w <= 0) return fail(ELO_ERR_ARG, "%s: FusedConv expects positive kernel_size_W", who);
    if (a->K <= 0) return fail(ELO_ERR_ARG, "%s: FusedConv expects positive K", who);
    if (a->flag_copy != 0 && a->flag_copy != 1) return fail(ELO_ERR_ARG, "%s: FusedConv expects 0 OR 1 flag_copy", who);
    if (!(a->distance > 0.0f)) return fail(ELO_ERR_ARG, "%s: FusedConv expects positive distance", who);
    if (a->stride_h <= 0) return fail(ELO_ERR_ARG, "%s: FusedConv expects positive stride_h", who);
    if (a->stride_w <= 0) return fail(ELO_ERR_ARG, "%s: FusedConv expects positive stride_w", who);
    if (a->H2 != (a->H + a->stride_h - 1) / a->stride_h || a->W2 != (a->W + a->stride_w - 1) / a->stride_w)
        return fail(ELO_ERR_ARG, "%s: expects (batch, ceil(H/stride_h), ceil(W/stride_w), 3) xyz2 shape", who);
    if ((long)a->kernel_h * a->kernel_w > ELO_MAX_WINDOW)
        return fail(ELO_ERR_LIMIT, "%s: kernel window %dx%d exceeds %d slots", who, a->kernel_h, a->kernel_w, ELO_MAX_WINDOW);
    if (a->kernel_w / 2 > a->W2)
        return fail(ELO_ERR_LIMIT, "%s: kernel_size_W/2 = %d exceeds the queried width %d (single wrap)", who, a->kernel_w / 2, a->W2);
    if (a->H2 >= 32768 || a->W2 >= 65536) return fail(ELO_ERR_LIMIT, "%s: queried grid larger than 32767 x 65535", who);
    if (!a->xyz1 || !a->xyz2 || (!a->idx_n2 && !dense) || !a->random_hw || !a->selected_bhw_idx || !a->selected_mask)
        return fail(ELO_ERR_ARG, "%s: null tensor pointer", who);
    if (dense && a->npoints != a->H * a->W)
        return fail(ELO_ERR_ARG, "%s: every pixel is a centre: npoints must equal H*W", who);
    return ELO_OK;
}

}  // namespace
}  // namespace elo

extern "C" int elo_fused_conv_random_k(const elo_group_args *a, elo_stream_t stream)
{
    using namespace elo;
    if (int rc = check_args(a, "elo_fused_conv_random_k")) return rc;
    const long total = (long)a->batch * a->npoints;
    if (total == 0) return ELO_OK;
    const int KT = a->kernel_h * a->kernel_w;
    const size_t lds = sizeof(int) * KT;
    hipStream_t s = (hipStream_t)stream;
    const auto grid = [&](int G) { return dim3((unsigned)((total + ELO_BLOCK / G - 1) / (ELO_BLOCK / G))); };
    if (KT <= 16)
        hipLaunchKernelGGL((group_random_k<16, 1>), grid(16), dim3(ELO_BLOCK), lds, s, *a, total);
    else if (KT <= 32)
        hipLaunchKernelGGL((group_random_k<32, 1>), grid(32), dim3(ELO_BLOCK), lds, s, *a, total);
    else if (a->K <= 16)
        hipLaunchKernelGGL((group_random_k<16, 4>), grid(16), dim3(ELO_BLOCK), lds, s, *a, total);
    else
        hipLaunchKernelGGL((group_random_k<32, 2>), grid(32), dim3(ELO_BLOCK), lds, s, *a, total);
    return check_launch("elo_fused_conv_random_k");
}

// LDS bytes of the dense form with `rows` x 64 centres per workgroup, 0 = it does not fit
static size_t dense_lds_bytes(const elo_group_args *a, int rows)
{
    using namespace elo;
    const int KT = a->kernel_h * a->kernel_w, threads = rows * 64;
    const DenseGeom g = dense_geom(rows, a->kernel_h, a->kernel_w, a->stride_h, a->stride_w);
    const int KTp = (KT + DENSE_CHUNK - 1) / DENSE_CHUNK * DENSE_CHUNK;
    const size_t bytes = sizeof(int) * KTp + sizeof(float4) * (size_t)g.RH * g.RW +
                         sizeof(int) * ((size_t)(threads + 1) * a->K + 2 * threads);
    return bytes <= 64 * 1024 ? bytes : 0;
}

extern "C" int elo_fused_conv_random_k_dense(const elo_group_args *a, elo_stream_t stream)
{
    using namespace elo;
    const char *who = "elo_fused_conv_random_k_dense";
    if (int rc = check_args(a, who, true)) return rc;
    if (a->batch == 0) return ELO_OK;
    // 4 rows per workgroup share more of the window; 2 rows give twice the workgroups (small grids, large windows)
    const int forced = tuning().random_dense_rows;
    const long tiles4 = (long)((a->W + DENSE_COLS - 1) / DENSE_COLS) * ((a->H + 3) / 4) * a->batch;
    int rows = forced ? forced : (tiles4 >= 1024 && dense_lds_bytes(a, 4) ? 4 : 2);
    if (rows == 4 && !dense_lds_bytes(a, 4)) rows = 2;
    const size_t lds = dense_lds_bytes(a, rows);
    if ((rows != 2 && rows != 4) || lds == 0)
        return fail(ELO_ERR_LIMIT, "%s: window %dx%d with K = %d does not fit the LDS tile (use elo_fused_conv_random_k)",
                    who, a->kernel_h, a->kernel_w, a->K);
    const dim3 grid((unsigned)((a->W + DENSE_COLS - 1) / DENSE_COLS), (unsigned)((a->H + rows - 1) / rows), (unsigned)a->batch);
    if (rows == 4) hipLaunchKernelGGL(group_random_k_dense<4>, grid, dim3(256), lds, (hipStream_t)stream, *a);
    else hipLaunchKernelGGL(group_random_k_dense<2>, grid, dim3(128), lds, (hipStream_t)stream, *a);
    return check_launch(who);
}

extern "C" int elo_fused_conv_select_k(const elo_group_args *a, elo_stream_t stream)
{
    using namespace elo;
    if (int rc = check_args(a, "elo_fused_conv_select_k")) return rc;
    const long total = (long)a->batch * a->npoints;
    if (total == 0) return ELO_OK;
    const int KT = a->kernel_h * a->kernel_w;
    // 4 waves per block while the per-wave window state fits comfortably in LDS
    int wpb = 4;
    const size_t per_wave = (size_t)select_wave_words(KT), order = ((size_t)KT + 3) & ~(size_t)3;
    while (wpb > 1 && sizeof(int) * (order + per_wave * wpb) > 64 * 1024) wpb >>= 1;
    const size_t lds = sizeof(int) * (order + per_wave * wpb);
    const unsigned grid = (unsigned)((total + wpb - 1) / wpb);
    hipLaunchKernelGGL(group_select_k, dim3(grid), dim3(wpb * ELO_WAVE), lds, (hipStream_t)stream, *a, total, wpb);
    return check_launch("elo_fused_conv_select_k");
}

extern "C" int elo_debug_select_dense_waves(int waves)      // (shorthand for elo_set_tuning: elo_tuning.select_dense_waves)
{
    const int prev = elo::tuning().select_dense_waves;
    elo::tuning().select_dense_waves = waves > 0 ? waves : elo::tuning_base().select_dense_waves;     // 0: back to elo_set_tuning's value
    return prev;
}

// select-k for the call shape "every pixel a centre" (see group_select_k_dense): K <= 7, flag_copy 0, windows up to 512
// slots whose union over 64 centres fits the LDS tile
extern "C" int elo_fused_conv_select_k_dense(const elo_group_args *a, elo_stream_t stream)
{
    using namespace elo;
    const char *who = "elo_fused_conv_select_k_dense";
    if (int rc = check_args(a, who, true)) return rc;
    if (a->batch == 0) return ELO_OK;
    const int KT = a->kernel_h * a->kernel_w;
    if (a->K > 7 || a->flag_copy != 0 || KT > 512)
        return fail(ELO_ERR_LIMIT, "%s: K = %d, flag_copy = %d, window %dx%d outside the dense form (K <= 7, flag_copy 0, <= 512 slots): "
                    "use elo_fused_conv_select_k", who, a->K, a->flag_copy, a->kernel_h, a->kernel_w);
    const bool counts = a->valid_idx || a->valid_in_dis_idx;
    const long tiles = (long)((a->W + DENSE_COLS - 1) / DENSE_COLS) * a->H * a->batch;
    // waves per tile: few tiles -> many waves each (latency: a wave's walk is KT / P probes, twice), many tiles -> 4
    const int forced = tuning().select_dense_waves;
    int P = forced ? forced : tiles >= 1024 ? 4 : tiles >= 256 ? 8 : 16;
    if (P != 4 && P != 8 && P != 16) return fail(ELO_ERR_ARG, "%s: elo_tuning.select_dense_waves must be 4, 8 or 16", who);
    const size_t lds = sizeof(int) * select_dense_lds_words(P, a->kernel_h, a->kernel_w, a->stride_h, a->stride_w, counts);
    if (lds > 64 * 1024)
        return fail(ELO_ERR_LIMIT, "%s: window %dx%d does not fit the LDS tile (use elo_fused_conv_select_k)", who, a->kernel_h, a->kernel_w);
    const dim3 grid((unsigned)((a->W + DENSE_COLS - 1) / DENSE_COLS), (unsigned)a->H, (unsigned)a->batch);
    hipStream_t s = (hipStream_t)stream;
#define ELO_SD(PW)                                                                                              \
    do {                                                                                                        \
        if (counts) hipLaunchKernelGGL((group_select_k_dense<PW, true>), grid, dim3(PW * 64), lds, s, *a);      \
        else hipLaunchKernelGGL((group_select_k_dense<PW, false>), grid, dim3(PW * 64), lds, s, *a);            \
    } while (0)
    if (P == 4) ELO_SD(4);
    else if (P == 8) ELO_SD(8);
    else ELO_SD(16);
#undef ELO_SD
    return check_launch(who);
}

// ---------------------------------------------------------------- visiting orders of a captured forward
// The reference draws tf.random_shuffle(tf.range(KT)) inside every operator on every sess.run
// (utils/pointnet_util.py:45,104,193,270).  A captured hipGraph bakes the ADDRESSES of the order tensors (and of their
// decoded (dh, dw) forms) into its kernel arguments, so fresh orders per replay mean fresh CONTENTS at fixed addresses:
// all orders of a forward live in one flat buffer, `pool` holds R pre-drawn versions of it, and one workgroup copies
// version (cursor % R) in, decodes it, and advances the cursor, which lives on the device -- as this kernel,
// or riding on the last launch of a forward (elo_pose_head_args.next_orders: the orders of the NEXT replay, no launch
// of its own).  table[e] = (offset, KT, kH, kW) of entry e, entry_of[i] = the entry slot i belongs to.
__global__ void perm_refresh_kernel(const elo_perm_refresh_args a) { elo::perm_refresh_block(a); }

extern "C" int elo_perm_refresh(const elo_perm_refresh_args *a, elo_stream_t stream)
{
    using namespace elo;
    const char *who = "elo_perm_refresh";
    if (!a || !a->pool || !a->cursor || !a->flat || !a->decoded || !a->entry_of || !a->table)
        return fail(ELO_ERR_ARG, "%s: null pointer", who);
    if (a->versions <= 0 || a->total < 0) return fail(ELO_ERR_ARG, "%s: bad sizes", who);
    if (a->total == 0) return ELO_OK;
    hipLaunchKernelGGL(perm_refresh_kernel, dim3(1), dim3(ELO_BLOCK), 0, (hipStream_t)stream, *a);
    return check_launch(who);
}
