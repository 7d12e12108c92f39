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
    c.b = (int)(u / a.npoints);
    const int hc = a.idx_n2[u * 2 + 0], wc = a.idx_n2[u * 2 + 1];
    const float *p = a.xyz1 + (((size_t)c.b * a.H + hc) * a.W + wc) * 3;
    c.x = p[0]; c.y = p[1]; c.z = p[2];
    c.ok = !(pick_max(sq3(c.x, c.y, c.z), ELO_EPS) <= ELO_EPS);      // :62-70
    c.base_h = hc / a.stride_h;
    c.base_w = wc / a.stride_w;
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

// ---------------------------------------------------------------- select-k
// One wave per centre. LDS: [KT] decoded offsets (block) + per wave [KT] distance
// bits and [KT] packed hw.
__global__ __launch_bounds__(ELO_BLOCK) void group_select_k(const elo_group_args a, const long total,
                                                            const int waves_per_block)
{
    extern __shared__ int lds[];
    const int KT = a.kernel_h * a.kernel_w, K = a.K;
    int *lds_off = lds;
    const int wave = threadIdx.x / ELO_WAVE, lane = threadIdx.x % ELO_WAVE;
    unsigned *dist = (unsigned *)(lds + KT) + (size_t)wave * 2 * KT;
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

    if (c.ok && a.flag_copy == 0 && !a.valid_idx && !a.valid_in_dis_idx && select_in_registers(KT, K)) {
        // the model's call shape (flag_copy 0, prefix masks not requested): the register-resident wave form shared with
        // the fused kernels (elo_group_device.h) -- no LDS scans, probes batched four-deep
        const int count = wave_select_k(grid2, a.H2, a.W2, KT, K, lds_off, c.base_h, c.base_w, c.x, c.y, c.z, r2, dist, pay,
                                        [&](int slot, int hw) {
                                            o_sel[slot * 3 + 0] = c.b; o_sel[slot * 3 + 1] = hw >> 16; o_sel[slot * 3 + 2] = hw & 0xffff;
                                            o_mask[slot] = 1.0f;
                                        });
        for (int k = count + lane; k < K; k += ELO_WAVE) {
            o_sel[k * 3 + 0] = 0; o_sel[k * 3 + 1] = 0; o_sel[k * 3 + 2] = 0;
            o_mask[k] = 0.0f;
        }
        return;
    }
    int taken = 0, seen = 0;
    if (c.ok) {
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

int check_args(const elo_group_args *a, const char *who)
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
    if (!a->xyz1 || !a->xyz2 || !a->idx_n2 || !a->random_hw || !a->selected_bhw_idx || !a->selected_mask)
        return fail(ELO_ERR_ARG, "%s: null tensor pointer", who);
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

extern "C" int elo_fused_conv_select_k(const elo_group_args *a, elo_stream_t stream)
{
    using namespace elo;
    if (int rc = check_args(a, "elo_fused_conv_select_k")) return rc;
    const long total = (long)a->batch * a->npoints;
    if (total == 0) return ELO_OK;
    const int KT = a->kernel_h * a->kernel_w;
    // 4 waves per block while the per-wave window state fits comfortably in LDS
    int wpb = 4;
    while (wpb > 1 && sizeof(int) * KT * (1 + 2 * (size_t)wpb) > 64 * 1024) wpb >>= 1;
    const size_t lds = sizeof(int) * KT * (1 + 2 * (size_t)wpb);
    const unsigned grid = (unsigned)((total + wpb - 1) / wpb);
    hipLaunchKernelGGL(group_select_k, dim3(grid), dim3(wpb * ELO_WAVE), lds, (hipStream_t)stream, *a, total, wpb);
    return check_launch("elo_fused_conv_select_k");
}
