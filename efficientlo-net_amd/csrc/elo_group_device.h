// elo_group_device.h -- device-side pieces of the neighbour grouping shared by the stand-alone grouping
// kernels (elo_grouping.hip) and the fused kernels that group in-kernel (elo_fused.hip).
// Reference semantics: tf_ops/2d_conv_random_k/fused_conv_g.cu:13-156, tf_ops/2d_conv_select_k/fused_conv_g.cu:11-209.
#pragma once
#include "elo_common.h"

namespace elo {

#define ELO_EPS 1e-10f
#define ELO_FAR 1e10f

struct Probe {
    bool valid;   // in-grid, non-empty pixel
    bool hit;     // valid and within the radius
    float d;      // clamped squared distance (hit or not)
    int hw;       // (h << 16) | w of the probed pixel
};

// One window slot of one centre. fused_conv_g.cu:80-123.
__device__ __forceinline__ Probe probe_slot(const float *__restrict__ grid2, int H2, int W2, int off,
                                            int base_h, int base_w, float cx, float cy, float cz, float r2)
{
    Probe p{false, false, ELO_FAR, 0};
    int h = base_h + (off >> 16);
    int w = base_w + (int)(short)(off & 0xffff);
    if (h < 0 || h >= H2) return p;
    if (w < 0) w += W2;
    if (w >= W2) w -= W2;
    const float *q = grid2 + ((size_t)h * W2 + w) * 3;
    const float qx = q[0], qy = q[1], qz = q[2];
    if (sq3(qx, qy, qz) <= ELO_EPS) return p;
    p.valid = true;
    p.d = pick_max(sq3(__fsub_rn(cx, qx), __fsub_rn(cy, qy), __fsub_rn(cz, qz)), ELO_EPS);
    p.hit = !(p.d > r2);
    p.hw = (h << 16) | w;
    return p;
}

// Decode the visiting order once per block: LDS[i] = (dh << 16) | (dw & 0xffff).
__device__ __forceinline__ void stage_offsets(int *lds_off, const int *__restrict__ perm, int kH, int kW)
{
    const int KT = kH * kW, hh = kH / 2, hw = kW / 2;
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
template <class Emit>
__device__ __forceinline__ int wave_random_k(const float *__restrict__ grid2, int H2, int W2, int KT, int K,
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

// K nearest in-range neighbours, reference tie order (selection sort with swaps). dist/pay: [KT] wave-private LDS.
template <class Emit>
__device__ __forceinline__ int wave_select_k(const float *__restrict__ grid2, int H2, int W2, int KT, int K,
                                             const int *lds_off, int base_h, int base_w, float cx, float cy, float cz,
                                             float r2, unsigned *dist, int *pay, Emit emit)
{
    const int lane = threadIdx.x & 63;
    const unsigned FAR_BITS = __float_as_uint(ELO_FAR);
    for (int base = 0; base < KT; base += ELO_WAVE) {
        const int i = base + lane;
        if (i < KT) {
            const Probe p = probe_slot(grid2, H2, W2, lds_off[i], base_h, base_w, cx, cy, cz, r2);
            dist[i] = p.hit ? __float_as_uint(p.d) : FAR_BITS;
            pay[i] = p.hit ? p.hw : 0;
        }
    }
    const int rounds = K < KT ? K : KT;
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

}  // namespace elo
