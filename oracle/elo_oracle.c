/*
 * oracle/elo_oracle.c -- CPU restatement of the two EfficientLO-Net grouping ops.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product path:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker / the timed CPU baseline.
 *
 * What it restates (reference file:line, all under /root/reference):
 *   random-k : tf_ops/2d_conv_random_k/fused_conv_g.cu:13-156
 *   select-k : tf_ops/2d_conv_select_k/fused_conv_g.cu:11-209
 *   zero-fill of the four outputs: tf_ops/2d_conv_random_k/fused_conv.cpp:154-166
 *
 * Pinning: this restatement is compared bit-for-bit against the reference
 * kernel bodies compiled for the host (oracle/build_ref.sh -> oracle/_ref/)
 * in tests/test_oracle_vs_ref.py (runs wherever oracle/_ref exists) and
 * against the committed golden vectors in tests/golden/ that were produced by
 * that same reference build (tests/golden/make_golden.py).
 *
 * Arithmetic contract (shared with the HIP kernels): squared distances are
 * ((dx*dx + dy*dy) + dz*dz) in fp32 with NO fused multiply-add; build with
 * -ffp-contract=off.  "max(a,b)" follows the reference host shim / CUDA
 * fmaxf on a NaN first operand: (a > b ? a : b).
 */
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ELO_EPS 1e-10f
#define ELO_FAR 1e10f

static inline float pick_max(float a, float b) { return a > b ? a : b; }

static inline float sq3(float x, float y, float z) { return x * x + y * y + z * z; }

/* One candidate of the window walk. Returns 0 = row outside the grid (padding),
 * 1 = empty pixel, 2 = valid but farther than the radius, 3 = hit.
 * fused_conv_g.cu:80-123 (random) / :79-124 (select). */
static inline int probe(const float *grid2, int H2, int W2, int kH, int kW,
                        int base_h, int base_w, int slot, float cx, float cy, float cz,
                        float r2, int *oh, int *ow, float *od)
{
    int h = base_h + slot / kW - kH / 2;
    int w = base_w + slot % kW - kW / 2;
    if (h < 0 || h >= H2) return 0;
    if (w < 0) w += W2;           /* single cylindrical wrap, either side */
    if (w >= W2) w -= W2;
    const float *q = grid2 + ((long)h * W2 + w) * 3;
    float qx = q[0], qy = q[1], qz = q[2];
    if (sq3(qx, qy, qz) <= ELO_EPS) return 1;
    float dx = cx - qx, dy = cy - qy, dz = cz - qz;
    float d = pick_max(sq3(dx, dy, dz), ELO_EPS);
    *oh = h; *ow = w; *od = d;
    return d > r2 ? 2 : 3;
}

static void centre_random(int b, int n, int H, int W, int H2, int W2, int N, int kH, int kW,
                          int K, int flag_copy, float r2, int sh, int sw,
                          const float *xyz1, const float *xyz2, const int *idx_n2,
                          const int *perm, int *sel, float *valid, float *indis, float *mask)
{
    (void)H;
    const int KT = kH * kW;
    const int hc = idx_n2[((long)b * N + n) * 2 + 0];
    const int wc = idx_n2[((long)b * N + n) * 2 + 1];
    const float *c = xyz1 + (((long)b * H + hc) * W + wc) * 3;
    if (pick_max(sq3(c[0], c[1], c[2]), ELO_EPS) <= ELO_EPS) return;   /* :64-70 */

    const float *grid2 = xyz2 + (long)b * H2 * W2 * 3;
    int *o_sel = sel + ((long)b * N + n) * K * 3;
    float *o_mask = mask + ((long)b * N + n) * K;
    float *o_valid = valid ? valid + ((long)b * N + n) * KT : NULL;
    float *o_indis = indis ? indis + ((long)b * N + n) * KT : NULL;

    int taken = 0, seen = 0;
    for (int i = 0; i < KT; ++i) {
        int h, w; float d;
        int what = probe(grid2, H2, W2, kH, kW, hc / sh, wc / sw, perm[i],
                         c[0], c[1], c[2], r2, &h, &w, &d);
        if (what < 2) continue;
        if (o_valid) o_valid[seen] = 1.0f;                             /* :115-116 */
        ++seen;
        if (what == 2) continue;
        if (flag_copy == 1 && taken == 0) {                            /* :126-138 */
            for (int k = 0; k < K; ++k) {
                o_sel[k * 3 + 0] = b; o_sel[k * 3 + 1] = h; o_sel[k * 3 + 2] = w;
                o_mask[k] = 1.0f;
            }
        }
        o_sel[taken * 3 + 0] = b; o_sel[taken * 3 + 1] = h; o_sel[taken * 3 + 2] = w;
        o_mask[taken] = 1.0f;
        if (o_indis) o_indis[taken] = 1.0f;
        if (++taken >= K) break;                                       /* :149-150 */
    }
}

static void centre_select(int b, int n, int H, int W, int H2, int W2, int N, int kH, int kW,
                          int K, int flag_copy, float r2, int sh, int sw,
                          const float *xyz1, const float *xyz2, const int *idx_n2,
                          const int *perm, int *sel, float *valid, float *indis, float *mask,
                          float *dist, int *ph, int *pw)
{
    (void)H;
    const int KT = kH * kW;
    const int hc = idx_n2[((long)b * N + n) * 2 + 0];
    const int wc = idx_n2[((long)b * N + n) * 2 + 1];
    const float *c = xyz1 + (((long)b * H + hc) * W + wc) * 3;
    if (pick_max(sq3(c[0], c[1], c[2]), ELO_EPS) <= ELO_EPS) return;

    const float *grid2 = xyz2 + (long)b * H2 * W2 * 3;
    int *o_sel = sel + ((long)b * N + n) * K * 3;
    float *o_mask = mask + ((long)b * N + n) * K;
    float *o_valid = valid ? valid + ((long)b * N + n) * KT : NULL;
    float *o_indis = indis ? indis + ((long)b * N + n) * KT : NULL;

    /* the reference keeps 5000-entry per-thread arrays (:42-50); only the
     * first KT entries are ever read, so KT entries are enough here. */
    for (int i = 0; i < KT; ++i) { dist[i] = ELO_FAR; ph[i] = 0; pw[i] = 0; }

    int taken = 0, seen = 0;
    for (int i = 0; i < KT; ++i) {          /* slot position == visit order (:84,:110,:123,:138) */
        int h, w; float d;
        int what = probe(grid2, H2, W2, kH, kW, hc / sh, wc / sw, perm[i],
                         c[0], c[1], c[2], r2, &h, &w, &d);
        if (what < 2) continue;
        if (o_valid) o_valid[seen] = 1.0f;
        ++seen;
        if (what == 2) continue;
        if (o_indis) o_indis[taken] = 1.0f;
        dist[i] = d; ph[i] = h; pw[i] = w;
        ++taken;
    }

    /* partial selection sort WITH swaps (:148-204): ties resolve to the lowest
     * current array position, and the swap itself perturbs later tie order. */
    for (int s = 0; s < K; ++s) {
        int m = s;
        for (int t = s + 1; t < KT; ++t)
            if (dist[t] < dist[m]) m = t;
        if (m != s) {
            float td = dist[m]; int th = ph[m], tw = pw[m];
            dist[m] = dist[s]; ph[m] = ph[s]; pw[m] = pw[s];
            dist[s] = td; ph[s] = th; pw[s] = tw;
        }
        if (flag_copy == 1 && s == 0) {     /* :179-191, fires even when slot 0 is empty */
            for (int k = 0; k < K; ++k) {
                o_sel[k * 3 + 0] = b; o_sel[k * 3 + 1] = ph[0]; o_sel[k * 3 + 2] = pw[0];
                o_mask[k] = 1.0f;
            }
        }
        /* s can run past KT when K > KT in the reference (reads its 5000-slot
         * arrays, all "far"); here: nothing left to emit. */
        if (s < KT && dist[s] < ELO_FAR) {
            o_sel[s * 3 + 0] = b; o_sel[s * 3 + 1] = ph[s]; o_sel[s * 3 + 2] = pw[s];
            o_mask[s] = 1.0f;
        }
    }
}

static void zero_outputs(int B, int N, int KT, int K, int *sel, float *valid, float *indis,
                         float *mask)
{
    memset(sel, 0, sizeof(int) * (size_t)B * N * K * 3);
    memset(mask, 0, sizeof(float) * (size_t)B * N * K);
    if (valid) memset(valid, 0, sizeof(float) * (size_t)B * N * KT);
    if (indis) memset(indis, 0, sizeof(float) * (size_t)B * N * KT);
}

/* Flat argument list in the order of the reference launcher
 * (tf_ops/2d_conv_random_k/fused_conv_g.cu:162); `threads` <= 1 runs scalar.
 * valid_idx / valid_in_dis_idx may be NULL. Returns 0, or -1 on a bad shape. */
int elo_oracle_fused_conv_random_k(int batch, int H, int W, int npoints, int kH, int kW, int K,
                                   int flag_copy, float distance, int stride_h, int stride_w,
                                   const float *xyz1, const float *xyz2, const int *idx_n2,
                                   const int *random_hw, int *selected_bhw_idx, float *valid_idx,
                                   float *valid_in_dis_idx, float *selected_mask, int H2, int W2,
                                   int threads)
{
    if (batch < 0 || npoints <= 0 || kH <= 0 || kW <= 0 || K <= 0 || stride_h <= 0 || stride_w <= 0)
        return -1;
    const int KT = kH * kW;
    const float r2 = distance * distance;
    zero_outputs(batch, npoints, KT, K, selected_bhw_idx, valid_idx, valid_in_dis_idx, selected_mask);
    const long total = (long)batch * npoints;
    (void)threads;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(threads > 1 ? threads : 1)
#endif
    for (long u = 0; u < total; ++u)
        centre_random((int)(u / npoints), (int)(u % npoints), H, W, H2, W2, npoints, kH, kW, K,
                      flag_copy, r2, stride_h, stride_w, xyz1, xyz2, idx_n2, random_hw,
                      selected_bhw_idx, valid_idx, valid_in_dis_idx, selected_mask);
    return 0;
}

int elo_oracle_fused_conv_select_k(int batch, int H, int W, int npoints, int kH, int kW, int K,
                                   int flag_copy, float distance, int stride_h, int stride_w,
                                   const float *xyz1, const float *xyz2, const int *idx_n2,
                                   const int *random_hw, int *selected_bhw_idx, float *valid_idx,
                                   float *valid_in_dis_idx, float *selected_mask, int H2, int W2,
                                   int threads)
{
    if (batch < 0 || npoints <= 0 || kH <= 0 || kW <= 0 || K <= 0 || stride_h <= 0 || stride_w <= 0)
        return -1;
    const int KT = kH * kW;
    if (KT > 5000) return -1;               /* the reference's array bound (:42-43) */
    const float r2 = distance * distance;
    zero_outputs(batch, npoints, KT, K, selected_bhw_idx, valid_idx, valid_in_dis_idx, selected_mask);
    const long total = (long)batch * npoints;
    (void)threads;
#ifdef _OPENMP
#pragma omp parallel num_threads(threads > 1 ? threads : 1)
#endif
    {
        float *dist = (float *)malloc(sizeof(float) * KT);
        int *ph = (int *)malloc(sizeof(int) * KT);
        int *pw = (int *)malloc(sizeof(int) * KT);
#ifdef _OPENMP
#pragma omp for schedule(static)
#endif
        for (long u = 0; u < total; ++u)
            centre_select((int)(u / npoints), (int)(u % npoints), H, W, H2, W2, npoints, kH, kW, K,
                          flag_copy, r2, stride_h, stride_w, xyz1, xyz2, idx_n2, random_hw,
                          selected_bhw_idx, valid_idx, valid_in_dis_idx, selected_mask, dist, ph, pw);
        free(dist); free(ph); free(pw);
    }
    return 0;
}
