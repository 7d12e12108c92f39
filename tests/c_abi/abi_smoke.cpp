// tests/c_abi/abi_smoke.cpp -- the C ABI of libelo_hip.so driven WITHOUT Python or torch: plain hipMalloc'd
// buffers in, the two grouping entry points, results compared bit for bit with the CPU oracle
// (oracle/libelo_oracle.so).  Built and run by tests/test_c_abi_gpu.py.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/elo.h"

extern "C" int elo_oracle_fused_conv_random_k(int, int, int, int, int, int, int, int, float, int, int, const float *,
                                              const float *, const int *, const int *, int *, float *, float *, float *,
                                              int, int, int);
extern "C" int elo_oracle_fused_conv_select_k(int, int, int, int, int, int, int, int, float, int, int, const float *,
                                              const float *, const int *, const int *, int *, float *, float *, float *,
                                              int, int, int);

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 2; } } while (0)

template <typename T> static T *to_device(const std::vector<T> &v)
{
    T *p = nullptr;
    if (hipMalloc(&p, v.size() * sizeof(T)) != hipSuccess) return nullptr;
    hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice);
    return p;
}

int main()
{
    const int B = 2, H = 16, W = 225, kH = 7, kW = 11, K = 8, KT = kH * kW, N = H * W;
    srand(7);
    std::vector<float> xyz((size_t)B * H * W * 3);
    for (int b = 0; b < B; ++b)
        for (int h = 0; h < H; ++h)
            for (int w = 0; w < W; ++w) {
                float *p = &xyz[(((size_t)b * H + h) * W + w) * 3];
                const float az = 6.2831853f * w / W, r = 8.0f + 2.0f * sinf(3 * az) + 0.01f * (rand() % 100);
                p[0] = r * cosf(az); p[1] = r * sinf(az); p[2] = -0.1f * h;
                if (rand() % 10 == 0) p[0] = p[1] = p[2] = 0.0f;              // holes
            }
    std::vector<int> idx((size_t)B * N * 2), perm(KT);
    for (int b = 0; b < B; ++b)
        for (int n = 0; n < N; ++n) { idx[((size_t)b * N + n) * 2] = n / W; idx[((size_t)b * N + n) * 2 + 1] = n % W; }
    for (int i = 0; i < KT; ++i) perm[i] = i;
    for (int i = KT - 1; i > 0; --i) { int j = rand() % (i + 1); int t = perm[i]; perm[i] = perm[j]; perm[j] = t; }

    float *d_xyz = to_device(xyz);
    int *d_idx = to_device(idx), *d_perm = to_device(perm);
    int *d_sel; float *d_valid, *d_indis, *d_mask;
    HIP_OK(hipMalloc(&d_sel, (size_t)B * N * K * 3 * sizeof(int)));
    HIP_OK(hipMalloc(&d_valid, (size_t)B * N * KT * sizeof(float)));
    HIP_OK(hipMalloc(&d_indis, (size_t)B * N * KT * sizeof(float)));
    HIP_OK(hipMalloc(&d_mask, (size_t)B * N * K * sizeof(float)));
    hipStream_t stream;
    HIP_OK(hipStreamCreate(&stream));

    int failures = 0;
    for (int op = 0; op < 2; ++op) {
        const float dist = op == 0 ? 1.0f : 1000.0f;
        elo_group_args a;
        memset(&a, 0, sizeof a);
        a.batch = B; a.H = H; a.W = W; a.H2 = H; a.W2 = W; a.npoints = N; a.kernel_h = kH; a.kernel_w = kW; a.K = K;
        a.flag_copy = 0; a.distance = dist; a.stride_h = 1; a.stride_w = 1;
        a.xyz1 = d_xyz; a.xyz2 = d_xyz; a.idx_n2 = d_idx; a.random_hw = d_perm;
        a.selected_bhw_idx = d_sel; a.valid_idx = d_valid; a.valid_in_dis_idx = d_indis; a.selected_mask = d_mask;
        // outputs deliberately NOT zero-filled: the kernels write every element (no cudaMemset, fused_conv.cpp:154-166)
        hipMemsetAsync(d_sel, 0x5a, (size_t)B * N * K * 3 * sizeof(int), stream);
        const int rc = op == 0 ? elo_fused_conv_random_k(&a, stream) : elo_fused_conv_select_k(&a, stream);
        if (rc != ELO_OK) { printf("entry point failed: %s\n", elo_last_error()); return 3; }
        HIP_OK(hipStreamSynchronize(stream));
        std::vector<int> sel((size_t)B * N * K * 3), o_sel(sel.size());
        std::vector<float> valid((size_t)B * N * KT), indis(valid.size()), mask((size_t)B * N * K);
        std::vector<float> o_valid(valid.size()), o_indis(valid.size()), o_mask(mask.size());
        HIP_OK(hipMemcpy(sel.data(), d_sel, sel.size() * sizeof(int), hipMemcpyDeviceToHost));
        HIP_OK(hipMemcpy(valid.data(), d_valid, valid.size() * sizeof(float), hipMemcpyDeviceToHost));
        HIP_OK(hipMemcpy(indis.data(), d_indis, indis.size() * sizeof(float), hipMemcpyDeviceToHost));
        HIP_OK(hipMemcpy(mask.data(), d_mask, mask.size() * sizeof(float), hipMemcpyDeviceToHost));
        auto oracle = op == 0 ? elo_oracle_fused_conv_random_k : elo_oracle_fused_conv_select_k;
        oracle(B, H, W, N, kH, kW, K, 0, dist, 1, 1, xyz.data(), xyz.data(), idx.data(), perm.data(), o_sel.data(),
               o_valid.data(), o_indis.data(), o_mask.data(), H, W, 8);
        const bool same = sel == o_sel && valid == o_valid && indis == o_indis && mask == o_mask;
        printf("%s: %s\n", op == 0 ? "elo_fused_conv_random_k" : "elo_fused_conv_select_k", same ? "bit-exact" : "MISMATCH");
        failures += !same;
    }
    // error convention: a bad attribute comes back as a status + message, nothing is launched
    elo_group_args bad;
    memset(&bad, 0, sizeof bad);
    bad.batch = 1; bad.H = bad.H2 = 4; bad.W = bad.W2 = 8; bad.npoints = 1; bad.kernel_h = 3; bad.kernel_w = 5; bad.K = 0;
    bad.distance = 1.0f; bad.stride_h = bad.stride_w = 1;
    if (elo_fused_conv_random_k(&bad, stream) != ELO_ERR_ARG || !strstr(elo_last_error(), "positive K")) { printf("error path broken\n"); ++failures; }
    printf(failures ? "FAIL\n" : "PASS\n");
    return failures ? 1 : 0;
}
