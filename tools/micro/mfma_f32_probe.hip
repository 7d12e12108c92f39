// mfma_f32_probe.hip -- what a stream of v_mfma_f32_16x16x4_f32 reaches on one SIMD when the pieces of the training kernels
// (csrc/elo_train_dense.hip, elo_train.hip) are added one at a time: bare MFMAs -> + the A operands from LDS (ds_read_b128 per tile)
// -> + the B operands from global memory (16-byte loads one chunk ahead).  1, 2 or 4 waves per SIMD (workgroups of 256 / 512 / 1024).
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/build/mfma_f32_probe tools/micro/mfma_f32_probe.hip && tools/micro/build/mfma_f32_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// one "chunk": NT tiles x 4 steps x RB row blocks MFMAs (64 for NT = 8, RB = 2)
template <int NT, int RB, int MODE, int TBS>      // MODE 0 bare, 1 + LDS A operands, 2 + global B operands (cache-resident), 3 streaming B from
                                                  // HBM, 4 + the accumulators stored every 8 chunks (16-byte stores, 64-byte row segments), 5 + column sums
__global__ __launch_bounds__(TBS) void probe(const float *__restrict__ x, float *__restrict__ out, int iters, long stride)
{
    extern __shared__ float4 wl[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < NT * 64 * 8; i += TBS) wl[i] = float4{1.f, 0.5f, 0.25f, 0.125f};
    __syncthreads();
    f32x4 acc[RB][NT];
    for (int rb = 0; rb < RB; ++rb) for (int t = 0; t < NT; ++t) acc[rb][t] = f32x4{0, 0, 0, 0};
    f32x4 cur[RB], nxt[RB], st[NT], st2[NT];
    for (int t = 0; t < NT; ++t) st[t] = st2[t] = f32x4{0, 0, 0, 0};
    const float *p = x + ((long)blockIdx.x * TBS + threadIdx.x) * 4;
    for (int rb = 0; rb < RB; ++rb) cur[rb] = nxt[rb] = f32x4{1.f + lane, 2.f, 3.f, 4.f};
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
        if (MODE >= 2) {
            const long at = MODE >= 3 ? (long)it * gridDim.x * TBS * 4 * RB : (long)(it & 63) * stride;
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) { const float4 v = *reinterpret_cast<const float4 *>(p + at + (long)rb * (MODE >= 3 ? gridDim.x * TBS * 4 : 2048)); nxt[rb] = f32x4{v.x, v.y, v.z, v.w}; }
        }
        float4 aw[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) aw[t] = MODE >= 1 ? wl[((it & 7) * NT + t) * 64 + lane] : float4{1.f, 0.5f, 0.25f, 0.125f};
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const float as[4] = {aw[t].x, aw[t].y, aw[t].z, aw[t].w};
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) acc[rb][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(as[e], cur[rb][e], acc[rb][t], 0, 0, 0);
        }
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) cur[rb] = nxt[rb];
        if (MODE >= 4 && (it & 7) == 7) {
            float *o = out + 64 + (((long)(it >> 3) * gridDim.x + blockIdx.x) * (TBS / 64) + (threadIdx.x >> 6)) * (16 * RB * 16 * NT);
#pragma unroll
            for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const f32x4 v = acc[rb][t];
                    *reinterpret_cast<float4 *>(o + ((rb * 16 + (lane & 15)) * 16 * NT) + 16 * t + 4 * (lane >> 4)) = float4{v[0], v[1], v[2], v[3]};
                    if (MODE >= 5) { st[t] += v; st2[t] += v * v; }
                    acc[rb][t] = f32x4{0, 0, 0, 0};
                }
        }
    }
    if (MODE >= 5) for (int t = 0; t < NT; ++t) acc[0][t] += st[t] + st2[t];
    f32x4 s{0, 0, 0, 0};
    for (int rb = 0; rb < RB; ++rb) for (int t = 0; t < NT; ++t) s += acc[rb][t];
    if (s[0] + s[1] + s[2] + s[3] == 1.2345f) out[0] = s[0];
}

template <int NT, int RB, int MODE, int TBS>
static void run(const char *what, const float *x, float *out)
{
    const int iters = MODE >= 3 ? 240 * 512 / TBS * 2 / RB : 2000, grid = 256 * (1024 / TBS >= 1 ? 1 : 1);
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    auto k = probe<NT, RB, MODE, TBS>;
    const size_t lds = NT * 64 * 8 * sizeof(float4);
    hipLaunchKernelGGL(k, dim3(grid), dim3(TBS), lds, 0, x, out, 10, 4096l * 64);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a, 0));
    hipLaunchKernelGGL(k, dim3(grid), dim3(TBS), lds, 0, x, out, iters, 4096l * 64);
    CK(hipEventRecord(b, 0)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    const double mfmas = (double)grid * (TBS / 64) * iters * NT * 4 * RB;
    printf("  %-64s %2d waves/SIMD  %7.1f TF/s  (%.1f cycles per MFMA and SIMD at 2.4 GHz)\n", what, TBS / 256, mfmas * 2048 / (ms * 1e-3) / 1e12,
           ms * 1e-3 * 2.4e9 / (mfmas / (grid * 4)));
    fflush(stdout);
}

int main()
{
    float *x, *out;
    CK(hipMalloc(&x, 1l << 30)); CK(hipMemset(x, 0, 1l << 30)); CK(hipMalloc(&out, 1l << 30));
    printf("v_mfma_f32_16x16x4_f32 streams, 256 workgroups (one per CU); peak 157.3 TF/s = 32 cycles per MFMA and SIMD\n");
    run<8, 2, 0, 256>("bare: 8 tiles x 2 row blocks (64 MFMAs / chunk)", x, out);
    run<8, 2, 0, 512>("bare", x, out);
    run<8, 2, 1, 256>("+ A operands from LDS (8 ds_read_b128 / chunk)", x, out);
    run<8, 2, 1, 512>("+ A operands from LDS", x, out);
    run<8, 2, 2, 256>("+ B operands from global, one chunk ahead (2 x 16 B / chunk)", x, out);
    run<8, 2, 2, 512>("+ B operands from global, one chunk ahead", x, out);
    run<8, 2, 3, 512>("+ B streaming from HBM (every byte once)", x, out);
    run<8, 2, 4, 512>("+ accumulators stored every 8 chunks (the 128 -> 128 layer's epilogue)", x, out);
    run<8, 2, 5, 512>("+ column sums and sums of squares in registers", x, out);
    run<8, 1, 4, 1024>("8 tiles x 1 row block, streaming + stores", x, out);
    run<4, 4, 2, 512>("4 tiles x 4 row blocks, LDS + global", x, out);
    run<2, 4, 2, 1024>("2 tiles x 4 row blocks, LDS + global", x, out);
    run<8, 1, 2, 1024>("8 tiles x 1 row block, LDS + global", x, out);
    return 0;
}
