// Which VALU / LDS instructions of ANOTHER wave overlap with MFMAs on a gfx950 SIMD?  
//   hipcc --offload-arch=gfx950 -O3 tools/micro/coexec.hip -o /tmp/coexec && /tmp/coexec
// One workgroup of 8 waves on one CU, two waves per SIMD (wave w and w + 4).  Waves 0-3 run 4 independent
// v_mfma_f32_16x16x32_f16 per iteration (16 cycles each), waves 4-7 run 16 independent instructions of ONE kind per
// iteration (inline asm: the compiler cannot pack or fold them).  Three launches per kind: MFMA waves alone, the other
// waves alone, both.  overlap = (t_mfma + t_other - t_both) / min(t_mfma, t_other): 1 = they run side by side, 0 = one after
// the other.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

#define OPS(X) X(0, "v_fma_f32 %0, %0, %1, %2") X(1, "v_mul_f32 %0, %0, %1") X(2, "v_add_f32 %0, %0, %1") \
    X(3, "v_cvt_pkrtz_f16_f32 %0, %0, %1") X(4, "v_fma_mix_f32 %0, %0, %1, %2 op_sel_hi:[1,0,0]") X(5, "v_max_i32 %0, %0, %1") \
    X(6, "v_mov_b32 %0, %1") X(7, "v_exp_f32 %0, %0") X(8, "v_max_f32 %0, %0, %1") X(9, "v_pk_add_f32 %3, %3, %3") \
    X(10, "v_cndmask_b32 %0, %0, %1, vcc") X(11, "v_cvt_f32_f16 %0, %0") X(12, "v_pk_fma_f32 %3, %3, %3, %3") \
    X(13, "v_xor_b32 %0, %0, %1") X(14, "v_pk_mul_f32 %3, %3, %3") X(15, "v_perm_b32 %0, %0, %1, %2") X(16, "v_pk_max_f16 %0, %0, %1") \
    X(17, "v_lshl_add_u32 %0, %0, 1, %1") X(18, "v_and_or_b32 %0, %0, %1, %2") X(19, "v_pk_add_f16 %0, %0, %1") X(20, "v_mad_u32_u24 %0, %0, %1, %2") \
    X(21, "v_mul_lo_u32 %0, %0, %1") X(22, "v_mul_hi_u32 %0, %0, %1") X(23, "v_mul_u32_u24 %0, %0, %1") X(24, "v_mad_u64_u32 %3, vcc, %0, %1, %3")
constexpr int NOPS = 25;

template <int OP> __device__ __forceinline__ void op16(float (&v)[16], float2 (&p)[16], float a, float b)
{
#define X(n, text) if constexpr (OP == n) { _Pragma("unroll") for (int u = 0; u < 16; ++u) asm volatile(text : "+v"(v[u]) : "v"(a), "v"(b), "v"(p[u])); }
    OPS(X)
#undef X
}

template <int OP, int CHAINS>
__global__ __launch_bounds__(512) void k(float *out, int iters, int who, long long *ticks)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(lane * 0.001f + i); b[i] = (_Float16)(i * 0.01f); }
    f4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    float v[16];
    float2 p[16];
    for (int i = 0; i < 16; ++i) { v[i] = lane + i; p[i] = float2{1.0f + i, 0.5f}; }
    const float ca = 1.0001f + lane * 1e-6f, cb = 0.25f;
    __syncthreads();
    const long long t0 = wall_clock64();
    if (wave < 4) {
        if (who & 1)
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j % CHAINS] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[j % CHAINS], 0, 0, 0);
            }
    } else if (who & 2) {
        for (int it = 0; it < iters; ++it) op16<OP>(v, p, ca, cb);
    }
    const long long t1 = wall_clock64();
    float s = 0;
    for (int j = 0; j < 4; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
    for (int u = 0; u < 16; ++u) s += v[u] + p[u].x;
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if (lane == 0) ticks[wave] = t1 - t0;
}

template <int OP, int CHAINS> float run(float *out, long long *ticks, int iters, int who)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<OP, CHAINS>), dim3(1), dim3(512), 0, 0, out, iters, who, ticks);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<OP, CHAINS>), dim3(1), dim3(512), 0, 0, out, iters, who, ticks);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f;
}

template <int OP, int CHAINS = 4> void report(float *out, long long *ticks, const char *text)
{
    const int iters = 20000;
    const float m = run<OP, CHAINS>(out, ticks, iters, 1), o = run<OP, CHAINS>(out, ticks, iters, 2), both = run<OP, CHAINS>(out, ticks, iters, 3);
    printf("%d accumulator chains | %-48s mfma alone %7.1f us  other alone %7.1f us (%.1f ns per instruction)  both %7.1f us  overlap %.2f\n", CHAINS, text, m, o,
           o * 1e3 / (16.0 * iters), both, (m + o - both) / (m < o ? m : o));
}

int main()
{
    float *out; long long *ticks;
    hipMalloc(&out, 1024 * 512 * 4); hipMalloc(&ticks, 16 * 8);
#define X(n, text) report<n>(out, ticks, text);
    OPS(X)
#undef X
    // the same with the MFMAs in 2 and in 1 dependent accumulator chains (the register-resident kernels run 2)
    report<5, 2>(out, ticks, "v_max_i32"); report<3, 2>(out, ticks, "v_cvt_pkrtz_f16_f32"); report<4, 2>(out, ticks, "v_fma_mix_f32");
    report<5, 1>(out, ticks, "v_max_i32"); report<3, 1>(out, ticks, "v_cvt_pkrtz_f16_f32");
    return 0;
}
