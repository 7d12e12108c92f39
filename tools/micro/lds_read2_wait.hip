// Is `s_waitcnt lgkmcnt(1)` enough for the OLDER of two outstanding LDS reads when the older one is a two-address read
// (ds_read2st64_b64) and the younger a ds_read_b128?  (finding 4: the instruction pair the bisect of
// the chain kernel's wrong rows ended at -- tools/rr_bisect.sh, gpurun_out/r04_bisect/.)
//   hipcc --offload-arch=gfx950 -O3 tools/micro/lds_read2_wait.hip -o /tmp/lds_read2_wait && /tmp/lds_read2_wait
// The environment is the chain kernels' (tools/micro/rr_ring.hip): eight waves per workgroup, two workgroups per CU, a
// two-slot LDS ring written with ds_write_b128 by the even waves, one `s_waitcnt lgkmcnt(0); s_barrier` per superstep.
// Every LDS word holds the fp16 pair (1.0, 1.0).  Per superstep every wave runs, as ONE inline-asm block on fixed
// registers (so that nothing is rescheduled):
//     v_mov  R2[0..3] = -1 (fp16 NaNs)          <- what a too-early consumer would see
//     ds_read_b128      R1 <- slot              (older)
//     <READ R2>         R2 <- slot              ds_read2st64_b64, or two ds_read_b64, or one ds_read_b128
//     s_waitcnt lgkmcnt(1 or 2)                 -> R1 has arrived
//     v_mfma_f32_16x16x32_f16 acc0 += R1 * x
//     ds_read_b128      R3 <- slot              (younger than R2)
//     s_waitcnt lgkmcnt(1)                      -> R2 has arrived, R3 may be outstanding  [the instruction in question]
//     <GAP: s_nop n or nothing>
//     v_mfma_f32_16x16x16_f16 acc1 += R2[0:1] * x   (or v_mov capture of R2[0])
//     s_waitcnt lgkmcnt(0)
// acc1 turning NaN (or the captured word being -1) means R2[0:1] was consumed before the LDS data landed.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));

constexpr int SLOT_U4 = 512, STAGE = 3;

// R1 = v[100:103], R2 = v[104:107], R3 = v[108:111], acc0 = v[112:115], acc1 = v[116:119], capture = v120
#define CLOBBERS "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", \
                 "v115", "v116", "v117", "v118", "v119", "v120", "memory"

template <int READ2, int GAP, int CONSUMER>
__device__ __forceinline__ void consume(unsigned addr, h8 x8, h4 x4)
{
    // READ2: 0 = ds_read2st64_b64, 1 = two ds_read_b64, 2 = ds_read_b128 (16 bytes at the first address), 3 = ds_read2_b64 (adjacent)
    // CONSUMER: 0 = 16x16x16 MFMA on R2[0:1], 1 = v_mov capture
#define READ_R2_0 "ds_read2st64_b64 v[104:107], %0 offset0:8 offset1:12\n\t"
#define READ_R2_1 "ds_read_b64 v[104:105], %0 offset:4096\n\tds_read_b64 v[106:107], %0 offset:6144\n\t"
#define READ_R2_2 "ds_read_b128 v[104:107], %0 offset:4096\n\t"
#define READ_R2_3 "ds_read2_b64 v[104:107], %0 offset0:64 offset1:65\n\t"
#define BODY(READ_R2, WAIT_R1, WAIT_R2, GAPTXT, CONS)                                                                   \
    asm volatile("v_mov_b32 v104, -1\n\tv_mov_b32 v105, -1\n\tv_mov_b32 v106, -1\n\tv_mov_b32 v107, -1\n\t"            \
                 "ds_read_b128 v[100:103], %0\n\t" READ_R2 "s_waitcnt lgkmcnt(" WAIT_R1 ")\n\t"                           \
                 "v_mfma_f32_16x16x32_f16 v[112:115], v[100:103], %1, v[112:115]\n\t"                                   \
                 "ds_read_b128 v[108:111], %0 offset:2048\n\t"                                                          \
                 "s_waitcnt lgkmcnt(" WAIT_R2 ")\n\t" GAPTXT CONS "s_waitcnt lgkmcnt(0)\n\t"                              \
                 "s_nop 7\n\t"                                                                                          \
                 :: "v"(addr), "v"(x8), "v"(x4) : CLOBBERS)
#define CONS_0 "v_mfma_f32_16x16x16_f16 v[116:119], v[104:105], %2, v[116:119]\n\t"
#define CONS_1 "v_or_b32 v120, v120, v104\n\t"
#define CONS_2 "v_mfma_f32_16x16x16_f16 v[116:119], v[104:105], %2, v[112:115]\n\t"      /* the kernel's: SrcC = the 16x16x32's result, two instructions earlier */
#define GAP_0 ""
#define GAP_1 "s_nop 0\n\t"
#define GAP_2 "s_nop 3\n\t"
#define GAP_3 "s_nop 7\n\t"
#define PICK_CONS(R, W1, W2, G)                                                \
    do { if constexpr (CONSUMER == 0) BODY(R, W1, W2, G, CONS_0); else if constexpr (CONSUMER == 1) BODY(R, W1, W2, G, CONS_1); \
         else BODY(R, W1, W2, G, CONS_2); } while (0)
#define PICK_GAP(R, W1, W2)                                                    \
    do { if constexpr (GAP == 0) PICK_CONS(R, W1, W2, GAP_0); else if constexpr (GAP == 1) PICK_CONS(R, W1, W2, GAP_1);   \
         else if constexpr (GAP == 2) PICK_CONS(R, W1, W2, GAP_2); else PICK_CONS(R, W1, W2, GAP_3); } while (0)
    if constexpr (CONSUMER == 3) {
        asm volatile("ds_read_b128 v[100:103], %0\n\tds_read2st64_b64 v[104:107], %0 offset0:8 offset1:12\n\ts_waitcnt lgkmcnt(0)\n\ts_nop 7\n\t"
                     "v_mfma_f32_16x16x32_f16 v[112:115], v[100:103], %1, v[112:115]\n\t"
                     "s_nop 1\n\t"
                     "v_mfma_f32_16x16x16_f16 v[116:119], v[104:105], %2, v[112:115]\n\t"
                     "s_nop 7\n\t"
                     :: "v"(addr), "v"(x8), "v"(x4) : CLOBBERS);
        return;
    }
    if constexpr (READ2 == 0) PICK_GAP(READ_R2_0, "1", "1");
    else if constexpr (READ2 == 1) PICK_GAP(READ_R2_1, "2", "2");       // (R2 is two instructions: the older must be done, two may be outstanding)
    else if constexpr (READ2 == 2) PICK_GAP(READ_R2_2, "1", "1");
    else PICK_GAP(READ_R2_3, "1", "1");
}

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_of(const void *p)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, 0x7fffffff, 0x00020000);
}

template <int READ2, int GAP, int CONSUMER, int IDLE_SKIP>
__global__ __launch_bounds__(512, 4) void k(const uint4 *__restrict__ W, int nsuper, unsigned *bad, unsigned *bad_by_wave)
{
    extern __shared__ __align__(16) float lds[];
    uint4 *ring = reinterpret_cast<uint4 *>(lds);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool idle = wave & 1;
    const __amdgpu_buffer_rsrc_t rs = rsrc_of(W);
    uint4 st[STAGE] = {};
    auto issue = [&](int S, int reg) {
        if (S >= nsuper || (IDLE_SKIP && idle)) return;
        st[reg] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16u, (S * SLOT_U4 + (wave & ~1) * 64) * 16, 0));
    };
    auto commit = [&](int S, int reg, int slot) {
        if (S >= nsuper || idle) return;
        ring[slot * SLOT_U4 + wave * 64 + lane] = st[reg];
    };
    asm volatile("v_mov_b32 v112, 0\n\tv_mov_b32 v113, 0\n\tv_mov_b32 v114, 0\n\tv_mov_b32 v115, 0\n\tv_mov_b32 v116, 0\n\tv_mov_b32 v117, 0\n\t"
                 "v_mov_b32 v118, 0\n\tv_mov_b32 v119, 0\n\tv_mov_b32 v120, 0" ::: CLOBBERS);
    h8 x8; h4 x4;
    for (int i = 0; i < 8; ++i) x8[i] = (_Float16)1.0f;
    for (int i = 0; i < 4; ++i) x4[i] = (_Float16)1.0f;
    // every slot word that is read must hold 1.0 halves from the start (the odd chunks are never committed)
    for (int i = threadIdx.x; i < 2 * SLOT_U4; i += 512) ring[i] = uint4{0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};
    issue(0, 0); issue(1, 1); issue(2, 2);
    __syncthreads();
    commit(0, 0, 0); issue(STAGE, 0);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    for (int S0 = 0; S0 < nsuper; S0 += 6) {
#pragma unroll
        for (int u = 0; u < 6; ++u) {
            const int S = S0 + u;
            if (S < nsuper) {
                const unsigned addr = (unsigned)((u % 2) * SLOT_U4 * 16 + lane * 16);
                consume<READ2, GAP, CONSUMER>(addr, x8, x4);
                commit(S + 1, (u + 1) % STAGE, (u + 1) % 2);
                issue(S + 1 + STAGE, (u + 1) % STAGE);
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            }
        }
    }
    float a0, a1, a2, a3; unsigned cap;
    asm volatile("s_nop 7\n\ts_nop 7\n\tv_mov_b32 %0, v116\n\tv_mov_b32 %1, v117\n\tv_mov_b32 %2, v118\n\tv_mov_b32 %3, v119\n\tv_mov_b32 %4, v120"
                 : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3), "=v"(cap) :: CLOBBERS);
    const float s = a0 + a1 + a2 + a3;
    // CONSUMER 2 / 3: acc1 = acc0 + 16 with acc0 = 32 per superstep, exactly
    const bool wrong = CONSUMER == 0 ? !(s == s) : CONSUMER == 1 ? cap == 0xffffffffu : !(a0 == 32.0f * nsuper + 16.0f && a1 == a0 && a2 == a0 && a3 == a0);
    if (wrong) { atomicAdd(bad, 1u); atomicAdd(bad_by_wave + wave, 1u); }
}

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int READ2, int GAP, int CONSUMER, int IDLE_SKIP>
static void run(const char *what, const uint4 *dW, int nsuper, int wgs, int runs, unsigned *dbad)
{
    unsigned bad_runs = 0, lanes = 0, by_wave[8] = {};
    for (int r = 0; r < runs; ++r) {
        CHECK(hipMemset(dbad, 0, 9 * 4));
        hipLaunchKernelGGL((k<READ2, GAP, CONSUMER, IDLE_SKIP>), dim3(wgs), dim3(512), 37376, 0, dW, nsuper, dbad, dbad + 1);
        CHECK(hipDeviceSynchronize());
        unsigned h[9];
        CHECK(hipMemcpy(h, dbad, 9 * 4, hipMemcpyDeviceToHost));
        if (h[0]) { ++bad_runs; lanes += h[0]; for (int w = 0; w < 8; ++w) by_wave[w] += h[1 + w]; }
    }
    printf("%-86s runs with an early consumer: %2u / %d   lanes %8u   by wave:", what, bad_runs, runs, lanes);
    for (int w = 0; w < 8; ++w) printf(" %u", by_wave[w]);
    printf("\n");
    fflush(stdout);
}

int main(int argc, char **argv)
{
    const int wgs = argc > 1 ? atoi(argv[1]) : 1372, nsuper = argc > 2 ? atoi(argv[2]) : 24, runs = argc > 3 ? atoi(argv[3]) : 20;
    std::vector<uint4> hW((size_t)nsuper * SLOT_U4, uint4{0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u});
    uint4 *dW; unsigned *dbad;
    CHECK(hipMalloc(&dW, hW.size() * 16)); CHECK(hipMemcpy(dW, hW.data(), hW.size() * 16, hipMemcpyHostToDevice));
    CHECK(hipMalloc(&dbad, 9 * 4));
    printf("workgroups %d, supersteps %d\n", wgs, nsuper);
    run<0, 0, 0, 1>("ds_read2st64_b64, lgkmcnt(1), MFMA at once            (the chain kernel's pair; idle waves)", dW, nsuper, wgs, runs, dbad);
    run<0, 0, 0, 0>("ds_read2st64_b64, lgkmcnt(1), MFMA at once            (every wave loads)", dW, nsuper, wgs, runs, dbad);
    run<0, 0, 1, 1>("ds_read2st64_b64, lgkmcnt(1), v_or capture at once", dW, nsuper, wgs, runs, dbad);
    run<0, 1, 0, 1>("ds_read2st64_b64, lgkmcnt(1), s_nop 0, MFMA", dW, nsuper, wgs, runs, dbad);
    run<0, 2, 0, 1>("ds_read2st64_b64, lgkmcnt(1), s_nop 3, MFMA", dW, nsuper, wgs, runs, dbad);
    run<0, 3, 0, 1>("ds_read2st64_b64, lgkmcnt(1), s_nop 7, MFMA", dW, nsuper, wgs, runs, dbad);
    run<0, 0, 2, 1>("ds_read2st64_b64, lgkmcnt(1), MFMA whose SrcC is the previous MFMA's result (the kernel's)", dW, nsuper, wgs, runs, dbad);
    run<0, 3, 2, 1>("... the same with s_nop 7 in front of it", dW, nsuper, wgs, runs, dbad);
    run<0, 0, 3, 1>("LDS data waited for (lgkmcnt(0)), 16x16x32 MFMA, s_nop 1, 16x16x16 MFMA on its result", dW, nsuper, wgs, runs, dbad);
    run<1, 0, 0, 1>("two ds_read_b64, lgkmcnt(2), MFMA at once", dW, nsuper, wgs, runs, dbad);
    run<2, 0, 0, 1>("ds_read_b128, lgkmcnt(1), MFMA at once", dW, nsuper, wgs, runs, dbad);
    run<3, 0, 0, 1>("ds_read2_b64 (adjacent), lgkmcnt(1), MFMA at once", dW, nsuper, wgs, runs, dbad);
    run<1, 0, 1, 1>("two ds_read_b64, lgkmcnt(2), v_or capture at once", dW, nsuper, wgs, runs, dbad);
    run<2, 0, 1, 1>("ds_read_b128, lgkmcnt(1), v_or capture at once", dW, nsuper, wgs, runs, dbad);
    return 0;
}
