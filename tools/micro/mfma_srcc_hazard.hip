// How many wait states does gfx950 need between an MFMA and a LATER MFMA OF ANOTHER SHAPE that reads its result as SrcC?
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_srcc_hazard.hip -o /tmp/mfma_srcc_hazard && /tmp/mfma_srcc_hazard
// Found by bisecting the chain kernels' wrong rows (finding 4; gpurun_out/r04_bisect): hipcc
// (ROCm 7.2, clang 22) emitted
//     v_mfma_f32_16x16x32_f16 v[22:25], ...            ; a pair step of a layer's accumulator
//     ds_read_b128 ... ; s_waitcnt lgkmcnt(1)          ; two wait states
//     v_mfma_f32_16x16x16_f16 v[38:41], ..., v[22:25]  ; the 16-k tail step: SrcC = that accumulator
// and the second instruction read the accumulator as it was BEFORE the first one's update whenever the s_waitcnt did not
// happen to stall.  LLVM's hazard recognizer asks for 0 wait states when SrcC is exactly the previous MFMA's vDst
// (GCNHazardRecognizer::checkMAIHazards90A, "FullReg"), which holds for two MFMAs of the SAME shape (the hardware forwards /
// interlocks) -- this table measures what holds for mixed shapes.
// One wave, fixed registers, inline asm (nothing is rescheduled, no compiler-inserted nops): per iteration
//     A: acc  = A_shape(ones, ones, acc)           (+32 or +16 per element)
//     <gap: n wait states: nothing, or s_nop n-1>
//     B: out  = B_shape(ones, ones, acc)           (out = acc + 16 or + 32);  "same vDst": out is acc itself
//     s_nop 15 x2                                  (everything retired before the next iteration)
// and the exact expected value of `out` after 64 iterations is compared.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));

#define CLOB "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119"
#define MF32(dst, c) "v_mfma_f32_16x16x32_f16 " dst ", %0, %0, " c "\n\t"
#define MF16(dst, c) "v_mfma_f32_16x16x16_f16 " dst ", %1, %1, " c "\n\t"
#define DRAIN "s_nop 15\n\ts_nop 15\n\t"

// SHAPES: 0 = 16x16x32 then 16x16x16 (the kernels' pair -> tail), 1 = 16x16x16 then 16x16x32, 2 = 16x16x32 twice, 3 = 16x16x16 twice
template <int SHAPES, int GAP, bool SAME_DST>
__global__ void k(float *out, int iters)
{
    h8 x8; h4 x4;
    for (int i = 0; i < 8; ++i) x8[i] = (_Float16)1.0f;
    for (int i = 0; i < 4; ++i) x4[i] = (_Float16)1.0f;
    asm volatile("v_mov_b32 v112, 0\n\tv_mov_b32 v113, 0\n\tv_mov_b32 v114, 0\n\tv_mov_b32 v115, 0\n\t"
                 "v_mov_b32 v116, 0\n\tv_mov_b32 v117, 0\n\tv_mov_b32 v118, 0\n\tv_mov_b32 v119, 0\n\t" DRAIN ::: CLOB);
    for (int it = 0; it < iters; ++it) {
#define GAPTXT(n) (n == 0 ? "" : "")
#define EMIT(A, B, G) asm volatile(A G B DRAIN :: "v"(x8), "v"(x4) : CLOB)
#define WITH_GAP(A, B)                                                                                       \
        do {                                                                                                 \
            if constexpr (GAP == 0) EMIT(A, B, "");                                                          \
            else if constexpr (GAP == 1) EMIT(A, B, "s_nop 0\n\t");                                          \
            else if constexpr (GAP == 2) EMIT(A, B, "s_nop 1\n\t");                                          \
            else if constexpr (GAP == 3) EMIT(A, B, "s_nop 2\n\t");                                          \
            else if constexpr (GAP == 4) EMIT(A, B, "s_nop 3\n\t");                                          \
            else if constexpr (GAP == 5) EMIT(A, B, "s_nop 4\n\t");                                          \
            else if constexpr (GAP == 6) EMIT(A, B, "s_nop 5\n\t");                                          \
            else if constexpr (GAP == 7) EMIT(A, B, "s_nop 6\n\t");                                          \
            else if constexpr (GAP == 8) EMIT(A, B, "s_nop 7\n\t");                                          \
            else if constexpr (GAP == 10) EMIT(A, B, "s_nop 9\n\t");                                         \
            else EMIT(A, B, "s_nop 11\n\t");                                                                 \
        } while (0)
#define DST (SAME_DST ? 1 : 0)
        if constexpr (SHAPES == 0) { if constexpr (SAME_DST) WITH_GAP(MF32("v[112:115]", "v[112:115]"), MF16("v[112:115]", "v[112:115]")); else WITH_GAP(MF32("v[112:115]", "v[112:115]"), MF16("v[116:119]", "v[112:115]")); }
        if constexpr (SHAPES == 1) { if constexpr (SAME_DST) WITH_GAP(MF16("v[112:115]", "v[112:115]"), MF32("v[112:115]", "v[112:115]")); else WITH_GAP(MF16("v[112:115]", "v[112:115]"), MF32("v[116:119]", "v[112:115]")); }
        if constexpr (SHAPES == 2) { if constexpr (SAME_DST) WITH_GAP(MF32("v[112:115]", "v[112:115]"), MF32("v[112:115]", "v[112:115]")); else WITH_GAP(MF32("v[112:115]", "v[112:115]"), MF32("v[116:119]", "v[112:115]")); }
        if constexpr (SHAPES == 3) { if constexpr (SAME_DST) WITH_GAP(MF16("v[112:115]", "v[112:115]"), MF16("v[112:115]", "v[112:115]")); else WITH_GAP(MF16("v[112:115]", "v[112:115]"), MF16("v[116:119]", "v[112:115]")); }
    }
    float r;
    if constexpr (SAME_DST) asm volatile(DRAIN "v_mov_b32 %0, v112" : "=v"(r) :: CLOB);
    else asm volatile(DRAIN "v_mov_b32 %0, v116" : "=v"(r) :: CLOB);
    out[threadIdx.x] = r;
}

// ---- what counts as a wait state: the gap made of other instructions (16x16x32 -> 16x16x16, another vDst)
// KIND: 0 = n x s_mov_b32 (SALU), 1 = n x v_mov_b32 (VALU), 2 = ONE independent v_mfma_f32_16x16x32_f16 + (n - 1) x s_nop 0,
//       3 = ONE independent v_mfma_f32_16x16x16_f16 + (n - 1) x s_nop 0, 4 = n x s_nop 0
template <int KIND, int N>
__global__ void kgap(float *out, int iters)
{
    h8 x8; h4 x4;
    for (int i = 0; i < 8; ++i) x8[i] = (_Float16)1.0f;
    for (int i = 0; i < 4; ++i) x4[i] = (_Float16)1.0f;
    asm volatile("v_mov_b32 v112, 0\n\tv_mov_b32 v113, 0\n\tv_mov_b32 v114, 0\n\tv_mov_b32 v115, 0\n\t"
                 "v_mov_b32 v116, 0\n\tv_mov_b32 v117, 0\n\tv_mov_b32 v118, 0\n\tv_mov_b32 v119, 0\n\t"
                 "v_mov_b32 v120, 0\n\tv_mov_b32 v121, 0\n\tv_mov_b32 v122, 0\n\tv_mov_b32 v123, 0\n\tv_mov_b32 v124, 0\n\t" DRAIN ::: CLOB, "v120", "v121", "v122", "v123", "v124", "s40");
    for (int it = 0; it < iters; ++it) {
        asm volatile(MF32("v[112:115]", "v[112:115]") :: "v"(x8), "v"(x4) : CLOB);
        if constexpr (KIND == 2) asm volatile(MF32("v[120:123]", "v[120:123]") :: "v"(x8), "v"(x4) : "v120", "v121", "v122", "v123");
        if constexpr (KIND == 3) asm volatile(MF16("v[120:123]", "v[120:123]") :: "v"(x8), "v"(x4) : "v120", "v121", "v122", "v123");
#pragma unroll
        for (int j = (KIND == 2 || KIND == 3) ? 1 : 0; j < N; ++j) {
            if constexpr (KIND == 0) asm volatile("s_mov_b32 s40, 0" ::: "s40");
            else if constexpr (KIND == 1) asm volatile("v_mov_b32 v124, v124" ::: "v124");
            else asm volatile("s_nop 0");
        }
        asm volatile(MF16("v[116:119]", "v[112:115]") DRAIN :: "v"(x8), "v"(x4) : CLOB);
    }
    float r;
    asm volatile(DRAIN "v_mov_b32 %0, v116" : "=v"(r) :: CLOB);
    out[threadIdx.x] = r;
}

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int SHAPES, int GAP, bool SAME_DST>
static char cell(float *d, int iters)
{
    const float a = (SHAPES == 0 || SHAPES == 2) ? 32.0f : 16.0f, b = (SHAPES == 1 || SHAPES == 2) ? 32.0f : 16.0f;
    const float want = SAME_DST ? iters * (a + b) : iters * a + b;
    hipLaunchKernelGGL((k<SHAPES, GAP, SAME_DST>), dim3(1), dim3(64), 0, 0, d, iters);
    CHECK(hipDeviceSynchronize());
    float h[64];
    CHECK(hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost));
    int wrong = 0;
    for (int i = 0; i < 64; ++i) wrong += h[i] != want;
    return wrong == 0 ? '.' : wrong == 64 ? 'X' : 'x';
}

template <int KIND, int N>
static char gcell(float *d)
{
    const int iters = 64;
    hipLaunchKernelGGL((kgap<KIND, N>), dim3(1), dim3(64), 0, 0, d, iters);
    CHECK(hipDeviceSynchronize());
    float h[64];
    CHECK(hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost));
    int wrong = 0;
    for (int i = 0; i < 64; ++i) wrong += h[i] != iters * 32.0f + 16.0f;
    return wrong == 0 ? '.' : wrong == 64 ? 'X' : 'x';
}
template <int KIND>
static void grow(const char *name, float *d)
{
    printf("%-58s", name);
    if constexpr (KIND < 2 || KIND == 4) printf("  %c", gcell<KIND, 0>(d)); else printf("   ");
    printf("  %c  %c  %c  %c  %c  %c  %c  %c\n", gcell<KIND, 1>(d), gcell<KIND, 2>(d), gcell<KIND, 3>(d), gcell<KIND, 4>(d), gcell<KIND, 5>(d),
           gcell<KIND, 6>(d), gcell<KIND, 7>(d), gcell<KIND, 8>(d));
    fflush(stdout);
}

template <int SHAPES, bool SAME_DST>
static void row(const char *name, float *d)
{
    const int iters = 64;
    printf("%-58s", name);
    printf("  %c", cell<SHAPES, 0, SAME_DST>(d, iters)); printf("  %c", cell<SHAPES, 1, SAME_DST>(d, iters));
    printf("  %c", cell<SHAPES, 2, SAME_DST>(d, iters)); printf("  %c", cell<SHAPES, 3, SAME_DST>(d, iters));
    printf("  %c", cell<SHAPES, 4, SAME_DST>(d, iters)); printf("  %c", cell<SHAPES, 5, SAME_DST>(d, iters));
    printf("  %c", cell<SHAPES, 6, SAME_DST>(d, iters)); printf("  %c", cell<SHAPES, 7, SAME_DST>(d, iters));
    printf("  %c", cell<SHAPES, 8, SAME_DST>(d, iters)); printf("   %c", cell<SHAPES, 10, SAME_DST>(d, iters));
    printf("   %c\n", cell<SHAPES, 12, SAME_DST>(d, iters));
    fflush(stdout);
}

int main()
{
    float *d;
    CHECK(hipMalloc(&d, 64 * 4));
    printf("wait states between the two MFMAs ('.' = right, 'X' = every lane wrong, 'x' = some lanes wrong)\n");
    printf("%-58s  0  1  2  3  4  5  6  7  8  10  12\n", "first -> second (second reads the first's vDst as SrcC)");
    row<0, false>("16x16x32_f16 -> 16x16x16_f16, another vDst (the kernels')", d);
    row<0, true>("16x16x32_f16 -> 16x16x16_f16, same vDst", d);
    row<1, false>("16x16x16_f16 -> 16x16x32_f16, another vDst", d);
    row<1, true>("16x16x16_f16 -> 16x16x32_f16, same vDst", d);
    row<2, false>("16x16x32_f16 -> 16x16x32_f16, another vDst", d);
    row<2, true>("16x16x32_f16 -> 16x16x32_f16, same vDst", d);
    row<3, false>("16x16x16_f16 -> 16x16x16_f16, another vDst", d);
    row<3, true>("16x16x16_f16 -> 16x16x16_f16, same vDst", d);
    printf("\n16x16x32_f16 -> 16x16x16_f16 (another vDst), the gap made of n other instructions:\n");
    printf("%-58s  0  1  2  3  4  5  6  7  8\n", "");
    grow<4>("n x s_nop 0", d);
    grow<0>("n x s_mov_b32 (SALU)", d);
    grow<1>("n x v_mov_b32 (VALU)", d);
    grow<2>("one independent 16x16x32 MFMA, then (n - 1) x s_nop 0", d);
    grow<3>("one independent 16x16x16 MFMA, then (n - 1) x s_nop 0", d);
    return 0;
}
