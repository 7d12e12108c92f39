// Stand-alone reproducer for DESIGN.md section 3b, finding 4: "waves that never issue a global load".
//   hipcc --offload-arch=gfx950 -O3 tools/micro/rr_ring.hip -o /tmp/rr_ring && /tmp/rr_ring [workgroups] [supersteps] [runs]
// The W stream of the register-resident chain kernels (csrc/elo_fused.hip, RrStream) on its own: a workgroup of eight
// waves walks a stream of 8 KB "supersteps"; wave w requests chunk w of superstep S + 3 into a staging register
// (buffer_load_dwordx4), drops superstep S + 1's chunk into the other slot of a two-slot LDS ring (ds_write_b128),
// joins `s_waitcnt lgkmcnt(0); s_barrier`, and every wave then reads the four even chunks of the slot back
// (ds_read_b128) as matrix operands.  In the fp16-products mode only the EVEN chunks carry data, so the odd waves have
// nothing to stage.  Every 16-byte item of the stream names itself -- word 0 = (superstep << 16 | chunk << 8 | lane) --
// so a wave that reads a slot too early or too late sees WHICH superstep it got instead, and the kernel records
// (workgroup, wave, superstep wanted, word read).  The variants switch one ingredient each (FLAGS below); all run in one
// process, the table at the end says which of them misread.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

enum : unsigned {
    IDLE_SKIP = 1,          // odd waves issue no W load at all (the failing form); without it they fetch their partner's chunk and drop it
    FULL_BARRIER = 2,       // __syncthreads() (drains vmcnt too) instead of the LDS-only barrier
    IDLE_OTHER_LOAD = 4,    // with IDLE_SKIP: the idle waves issue ONE unrelated load per superstep (a fixed address) instead
    IDLE_WRITE = 8,         // with IDLE_SKIP: the idle waves still write (zeros) to their own, unread, chunk
    NO_MFMA = 16,           // no matrix instructions: the operands are only checked
    PRIO = 32,              // s_setprio 3 in the loading waves
    IDLE_SLEEP = 64,        // with IDLE_SKIP: s_sleep 4 in the idle waves where the load would be
    HIGH_IDLE = 128,        // the idle waves are 4..7 instead of the odd ones (every SIMD then has one loading wave)
    NO_JITTER = 256,        // no dependent gather in front of the stream (all waves start together)
    BARRIER_BUILTIN = 512,  // fence(release, workgroup) + __builtin_amdgcn_s_barrier() + fence(acquire) instead of the asm
    DOUBLE_BARRIER = 1024,  // two barriers per superstep
    THREE_SLOTS = 2048,     // a third ring slot
};

struct ErrRec { unsigned wg, wave, want, got; };

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_of(const void *p)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, 0x7fffffff, 0x00020000);
}

template <unsigned F>
__device__ __forceinline__ void barrier()
{
    if constexpr (F & FULL_BARRIER) __syncthreads();
    else if constexpr (F & BARRIER_BUILTIN) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    } else asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if constexpr (F & DOUBLE_BARRIER) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

constexpr int STAGE = 3;
constexpr int SLOT_U4 = 512;

template <unsigned F>
__global__ __launch_bounds__(512, 4) void ring_kernel(const uint4 *__restrict__ W, int nsuper, float *sink, ErrRec *err, unsigned *nerr,
                                                      const int *__restrict__ jitter, int jitter_n)
{
    extern __shared__ __align__(16) float lds[];
    constexpr int NSLOT = (F & THREE_SLOTS) ? 3 : 2;
    uint4 *ring = reinterpret_cast<uint4 *>(lds);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool idle = (F & HIGH_IDLE) ? wave >= 4 : (wave & 1);
    // the chunk this wave stages: HIGH_IDLE: waves 0..3 carry chunks 0, 2, 4, 6
    const int chunk = (F & HIGH_IDLE) ? (wave & 3) * 2 : wave;
    const __amdgpu_buffer_rsrc_t rs = rsrc_of(W);
    uint4 st[STAGE] = {};
    if constexpr (F & PRIO) { if (!idle) __builtin_amdgcn_s_setprio(3); }

    auto issue = [&](int S, int reg) {
        if (S >= nsuper) return;
        if constexpr (F & IDLE_SKIP) {
            if (idle) {
                if constexpr (F & IDLE_OTHER_LOAD) st[reg] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16u, 0, 0));
                if constexpr (F & IDLE_SLEEP) __builtin_amdgcn_s_sleep(4);
                return;
            }
        }
        const int c = idle ? (chunk & ~1) : chunk;                        // the partner's kilobyte (an L1 hit), dropped later
        st[reg] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16u, (S * SLOT_U4 + c * 64) * 16, 0));
    };
    auto commit = [&](int S, int reg, int slot) {
        if (S >= nsuper) return;
        if (idle) {
            if constexpr (F & IDLE_WRITE) ring[slot * SLOT_U4 + (wave | 1) * 64 + lane] = uint4{0u, 0u, 0u, 0u};
            return;
        }
        ring[slot * SLOT_U4 + chunk * 64 + lane] = st[reg];
    };

    // a dependent gather in front of the stream, as the kernels have (idx / mask, then the rows): the waves of a workgroup
    // reach the first barrier at different times
    float seed = 0.0f;
    issue(0, 0); issue(1, 1); issue(2, 2);
    if constexpr (!(F & NO_JITTER)) {
        int j = jitter[(blockIdx.x * 512 + threadIdx.x) % jitter_n];
        j = jitter[j % jitter_n];
        seed = (float)(j & 3);
    }
    commit(0, 0, 0); issue(STAGE, 0);
    barrier<F>();

    h8 x;
    for (int i = 0; i < 8; ++i) x[i] = (_Float16)(1.0f + seed * 0.0f);
    f4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    unsigned local_bad = 0, first_want = 0, first_got = 0;
    constexpr int UNROLL = 6 * (NSLOT == 3 ? 1 : 1);                     // lcm(2, 3) = lcm(3, 3) * 2 = 6: register and slot numbers are static
    for (int S0 = 0; S0 < nsuper; S0 += UNROLL) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const int S = S0 + u;
            if (S < nsuper) {
                const uint4 *slot = ring + (u % NSLOT) * SLOT_U4 + lane;
#pragma unroll
                for (int step = 0; step < 2; ++step) {
                    uint4 w[2];
#pragma unroll
                    for (int t = 0; t < 2; ++t) w[t] = slot[step * 256 + t * 128];      // chunks step * 4 + 2t
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        const unsigned want = ((unsigned)S << 16) | ((unsigned)(step * 4 + 2 * t) << 8) | (unsigned)lane;
                        if (w[t].x != want) { if (!local_bad) { first_want = want; first_got = w[t].x; } ++local_bad; }
                        if constexpr (!(F & NO_MFMA)) {
                            const uint4 wv{0x3c003c00u, w[t].y, w[t].z, w[t].w};      // (word 0 is the label, not an operand)
                            acc[step * 2 + t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, wv), x, acc[step * 2 + t], 0, 0, 0);
                            acc[step * 2 + t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, wv), x, acc[step * 2 + t], 0, 0, 0);
                        }
                    }
                }
                // end of superstep S: S + 1 into the slot S - 1 used, S + 1 + STAGE requested
                commit(S + 1, (u + 1) % STAGE, (u + 1) % NSLOT);
                issue(S + 1 + STAGE, (u + 1) % STAGE);
                barrier<F>();
            }
        }
    }
    float s = seed;
    for (int c = 0; c < 4; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    sink[(size_t)blockIdx.x * 512 + threadIdx.x] = s;
    if (local_bad) {
        const unsigned at = atomicAdd(nerr, 1u);
        if (at < 4096) err[at] = ErrRec{blockIdx.x, (unsigned)wave | (local_bad << 8), first_want, first_got};
    }
}

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct Variant { unsigned flags; const char *name; };

template <unsigned F>
static void run_variant(const char *name, const uint4 *dW, int nsuper, int wgs, int runs, float *dsink, ErrRec *derr, unsigned *dnerr,
                        const int *djit, int jn)
{
    const size_t lds = ((F & THREE_SLOTS) ? 3 : 2) * SLOT_U4 * 16 + 21376;      // the kernels' 37 KB: two workgroups per CU by registers
    unsigned bad_runs = 0, total = 0;
    std::vector<ErrRec> first;
    for (int r = 0; r < runs; ++r) {
        CHECK(hipMemset(dnerr, 0, 4));
        hipLaunchKernelGGL(ring_kernel<F>, dim3(wgs), dim3(512), lds, 0, dW, nsuper, dsink, derr, dnerr, djit, jn);
        CHECK(hipDeviceSynchronize());
        unsigned n = 0;
        CHECK(hipMemcpy(&n, dnerr, 4, hipMemcpyDeviceToHost));
        if (n) {
            ++bad_runs; total += n;
            if (first.empty()) { first.resize(n < 4096 ? n : 4096); CHECK(hipMemcpy(first.data(), derr, first.size() * sizeof(ErrRec), hipMemcpyDeviceToHost)); }
        }
    }
    printf("%-44s flags %4u  wgs %5d  supersteps %3d  runs with a misread: %d / %d  (wave-lanes misreading: %u)\n", name, F, wgs, nsuper, bad_runs, runs, total);
    if (!first.empty()) {
        unsigned by_wave[8] = {}, stale = 0, early = 0, other = 0;
        for (auto &e : first) {
            by_wave[e.wave & 7]++;
            const unsigned ws = e.want >> 16, gs = e.got >> 16;
            if ((e.got & 0xffff) == (e.want & 0xffff)) { if (gs < ws) ++stale; else ++early; } else ++other;
        }
        printf("    first bad run: %zu lanes; by wave:", first.size());
        for (int w = 0; w < 8; ++w) printf(" %u", by_wave[w]);
        printf("; read an OLDER superstep %u, a NEWER one %u, something else %u\n", stale, early, other);
        for (size_t i = 0; i < first.size() && i < 6; ++i)
            printf("    wg %u wave %u (%u items): wanted S=%u chunk %u lane %u, read 0x%08x (S=%u chunk %u lane %u)\n", first[i].wg, first[i].wave & 7,
                   first[i].wave >> 8, first[i].want >> 16, (first[i].want >> 8) & 255, first[i].want & 255, first[i].got, first[i].got >> 16,
                   (first[i].got >> 8) & 255, first[i].got & 255);
    }
    fflush(stdout);
}

int main(int argc, char **argv)
{
    const int wgs = argc > 1 ? atoi(argv[1]) : 1372, nsuper = argc > 2 ? atoi(argv[2]) : 24, runs = argc > 3 ? atoi(argv[3]) : 40;
    std::vector<uint4> hW((size_t)nsuper * SLOT_U4);
    for (int S = 0; S < nsuper; ++S)
        for (int c = 0; c < 8; ++c)
            for (int l = 0; l < 64; ++l) {
                const unsigned h = 0x3c003c00u;                               // halves 1.0, 1.0
                hW[(size_t)S * SLOT_U4 + c * 64 + l] = uint4{((unsigned)S << 16) | ((unsigned)c << 8) | (unsigned)l, h, h, h};
            }
    const int jn = 1 << 20;
    std::vector<int> hj(jn);
    srand(1);
    for (auto &v : hj) v = rand() % jn;
    uint4 *dW; float *dsink; ErrRec *derr; unsigned *dnerr; int *djit;
    CHECK(hipMalloc(&dW, hW.size() * 16)); CHECK(hipMemcpy(dW, hW.data(), hW.size() * 16, hipMemcpyHostToDevice));
    CHECK(hipMalloc(&dsink, (size_t)wgs * 512 * 4)); CHECK(hipMalloc(&derr, 4096 * sizeof(ErrRec))); CHECK(hipMalloc(&dnerr, 4));
    CHECK(hipMalloc(&djit, jn * 4)); CHECK(hipMemcpy(djit, hj.data(), jn * 4, hipMemcpyHostToDevice));
#define RUN(F) run_variant<(F)>(#F, dW, nsuper, wgs, runs, dsink, derr, dnerr, djit, jn)
    RUN(0u);                                   // every wave loads (the shipped form)
    RUN(IDLE_SKIP);                            // the failing form
    RUN(IDLE_SKIP | FULL_BARRIER);
    RUN(IDLE_SKIP | BARRIER_BUILTIN);
    RUN(IDLE_SKIP | DOUBLE_BARRIER);
    RUN(IDLE_SKIP | THREE_SLOTS);
    RUN(IDLE_SKIP | IDLE_OTHER_LOAD);
    RUN(IDLE_SKIP | IDLE_WRITE);
    RUN(IDLE_SKIP | IDLE_SLEEP);
    RUN(IDLE_SKIP | PRIO);
    RUN(IDLE_SKIP | HIGH_IDLE);
    RUN(IDLE_SKIP | NO_MFMA);
    RUN(IDLE_SKIP | NO_JITTER);
    RUN(IDLE_SKIP | NO_JITTER | NO_MFMA);
    return 0;
}
