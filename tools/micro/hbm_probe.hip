// hbm_probe.hip -- what a streaming kernel reaches on this GPU when its data is NOT in the 256 MB Infinity Cache: the ceiling the
// per-operator cost-volume kernels (softmax_pool_vec_kernel: read-dominated; cv_encode*: write-dominated) are compared with.
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/build/hbm_probe tools/micro/hbm_probe.hip && tools/micro/build/hbm_probe [MB per buffer]
// Every form runs over a RING of buffers (>= 2 x 256 MB of other buffers' traffic between two uses of one), 16-byte accesses:
//   read   one-shot (a thread loads U vectors, reduces, exits) / persistent (grid-stride, U loads in flight), plain or nontemporal
//   write  fill, plain or nontemporal stores
//   copy   read + write
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float v4 __attribute__((ext_vector_type(4)));

template <int U, bool NT, bool PERSIST>
__global__ __launch_bounds__(256) void read_kernel(const v4 *__restrict__ src, float *sink, long n)
{
    float acc = 0.f;
    const long stride = PERSIST ? (long)gridDim.x * 256 * U : 0;
    for (long base = (long)blockIdx.x * 256 * U + threadIdx.x; base < n; base += stride) {
        v4 x[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long i = base + (long)u * 256;
            if (i < n) x[u] = NT ? __builtin_nontemporal_load(src + i) : src[i];
            else x[u] = v4{0, 0, 0, 0};
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc += x[u][0] + x[u][1] + x[u][2] + x[u][3];
        if (!PERSIST) break;
    }
    if (acc == 1.2345e30f) *sink = acc;
}

template <bool NT>
__global__ __launch_bounds__(256) void write_kernel(v4 *__restrict__ dst, long n)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const v4 v{1.f, 2.f, 3.f, (float)i};
    if (NT) __builtin_nontemporal_store(v, dst + i);
    else dst[i] = v;
}

template <int U, bool NT, bool PERSIST>
__global__ __launch_bounds__(256) void copy_kernel(const v4 *__restrict__ src, v4 *__restrict__ dst, long n)
{
    const long stride = PERSIST ? (long)gridDim.x * 256 * U : 0;
    for (long base = (long)blockIdx.x * 256 * U + threadIdx.x; base < n; base += stride) {
        v4 x[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long i = base + (long)u * 256;
            if (i < n) x[u] = NT ? __builtin_nontemporal_load(src + i) : src[i];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long i = base + (long)u * 256;
            if (i < n) { if (NT) __builtin_nontemporal_store(x[u], dst + i); else dst[i] = x[u]; }
        }
        if (!PERSIST) break;
    }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <class F>
static double timed(F launch, int ring, int passes)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int r = 0; r < ring; ++r) launch(r);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a, 0));
    for (int p = 0; p < passes; ++p)
        for (int r = 0; r < ring; ++r) launch(r);
    CK(hipEventRecord(b, 0));
    CK(hipDeviceSynchronize());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    return ms * 1e-3 / (ring * passes);
}

int main(int argc, char **argv)
{
    const long mb = argc > 1 ? atol(argv[1]) : 200;
    const long bytes = mb << 20, n = bytes / 16;
    const int ring = (int)((2l * 256 * 1024 * 1024 + bytes - 1) / bytes) + 1;
    std::vector<v4 *> src(ring), dst(ring);
    for (int r = 0; r < ring; ++r) {
        CK(hipMalloc(&src[r], bytes)); CK(hipMalloc(&dst[r], bytes));
        CK(hipMemset(src[r], 0, bytes)); CK(hipMemset(dst[r], 0, bytes));
    }
    float *sink;
    CK(hipMalloc(&sink, 4));
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("%s, %d CUs; %ld MB per buffer, ring of %d (cold); fraction of 8 TB/s in brackets\n", prop.name, cus, mb, ring);
    auto rep = [&](const char *what, double sec, double moved) { printf("  %-58s %7.2f us  %6.0f GB/s (%.3f)\n", what, sec * 1e6, moved / sec / 1e9, moved / sec / 8e12); fflush(stdout); };
    const int one1 = (int)((n + 255) / 256), one4 = (int)((n + 1023) / 1024), one8 = (int)((n + 2047) / 2048);
#define READ(U, NT, P, G, label) rep(label, timed([&](int r) { hipLaunchKernelGGL((read_kernel<U, NT, P>), dim3(G), dim3(256), 0, 0, src[r], sink, n); }, ring, 5), (double)bytes)
    READ(1, false, false, one1, "read  one-shot, 1 x 16 B per thread");
    READ(4, false, false, one4, "read  one-shot, 4 x 16 B per thread");
    READ(8, false, false, one8, "read  one-shot, 8 x 16 B per thread");
    READ(8, true, false, one8, "read  one-shot, 8 x 16 B per thread, nontemporal");
    READ(4, false, true, cus * 8, "read  persistent (8 WG/CU), 4 in flight");
    READ(8, false, true, cus * 8, "read  persistent (8 WG/CU), 8 in flight");
    READ(8, true, true, cus * 8, "read  persistent (8 WG/CU), 8 in flight, nontemporal");
    READ(8, false, true, cus * 4, "read  persistent (4 WG/CU), 8 in flight");
    READ(8, false, true, cus * 16, "read  persistent (16 WG/CU), 8 in flight");
    READ(16, true, true, cus * 8, "read  persistent (8 WG/CU), 16 in flight, nontemporal");
#define WRITE(NT, label) rep(label, timed([&](int r) { hipLaunchKernelGGL((write_kernel<NT>), dim3(one1), dim3(256), 0, 0, dst[r], n); }, ring, 5), (double)bytes)
    WRITE(false, "write one-shot 16 B per thread");
    WRITE(true, "write one-shot 16 B per thread, nontemporal");
#define COPY(U, NT, P, G, label) rep(label, timed([&](int r) { hipLaunchKernelGGL((copy_kernel<U, NT, P>), dim3(G), dim3(256), 0, 0, src[r], dst[r], n); }, ring, 5), 2.0 * bytes)
    COPY(1, false, false, one1, "copy  one-shot 1 x 16 B (read + write bytes)");
    COPY(4, false, false, one4, "copy  one-shot 4 x 16 B");
    COPY(4, true, false, one4, "copy  one-shot 4 x 16 B, nontemporal");
    COPY(8, false, true, cus * 8, "copy  persistent (8 WG/CU), 8 in flight");
    COPY(8, true, true, cus * 8, "copy  persistent (8 WG/CU), 8 in flight, nontemporal");
    // warm reference: the same read on ONE 64 MB buffer (Infinity-Cache resident)
    {
        const long n2 = (64l << 20) / 16;
        hipLaunchKernelGGL((read_kernel<8, false, true>), dim3(cus * 8), dim3(256), 0, 0, src[0], sink, n2);
        rep("read  persistent, 8 in flight, ONE 64 MB buffer (warm)", timed([&](int) { hipLaunchKernelGGL((read_kernel<8, false, true>), dim3(cus * 8), dim3(256), 0, 0, src[0], sink, n2); }, 1, 20), 64.0 * (1 << 20));
    }
    return 0;
}
