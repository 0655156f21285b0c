// Write-only bandwidth of 16-byte-per-lane vector stores under each cache-policy modifier of the
// gfx950 global store (none, nt, sc0, sc1 and their combinations), for the decode kernel's shape
// (512 threads x 2 stores = 16 KiB per workgroup) and a 4-KiB shape.  Tuning aid, not product code.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int POLICY>
__device__ __forceinline__ void store16(u32x4 *p, u32x4 v) {
    if constexpr (POLICY == 0) asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(p), "v"(v) : "memory");
    else if constexpr (POLICY == 1) asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(p), "v"(v) : "memory");
    else if constexpr (POLICY == 2) asm volatile("global_store_dwordx4 %0, %1, off sc0" ::"v"(p), "v"(v) : "memory");
    else if constexpr (POLICY == 3) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
    else if constexpr (POLICY == 4) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
    else if constexpr (POLICY == 5) asm volatile("global_store_dwordx4 %0, %1, off sc0 nt" ::"v"(p), "v"(v) : "memory");
    else if constexpr (POLICY == 6) asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" ::"v"(p), "v"(v) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" ::"v"(p), "v"(v) : "memory");
}

template <int BLOCK, int ITS, int POLICY>
__global__ __launch_bounds__(BLOCK) void fill_k(u32x4 *__restrict__ out, unsigned x) {
    u32x4 *base = out + (size_t)blockIdx.x * BLOCK * ITS;
#pragma unroll
    for (int it = 0; it < ITS; it++) {
        const unsigned slot = it * BLOCK + threadIdx.x;
        u32x4 v = {x, slot, x, x};
        store16<POLICY>(base + slot, v);
    }
}

template <typename F> float time_ms(F &&f, int reps) {
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    for (int i = 0; i < 3; i++) f();
    std::vector<float> ts;
    for (int r = 0; r < reps; r++) { CHECK(hipEventRecord(a)); f(); CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b)); float ms; CHECK(hipEventElapsedTime(&ms, a, b)); ts.push_back(ms); }
    std::sort(ts.begin(), ts.end());
    return ts[ts.size() / 2];
}

static void *out;
static const size_t OUT = (size_t)2 << 30;
static const char *NAMES[8] = {"plain", "nt", "sc0", "sc1", "sc0 sc1", "sc0 nt", "sc1 nt", "sc0 sc1 nt"};

template <int BLOCK, int ITS, int POLICY> void run() {
    const size_t per = (size_t)BLOCK * ITS * 16;
    const unsigned grid = (unsigned)(OUT / per);
    float t = time_ms([&] { hipLaunchKernelGGL((fill_k<BLOCK, ITS, POLICY>), dim3(grid), dim3(BLOCK), 0, 0, (u32x4 *)out, 7u); }, 21);
    printf("block %4d its %2d (%6zu B/wg) %-12s : %.4f ms  %.0f GB/s\n", BLOCK, ITS, per, NAMES[POLICY], t, OUT / t / 1e6);
}
template <int BLOCK, int ITS> void run8() {
    run<BLOCK, ITS, 0>(); run<BLOCK, ITS, 1>(); run<BLOCK, ITS, 2>(); run<BLOCK, ITS, 3>();
    run<BLOCK, ITS, 4>(); run<BLOCK, ITS, 5>(); run<BLOCK, ITS, 6>(); run<BLOCK, ITS, 7>();
}

int main() {
    CHECK(hipMalloc(&out, OUT)); CHECK(hipMemset(out, 2, OUT));
    run8<512, 2>(); run8<256, 1>(); run8<256, 4>();
    return 0;
}
