// Write-bandwidth sweep: one workgroup per contiguous chunk, BLOCK threads, ITS 16-B stores per
// thread; store slot = it*BLOCK + tid (a wave writes 1 KiB contiguous per instruction).
// Tuning aid for the decode kernel's store pattern, not product code.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int BLOCK, int ITS, bool NT, bool WAVE_CONTIG>
__global__ __launch_bounds__(BLOCK) void fill_k(u32x4* __restrict__ out, unsigned x) {
    u32x4* base = out + (size_t)blockIdx.x * BLOCK * ITS;
#pragma unroll
    for (int it = 0; it < ITS; it++) {
        unsigned slot = WAVE_CONTIG ? ((threadIdx.x >> 6) * 64 * ITS + it * 64 + (threadIdx.x & 63)) : (it * BLOCK + threadIdx.x);
        u32x4 v = {x, slot, x, x};
        if (NT) __builtin_nontemporal_store(v, base + slot); else base[slot] = v;
    }
}

template <typename F> float time_ms(F&& f, int reps) {
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    for (int i = 0; i < 3; i++) f();
    std::vector<float> ts;
    for (int r = 0; r < reps; r++) { CHECK(hipEventRecord(a)); f(); CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b)); float ms; CHECK(hipEventElapsedTime(&ms, a, b)); ts.push_back(ms); }
    std::sort(ts.begin(), ts.end());
    return ts[ts.size() / 2];
}

static void* out;
static const size_t OUT = (size_t)2 << 30;

template <int BLOCK, int ITS, bool NT, bool WC> void run() {
    const size_t per = (size_t)BLOCK * ITS * 16;
    const unsigned grid = (unsigned)(OUT / per);
    float t = time_ms([&] { hipLaunchKernelGGL((fill_k<BLOCK, ITS, NT, WC>), dim3(grid), dim3(BLOCK), 0, 0, (u32x4*)out, 7u); }, 21);
    printf("block %4d its %2d (%6zu B/wg, grid %7u) nt=%d wave_contig=%d : %.4f ms  %.0f GB/s\n", BLOCK, ITS, per, grid, (int)NT, (int)WC, t, OUT / t / 1e6);
}
template <int BLOCK, int ITS> void run4() { run<BLOCK, ITS, false, false>(); run<BLOCK, ITS, true, false>(); run<BLOCK, ITS, false, true>(); run<BLOCK, ITS, true, true>(); }

int main() {
    CHECK(hipMalloc(&out, OUT)); CHECK(hipMemset(out, 2, OUT));
    run4<64, 4>(); run4<64, 16>(); run4<128, 4>(); run4<128, 8>();
    run4<256, 1>(); run4<256, 2>(); run4<256, 4>(); run4<256, 8>(); run4<256, 16>();
    run4<512, 2>(); run4<512, 4>(); run4<512, 8>(); run4<1024, 2>(); run4<1024, 4>();
    return 0;
}
