// Does an XCD-aware block -> chunk mapping change streaming-store bandwidth?  Tuning aid.
// MODE 0: chunk = blockIdx (round-robin over XCDs).  MODE 1: XCD x (= blockIdx % 8) owns the
// contiguous eighth x of the buffer.  MODE 2: groups of 8 consecutive chunks per XCD visit.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int BLOCK, int ITS, bool NT, int MODE>
__global__ __launch_bounds__(BLOCK) void fill_k(u32x4* __restrict__ out, unsigned x) {
    unsigned b = blockIdx.x, n = gridDim.x;
    unsigned chunk = b;
    if (MODE == 1) chunk = (b & 7) * (n >> 3) + (b >> 3);
    if (MODE == 2) { unsigned g = b >> 6, r = b & 63; chunk = g * 64 + (r & 7) * 8 + (r >> 3); }
    u32x4* base = out + (size_t)chunk * BLOCK * ITS;
#pragma unroll
    for (int it = 0; it < ITS; it++) {
        unsigned slot = it * BLOCK + threadIdx.x;
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
template <int BLOCK, int ITS, bool NT, int MODE> void run() {
    const size_t per = (size_t)BLOCK * ITS * 16;
    const unsigned grid = (unsigned)(OUT / per);
    float t = time_ms([&] { hipLaunchKernelGGL((fill_k<BLOCK, ITS, NT, MODE>), dim3(grid), dim3(BLOCK), 0, 0, (u32x4*)out, 7u); }, 21);
    printf("block %4d its %2d nt=%d mode=%d : %.4f ms  %.0f GB/s\n", BLOCK, ITS, (int)NT, MODE, t, OUT / t / 1e6);
}
template <int BLOCK, int ITS> void run6() { run<BLOCK, ITS, false, 0>(); run<BLOCK, ITS, false, 1>(); run<BLOCK, ITS, false, 2>(); run<BLOCK, ITS, true, 0>(); run<BLOCK, ITS, true, 1>(); run<BLOCK, ITS, true, 2>(); }
int main() {
    CHECK(hipMalloc(&out, OUT)); CHECK(hipMemset(out, 2, OUT));
    run6<256, 1>(); run6<256, 4>(); run6<512, 2>(); run6<1024, 1>();
    return 0;
}
