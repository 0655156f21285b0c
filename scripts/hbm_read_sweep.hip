// Streaming-read bandwidth for the encoder's passes (min/max, histogram): 2 GiB read, nothing
// written.  MODE 0 grid-stride persistent, MODE 1 one chunk per workgroup, MODE 2 one chunk per
// workgroup in XCD-chunked order, MODE 3 persistent with a contiguous range per workgroup.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int BLOCK, int ITS, bool NT, int MODE>
__global__ __launch_bounds__(BLOCK) void read_k(const u32x4* __restrict__ in, size_t nchunks, unsigned* __restrict__ sink) {
    unsigned acc = 0;
    auto chunk = [&](size_t c) {
        const u32x4* base = in + c * BLOCK * ITS;
        u32x4 v[ITS];
#pragma unroll
        for (int it = 0; it < ITS; it++) v[it] = NT ? __builtin_nontemporal_load(base + it * BLOCK + threadIdx.x) : base[it * BLOCK + threadIdx.x];
#pragma unroll
        for (int it = 0; it < ITS; it++) acc ^= v[it].x ^ v[it].y ^ v[it].z ^ v[it].w;
    };
    if (MODE == 0) { for (size_t c = blockIdx.x; c < nchunks; c += gridDim.x) chunk(c); }
    else if (MODE == 1) chunk(blockIdx.x);
    else if (MODE == 2) { size_t b = blockIdx.x, q = nchunks >> 3; chunk((b & 7) * q + (b >> 3)); }
    else { size_t per = nchunks / gridDim.x; for (size_t c = blockIdx.x * per; c < (blockIdx.x + 1) * per; c++) chunk(c); }
    if (acc == 0x12345678u) sink[0] = acc;
}
template <typename F> float time_ms(F&& f, int reps) {
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    for (int i = 0; i < 3; i++) f();
    std::vector<float> ts;
    for (int r = 0; r < reps; r++) { CHECK(hipEventRecord(a)); f(); CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b)); float ms; CHECK(hipEventElapsedTime(&ms, a, b)); ts.push_back(ms); }
    std::sort(ts.begin(), ts.end());
    return ts[ts.size() / 2];
}
static void *in, *sink;
static const size_t N = (size_t)2 << 30;
template <int BLOCK, int ITS, bool NT, int MODE> void run(unsigned persistent_grid) {
    const size_t nchunks = N / ((size_t)BLOCK * ITS * 16);
    const unsigned grid = (MODE == 0 || MODE == 3) ? persistent_grid : (unsigned)nchunks;
    float t = time_ms([&] { hipLaunchKernelGGL((read_k<BLOCK, ITS, NT, MODE>), dim3(grid), dim3(BLOCK), 0, 0, (const u32x4*)in, nchunks, (unsigned*)sink); }, 15);
    printf("block %4d its %2d nt=%d mode=%d grid %7u : %.4f ms  %.0f GB/s\n", BLOCK, ITS, (int)NT, MODE, grid, t, N / t / 1e6);
}
template <int BLOCK, int ITS, bool NT> void runall() {
    run<BLOCK, ITS, NT, 0>(2048); run<BLOCK, ITS, NT, 0>(8192); run<BLOCK, ITS, NT, 1>(0); run<BLOCK, ITS, NT, 2>(0); run<BLOCK, ITS, NT, 3>(2048);
}
int main() {
    CHECK(hipMalloc(&in, N)); CHECK(hipMalloc(&sink, 64)); CHECK(hipMemset(in, 1, N));
    runall<256, 4, false>(); runall<256, 4, true>(); runall<256, 8, false>(); runall<256, 8, true>();
    runall<512, 4, false>(); runall<256, 2, false>(); runall<256, 16, false>();
    return 0;
}
