// Measures what the HBM of this GPU sustains for the decode kernel's traffic mix, with kernels
// that do no arithmetic: (a) 1:1 copy, (b) write-only, (c) read R bytes + write 16 B per lane
// with R/16 chosen like a width-12 decode (1.5 B in per 8 B out).  Tuning aid, not product code.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <bool NT>
__global__ __launch_bounds__(256) void copy_k(const u32x4* __restrict__ in, u32x4* __restrict__ out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        u32x4 v = NT ? __builtin_nontemporal_load(in + i) : in[i];
        if (NT) __builtin_nontemporal_store(v, out + i); else out[i] = v;
    }
}
template <bool NT>
__global__ __launch_bounds__(256) void fill_k(u32x4* __restrict__ out, size_t n, unsigned x) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        u32x4 v = {x, (unsigned)i, x, x};
        if (NT) __builtin_nontemporal_store(v, out + i); else out[i] = v;
    }
}
// each lane: reads 3 dwords per 16-B store -> 12 B in : 16 B out?  no: mimic w=12: 2 values (24 bits = 3 B) per 16 B out.
// A workgroup tile of 2048 values reads 3072 B (192 x 16 B) and writes 16384 B.
template <bool NT>
__global__ __launch_bounds__(256) void expand_k(const u32x4* __restrict__ in, u32x4* __restrict__ out, size_t ntiles) {
    __shared__ u32x4 lds[192];
    for (size_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        if (threadIdx.x < 192) lds[threadIdx.x] = NT ? __builtin_nontemporal_load(in + t * 192 + threadIdx.x) : in[t * 192 + threadIdx.x];
        __syncthreads();
        const unsigned* l = (const unsigned*)lds;
#pragma unroll
        for (int it = 0; it < 4; it++) {
            unsigned j = it * 256 + threadIdx.x;       // output 16-B slot within tile
            unsigned a = l[(j * 3) >> 2], b = l[((j * 3) >> 2) + 1];
            u32x4 v = {a, b, a ^ b, j};
            if (NT) __builtin_nontemporal_store(v, out + t * 1024 + j); else out[t * 1024 + j] = v;
        }
        __syncthreads();
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

int main() {
    const size_t OUT = (size_t)2 << 30, IN = OUT * 3 / 16;
    void *in, *out;
    CHECK(hipMalloc(&in, OUT)); CHECK(hipMalloc(&out, OUT));
    CHECK(hipMemset(in, 1, OUT)); CHECK(hipMemset(out, 2, OUT));
    for (int grid : {1024, 2048, 4096, 8192, 65536}) {
        float c0 = time_ms([&] { hipLaunchKernelGGL(copy_k<false>, dim3(grid), dim3(256), 0, 0, (const u32x4*)in, (u32x4*)out, OUT / 16); }, 15);
        float c1 = time_ms([&] { hipLaunchKernelGGL(copy_k<true>, dim3(grid), dim3(256), 0, 0, (const u32x4*)in, (u32x4*)out, OUT / 16); }, 15);
        float f0 = time_ms([&] { hipLaunchKernelGGL(fill_k<false>, dim3(grid), dim3(256), 0, 0, (u32x4*)out, OUT / 16, 7u); }, 15);
        float f1 = time_ms([&] { hipLaunchKernelGGL(fill_k<true>, dim3(grid), dim3(256), 0, 0, (u32x4*)out, OUT / 16, 7u); }, 15);
        float e0 = time_ms([&] { hipLaunchKernelGGL(expand_k<false>, dim3(grid), dim3(256), 0, 0, (const u32x4*)in, (u32x4*)out, OUT / 16384); }, 15);
        float e1 = time_ms([&] { hipLaunchKernelGGL(expand_k<true>, dim3(grid), dim3(256), 0, 0, (const u32x4*)in, (u32x4*)out, OUT / 16384); }, 15);
        printf("grid %6d | copy 2+2 GiB: plain %.3f ms %.0f GB/s, nt %.3f ms %.0f GB/s | fill 2 GiB: plain %.3f ms %.0f GB/s, nt %.3f ms %.0f GB/s | expand 0.375+2 GiB: plain %.3f ms %.0f GB/s, nt %.3f ms %.0f GB/s\n",
               grid, c0, 2.0 * OUT / c0 / 1e6, c1, 2.0 * OUT / c1 / 1e6, f0, OUT / f0 / 1e6, f1, OUT / f1 / 1e6,
               e0, (double)(OUT + IN) / e0 / 1e6, e1, (double)(OUT + IN) / e1 / 1e6);
    }
    return 0;
}
