// Elementwise stage kernels behind the stage-level entry points: the reference exposes each
// stage of the pipeline as a public function (zbra-core/src/compression.rs:52-111) and its tests
// and benches call them one by one, so each has a kernel of its own.  Each lane moves 16 bytes.
#pragma once
#include "zbra_common.hpp"

namespace zbra {

template <typename F>
__device__ __forceinline__ void map_pairs(const uint64_t *__restrict__ in, uint64_t n, uint64_t *__restrict__ out, F &&f) {
    const uint64_t npairs = (n + 1) / 2;
    for (uint64_t pi = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; pi < npairs; pi += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t i = pi * 2;
        if (i + 1 < n) {
            const u64x2 v = *(const u64x2 *)(in + i);
            u64x2 r = {f(v.x), f(v.y)};
            *(u64x2 *)(out + i) = r;
        } else {
            out[i] = f(in[i]);
        }
    }
}

// frame_of_reference_encode's subtraction, compression.rs:71; the midpoint is read from the
// result words the median selection left on the device.
__global__ __launch_bounds__(THREADS) void for_sub_kernel(const int64_t *__restrict__ in, uint64_t n,
                                                           const zbra_dev_result *__restrict__ res, int64_t *__restrict__ out) {
    const uint64_t mid = (uint64_t)res->midpoint;
    map_pairs((const uint64_t *)in, n, (uint64_t *)out, [mid](uint64_t v) { return v - mid; });
}

// frame_of_reference_decode, compression.rs:77-82
__global__ __launch_bounds__(THREADS) void for_add_kernel(const int64_t *__restrict__ in, uint64_t n, int64_t midpoint,
                                                           int64_t *__restrict__ out) {
    const uint64_t mid = (uint64_t)midpoint;
    map_pairs((const uint64_t *)in, n, (uint64_t *)out, [mid](uint64_t v) { return v + mid; });
}

// zig_zag_encode, compression.rs:88-98
__global__ __launch_bounds__(THREADS) void zigzag_encode_kernel(const int64_t *__restrict__ in, uint64_t n,
                                                                 uint64_t *__restrict__ out) {
    map_pairs((const uint64_t *)in, n, out, [](uint64_t v) { return zigzag_encode64((int64_t)v); });
}

// zig_zag_decode, compression.rs:101-111
__global__ __launch_bounds__(THREADS) void zigzag_decode_kernel(const uint64_t *__restrict__ in, uint64_t n,
                                                                 int64_t *__restrict__ out) {
    map_pairs(in, n, (uint64_t *)out, [](uint64_t v) { return (uint64_t)zigzag_decode64(v); });
}

}  // namespace zbra
