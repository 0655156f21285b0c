// Shared device-side definitions of libzbra_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "zbra_hip.h"

namespace zbra {

// Values of one tile (one workgroup iteration).  A multiple of 64 so that a tile of any width w
// starts on a byte boundary of the packed stream: TILE*w bits = 256*w bytes.
constexpr int TILE = 2048;
constexpr int THREADS = 256;
constexpr uint32_t MAX_COUNT = 0xFFFFFFFFu;  // `values.len() as u32`, binary.rs:640

// Internal element kind of the stage functions bp64_pack / bp64_unpack: u64 values that are
// packed / unpacked as they are (no frame of reference, no zig-zag).
constexpr uint32_t KIND_U64_PLAIN = 2;

typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint64_t u64x2 __attribute__((ext_vector_type(2)));

// Pointers that reach a kernel through a descriptor in memory are generic to the compiler, which
// then emits flat_* instructions; every buffer of this library is global memory, so say so.
#define ZBRA_GLOBAL __attribute__((address_space(1)))
template <typename T>
__device__ __forceinline__ const ZBRA_GLOBAL T *gptr(uintptr_t p) { return (const ZBRA_GLOBAL T *)p; }
template <typename T>
__device__ __forceinline__ const ZBRA_GLOBAL T *gptr(const void *p) { return (const ZBRA_GLOBAL T *)p; }
template <typename T>
__device__ __forceinline__ ZBRA_GLOBAL T *gptr_mut(uintptr_t p) { return (ZBRA_GLOBAL T *)p; }
template <typename T>
__device__ __forceinline__ ZBRA_GLOBAL T *gptr_mut(void *p) { return (ZBRA_GLOBAL T *)p; }

// Decode plan of one array, produced on the device from the array's own header.
struct DecodePlan {
    const uint8_t *payload;  // first byte after the width byte (E + 13)
    void *out;
    uint64_t count;          // values to write; 0 = nothing to do (empty, or rejected)
    int64_t midpoint;
    uint64_t avail;          // payload bytes present: packed_len - 1
    uint32_t width;          // width byte as stored, 0..255
    uint32_t kind;           // zbra_elem_kind
};

__device__ __forceinline__ uint64_t load_le64_bytes(const uint8_t *p) {
    uint64_t v = 0;
#pragma unroll
    for (int b = 0; b < 8; b++) v |= (uint64_t)p[b] << (8 * b);
    return v;
}
__device__ __forceinline__ uint32_t load_le32_bytes(const uint8_t *p) {
    uint32_t v = 0;
#pragma unroll
    for (int b = 0; b < 4; b++) v |= (uint32_t)p[b] << (8 * b);
    return v;
}

// zig_zag_decode, compression.rs:101-111
__device__ __forceinline__ int64_t zigzag_decode64(uint64_t n) {
    return (int64_t)((n >> 1) ^ (0 - (n & 1)));
}
// zig_zag_encode, compression.rs:88-98
__device__ __forceinline__ uint64_t zigzag_encode64(int64_t n) {
    return ((uint64_t)n << 1) ^ (uint64_t)(n >> 63);
}

// The 16 bytes that start 4*Q + r bytes into the 32 bytes a[0..8) (r = 0..3), as four dwords.
template <int Q>
__device__ __forceinline__ void funnel_bytes(const uint32_t (&a)[8], uint32_t r, uint32_t (&o)[4]) {
#pragma unroll
    for (int k = 0; k < 4; k++) o[k] = __builtin_amdgcn_alignbyte(a[Q + k + 1], a[Q + k], r);
}

// First array whose tile range contains tile t: largest a with tile_start[a] <= t.
// tile_start has nd + 1 entries; every lane of the workgroup computes the same answer.
__device__ __forceinline__ uint32_t find_array(const uint64_t *__restrict__ tile_start, uint32_t nd, uint64_t t) {
    uint32_t lo = 0, hi = nd;  // invariant: tile_start[lo] <= t < tile_start[hi]
    while (hi - lo > 1) {
        uint32_t mid = (lo + hi) >> 1;
        if (tile_start[mid] <= t) lo = mid; else hi = mid;
    }
    return lo;
}

// Tile handled by workgroup b when there is one workgroup per tile.  Workgroups are dealt to
// the 8 XCDs round-robin (b % 8 labels the XCD), so this gives each XCD one contiguous eighth
// of the tile range: every L2 then streams one region of memory instead of every eighth tile
// (+9 % store bandwidth measured).  Bijective for any ntiles.  Affects speed only.
__device__ __forceinline__ uint64_t xcd_chunked_tile(uint64_t b, uint64_t ntiles) {
    const uint64_t q = ntiles >> 3, r = ntiles & 7, xcd = b & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
}

// The same idea at a finer grain: inside every run of 8 * G consecutive tiles XCD x takes tiles
// [x * G, x * G + G), so neighbouring tiles still meet in one L2 while all eight XCDs work in the
// same few megabytes of memory.  Tiles past the last whole run keep their index.
__device__ __forceinline__ uint64_t xcd_grouped_tile(uint64_t b, uint64_t ntiles, uint32_t G) {
    const uint64_t span = 8ull * G;
    if (b >= ntiles / span * span) return b;
    const uint64_t within = b % span;
    return b - within + (within & 7) * G + (within >> 3);
}

}  // namespace zbra
