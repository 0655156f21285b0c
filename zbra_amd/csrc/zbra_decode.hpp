// Decode kernels: header parse (plan) + fused unpack -> un-zig-zag -> + midpoint.
//
// Replaces, per array, the three passes and three allocations of decompress_int_array
// (zbra-core/src/compression.rs:295-330 -> bp64_unpack :181-233, zig_zag_decode :101-111,
// frame_of_reference_decode :77-82) with one pass: packed bytes in, final values out.
//
// The width is constant per array, so value i sits at bit i*w of the payload: decode needs no
// scan inside an array.  A tile is TILE consecutive values of one array; tile t of a submission
// is mapped to (array, first value) through the exclusive scan `tile_start` of tiles per array.
#pragma once
#include "zbra_common.hpp"

namespace zbra {

// ---- plan: parse and validate the header of each array on the device ------------------------

__device__ __forceinline__ void plan_one(const zbra_array_desc &d, DecodePlan *plan, zbra_dev_result *res) {
    DecodePlan p;
    p.payload = nullptr; p.out = d.out; p.count = 0; p.midpoint = 0; p.avail = 0; p.width = 0; p.kind = d.kind;
    zbra_dev_result r;
    r.status = ZBRA_OK; r.width = 0; r.out_count = 0; r.midpoint = 0; r.packed_len = 0;
    if (d.len == 0) {
        // compression.rs:296-298: empty data -> empty result whatever count says
    } else if (d.len < 12) {
        r.status = ZBRA_ERR_SHORT_HEADER;                       // :300-304
    } else {
        const int64_t mid = (int64_t)load_le64_bytes(d.data);   // :307
        const uint64_t packed_len = load_le32_bytes(d.data + 8);// :310
        r.midpoint = mid; r.packed_len = packed_len;
        if (d.len < 12 + packed_len) {
            r.status = ZBRA_ERR_SHORT_PACKED;                   // :312-316
        } else if (packed_len == 0) {
            // bp64_unpack on an empty slice returns an empty Vec (:182-184)
        } else {
            const uint32_t w = d.data[12];                      // :186
            r.width = w;
            p.payload = d.data + 13; p.midpoint = mid; p.avail = packed_len - 1; p.width = w;
            if (d.count == 0) {
                // :187-189 with count == 0
            } else if (w == 0 || w >= 32) {
                p.count = d.count;                              // zeros (:187-189) or raw u64 (:195-207)
                r.out_count = d.count;
            } else if (p.avail < (d.count * w + 7) / 8) {
                r.status = ZBRA_ERR_TRUNCATED;                  // reference: u32 underflow at :229
            } else {
                p.count = d.count;
                r.out_count = d.count;
            }
        }
    }
    *plan = p;
    if (res) *res = r;
}

__global__ void plan_decode_batch_kernel(const zbra_array_desc *__restrict__ descs, uint32_t nd,
                                         DecodePlan *__restrict__ plans, zbra_dev_result *results) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nd) return;
    plan_one(descs[i], plans + i, results ? results + i : nullptr);
}

// Single array: descriptor arrives in the kernel arguments; also writes tile_start = {0, ntiles}.
__global__ void plan_decode_one_kernel(zbra_array_desc d, uint64_t ntiles, DecodePlan *plan, uint64_t *tile_start,
                                       zbra_dev_result *result) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    plan_one(d, plan, result);
    tile_start[0] = 0;
    tile_start[1] = ntiles;
}

// bp64_unpack alone (compression.rs:181-233): `packed` = width byte + payload, already validated
// by the host (non-empty, count > 0, long enough).
__global__ void plan_unpack_kernel(const uint8_t *packed, uint64_t packed_len, uint64_t count, void *out,
                                   uint64_t ntiles, DecodePlan *plan, uint64_t *tile_start) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    DecodePlan p;
    p.payload = packed + 1; p.out = out; p.count = count; p.midpoint = 0; p.avail = packed_len - 1;
    p.width = packed[0]; p.kind = KIND_U64_PLAIN;
    *plan = p;
    tile_start[0] = 0;
    tile_start[1] = ntiles;
}

// ---- tile decode ---------------------------------------------------------------------------

// LDS image of one tile's packed bytes: at most 15 B of align-down slack + TILE*31/8 B + the
// dword the extractor reads past its value.
constexpr int LDS_DWORDS = 2048;
static_assert(15 + TILE * 31 / 8 + 8 <= LDS_DWORDS * 4, "LDS image too small");

// Value whose lowest bit is bit b of the LDS image (w <= 31 bits, mask = 2^w - 1).
__device__ __forceinline__ uint32_t extract_bits(const uint32_t *lds, uint32_t b, uint32_t mask) {
    const uint32_t idx = b >> 5;
    return __builtin_amdgcn_alignbit(lds[idx + 1], lds[idx], b & 31u) & mask;
}
// 32-bit zig-zag decode; exact for inputs below 2^31 (w <= 31).
__device__ __forceinline__ int32_t zigzag_decode32(uint32_t z) { return (int32_t)(z >> 1) ^ -(int32_t)(z & 1u); }

template <int Q>
__device__ __forceinline__ void funnel_bytes(const uint32_t (&a)[8], uint32_t r, uint32_t (&o)[4]) {
#pragma unroll
    for (int k = 0; k < 4; k++) o[k] = __builtin_amdgcn_alignbyte(a[Q + k + 1], a[Q + k], r);
}

__device__ __forceinline__ void tile_packed(const DecodePlan &p, uint64_t v0, uint32_t nv, uint32_t *lds) {
    const uint32_t tid = threadIdx.x;
    const uint32_t w = p.width;
    // v0 is a multiple of 8, so the tile starts on a byte of the payload.
    const uintptr_t A = (uintptr_t)p.payload + ((v0 * w) >> 3);
    const uintptr_t A0 = A & ~(uintptr_t)15;
    const uint32_t base_bits = (uint32_t)(A - A0) * 8u;
    const uint32_t nbytes = (nv * w + 7u) >> 3;
    const uint32_t nchunks = ((uint32_t)(A - A0) + nbytes + 15u) >> 4;
    // Each 16-B aligned chunk holds at least one byte of this array's payload, so it lies inside
    // a mapped page even when the payload itself starts or ends off a 16-B boundary.
    const ZBRA_GLOBAL u32x4 *src = gptr<u32x4>(A0);
    u32x4 *dst = (u32x4 *)lds;
    for (uint32_t c = tid; c < nchunks; c += THREADS) dst[c] = __builtin_nontemporal_load(src + c);
    __syncthreads();

    const uint32_t mask = (1u << w) - 1u;
    if (p.kind != ZBRA_ELEM_U32) {
        ZBRA_GLOBAL int64_t *out = gptr_mut<int64_t>(p.out) + v0;
        const int64_t mid = p.midpoint;
        const bool plain = p.kind == KIND_U64_PLAIN;  // bp64_unpack alone: no zig-zag, no midpoint
#pragma unroll
        for (int it = 0; it < TILE / (THREADS * 2); ++it) {
            const uint32_t j = it * THREADS * 2 + tid * 2;
            if (j < nv) {
                const uint32_t b = base_bits + j * w;
                const uint32_t z0 = extract_bits(lds, b, mask), z1 = extract_bits(lds, b + w, mask);
                const int64_t o0 = plain ? (int64_t)z0 : mid + (int64_t)zigzag_decode32(z0);
                const int64_t o1 = plain ? (int64_t)z1 : mid + (int64_t)zigzag_decode32(z1);
                if (j + 1 < nv) {
                    u64x2 v = {(uint64_t)o0, (uint64_t)o1};
                    __builtin_nontemporal_store(v, (ZBRA_GLOBAL u64x2 *)(out + j));
                } else {
                    out[j] = o0;
                }
            }
        }
    } else {  // ZBRA_ELEM_U32: `as u32` keeps the low 32 bits of midpoint + delta (binary.rs:673)
        ZBRA_GLOBAL uint32_t *out = gptr_mut<uint32_t>(p.out) + v0;
        const uint32_t mid = (uint32_t)p.midpoint;
#pragma unroll
        for (int it = 0; it < TILE / (THREADS * 4); ++it) {
            const uint32_t j = it * THREADS * 4 + tid * 4;
            if (j < nv) {
                const uint32_t b = base_bits + j * w;
                u32x4 v;
                v.x = mid + (uint32_t)zigzag_decode32(extract_bits(lds, b, mask));
                v.y = mid + (uint32_t)zigzag_decode32(extract_bits(lds, b + w, mask));
                v.z = mid + (uint32_t)zigzag_decode32(extract_bits(lds, b + 2 * w, mask));
                v.w = mid + (uint32_t)zigzag_decode32(extract_bits(lds, b + 3 * w, mask));
                if (j + 3 < nv) {
                    __builtin_nontemporal_store(v, (ZBRA_GLOBAL u32x4 *)(out + j));
                } else {
                    out[j] = v.x;
                    if (j + 1 < nv) out[j + 1] = v.y;
                    if (j + 2 < nv) out[j + 2] = v.z;
                }
            }
        }
    }
    __syncthreads();  // the next tile refills the LDS image
}

// Raw fallback (width byte >= 32): value i is the u64 at payload + 8*i when all 8 bytes are
// present, else 0 (compression.rs:195-207).
__device__ __forceinline__ void tile_raw(const DecodePlan &p, uint64_t v0, uint32_t nv) {
    const uint32_t tid = threadIdx.x;
    const uintptr_t A = (uintptr_t)p.payload + v0 * 8;  // v0*8 is a multiple of 16
    const uintptr_t A0 = A & ~(uintptr_t)15;
    const uint32_t m = (uint32_t)(A - A0);
    const uint32_t q = m >> 2, r = m & 3u;
    const uintptr_t end = (uintptr_t)p.payload + p.avail;
    const int64_t mid = p.midpoint;
#pragma unroll
    for (int it = 0; it < TILE / (THREADS * 2); ++it) {
        const uint32_t j = it * THREADS * 2 + tid * 2;
        if (j < nv) {
            const uintptr_t c0 = A0 + (uintptr_t)j * 8;
            u32x4 x = {0, 0, 0, 0}, y = {0, 0, 0, 0};
            if (c0 < end) x = __builtin_nontemporal_load(gptr<u32x4>(c0));
            if (m != 0 && c0 + 16 < end) y = __builtin_nontemporal_load(gptr<u32x4>(c0 + 16));
            const uint32_t a[8] = {x.x, x.y, x.z, x.w, y.x, y.y, y.z, y.w};
            uint32_t o[4];
            switch (q) {
                case 0: funnel_bytes<0>(a, r, o); break;
                case 1: funnel_bytes<1>(a, r, o); break;
                case 2: funnel_bytes<2>(a, r, o); break;
                default: funnel_bytes<3>(a, r, o); break;
            }
            uint64_t u0 = ((uint64_t)o[1] << 32) | o[0];
            uint64_t u1 = ((uint64_t)o[3] << 32) | o[2];
            const uint64_t gi = v0 + j;
            if (gi * 8 + 8 > p.avail) u0 = 0;
            if (gi * 8 + 16 > p.avail) u1 = 0;
            const bool plain = p.kind == KIND_U64_PLAIN;
            const int64_t o0 = plain ? (int64_t)u0 : mid + zigzag_decode64(u0);
            const int64_t o1 = plain ? (int64_t)u1 : mid + zigzag_decode64(u1);
            if (p.kind != ZBRA_ELEM_U32) {
                ZBRA_GLOBAL int64_t *out = gptr_mut<int64_t>(p.out) + v0;
                if (j + 1 < nv) {
                    u64x2 v = {(uint64_t)o0, (uint64_t)o1};
                    __builtin_nontemporal_store(v, (ZBRA_GLOBAL u64x2 *)(out + j));
                } else {
                    out[j] = o0;
                }
            } else {
                ZBRA_GLOBAL uint32_t *out = gptr_mut<uint32_t>(p.out) + v0;
                out[j] = (uint32_t)o0;
                if (j + 1 < nv) out[j + 1] = (uint32_t)o1;
            }
        }
    }
}

// Width byte 0: every unpacked value is 0 (compression.rs:187-189), so every output is the midpoint.
__device__ __forceinline__ void tile_fill(const DecodePlan &p, uint64_t v0, uint32_t nv) {
    const uint32_t tid = threadIdx.x;
    for (uint32_t j = tid; j < nv; j += THREADS) {
        if (p.kind != ZBRA_ELEM_U32) gptr_mut<int64_t>(p.out)[v0 + j] = p.midpoint;
        else gptr_mut<uint32_t>(p.out)[v0 + j] = (uint32_t)p.midpoint;
    }
}

__global__ __launch_bounds__(THREADS) void decode_tiles_kernel(const DecodePlan *__restrict__ plans,
                                                                const uint64_t *__restrict__ tile_start, uint32_t nd,
                                                                uint64_t ntiles) {
    __shared__ __attribute__((aligned(16))) uint32_t lds[LDS_DWORDS];
    for (uint64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const uint32_t a = nd > 1 ? find_array(tile_start, nd, t) : 0;
        const DecodePlan p = plans[a];
        const uint64_t v0 = (t - tile_start[a]) * TILE;
        if (v0 >= p.count) continue;  // rejected or empty array: its tiles do nothing
        const uint64_t left = p.count - v0;
        const uint32_t nv = left < (uint64_t)TILE ? (uint32_t)left : (uint32_t)TILE;
        if (p.width - 1u < 31u) tile_packed(p, v0, nv, lds);
        else if (p.width == 0) tile_fill(p, v0, nv);
        else tile_raw(p, v0, nv);
    }
}

}  // namespace zbra
