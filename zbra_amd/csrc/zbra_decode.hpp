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

// One workgroup per array: lane 0 parses the header, then the workgroup writes the array's index
// into its slice of the tile -> array table, so that a decode workgroup finds its array with one
// load instead of a binary search over tile_start (ten dependent loads for a thousand arrays).
// `orig` (may be null): the descriptors were handed over in another order than the caller's
// (narrow arrays first, see the host); orig[a] is where the caller expects array a's result words.
__global__ __launch_bounds__(256) void plan_decode_batch_kernel(const zbra_array_desc *__restrict__ descs, uint32_t nd,
                                                                const uint64_t *__restrict__ tile_start,
                                                                DecodePlan *__restrict__ plans, uint32_t *__restrict__ tile_map,
                                                                zbra_dev_result *results, const uint32_t *__restrict__ orig) {
    const uint32_t a = blockIdx.x;
    if (a >= nd) return;
    if (threadIdx.x == 0) plan_one(descs[a], plans + a, results ? results + (orig ? orig[a] : a) : nullptr);
    const uint64_t t0 = tile_start[a], t1 = tile_start[a + 1];
    for (uint64_t t = t0 + threadIdx.x; t < t1; t += blockDim.x) tile_map[t] = a;
}

// Single array: descriptor arrives in the kernel arguments; also writes tile_start = {0, ntiles}.
// (Parsing the header inside the tile kernel instead, to save this launch, was measured 17 %
// slower on the 2^28-row column: 0.497 ms against 0.427 ms.)
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
//
// Template parameters: DT = values per tile (a multiple of 4*TH, so that TH lanes x 2 (or 4)
// values fill whole iterations, and of 8, so that a tile of any width starts on a byte of the
// stream), TH = threads per workgroup, NT = which accesses are nontemporal (bit 0 loads, bit 1 stores)
// loads/stores, DIRECT = packed bits are read straight from global memory with dword-aligned
// loads instead of being staged through LDS.

// NT: bit 0 = nontemporal loads, bit 1 = nontemporal stores.
template <int NT, typename T>
__device__ __forceinline__ T ld_stream(const ZBRA_GLOBAL T *p) {
    if constexpr (NT & 1) return __builtin_nontemporal_load(p);
    else return *p;
}
template <int NT, typename T>
__device__ __forceinline__ void st_stream(T v, ZBRA_GLOBAL T *p) {
    if constexpr (NT & 2) __builtin_nontemporal_store(v, p);
    else *p = v;
}

// LDS image of one tile's packed bytes: at most 15 B of align-down slack + DT*31/8 B + the
// dword the extractor reads past its value; rounded up to 16-B chunks.
template <int DT>
constexpr int lds_dwords() { return ((15 + DT * 31 / 8 + 8 + 15) / 16) * 4; }

// Value whose lowest bit is bit b of the LDS image (w <= 31 bits, mask = 2^w - 1).
__device__ __forceinline__ uint32_t extract_bits(const uint32_t *lds, uint32_t b, uint32_t mask) {
    const uint32_t idx = b >> 5;
    return __builtin_amdgcn_alignbit(lds[idx + 1], lds[idx], b & 31u) & mask;
}
// 32-bit zig-zag decode; exact for inputs below 2^31 (w <= 31).
__device__ __forceinline__ int32_t zigzag_decode32(uint32_t z) { return (int32_t)(z >> 1) ^ -(int32_t)(z & 1u); }

template <int NT>
__device__ __forceinline__ void store_pair_i64(ZBRA_GLOBAL int64_t *out, uint32_t j, uint32_t nv, int64_t o0, int64_t o1) {
    if (j + 1 < nv) {
        u64x2 v = {(uint64_t)o0, (uint64_t)o1};
        st_stream<NT>(v, (ZBRA_GLOBAL u64x2 *)(out + j));
    } else {
        out[j] = o0;
    }
}

// Where the packed bytes of one tile are: 16-B aligned chunk range and the bit offset of the
// tile's first value inside it.
struct PackedSpan {
    uintptr_t A0;        // 16-B aligned address of the first chunk
    uint32_t base_bits;  // bit offset of the tile's first value from A0
    uint32_t nchunks;    // 16-B chunks that hold the tile's bytes
};
__device__ __forceinline__ PackedSpan packed_span(const DecodePlan &p, uint64_t v0, uint32_t nv) {
    const uint32_t w = p.width;
    // v0 is a multiple of 8, so the tile starts on a byte of the payload.
    const uintptr_t A = (uintptr_t)p.payload + ((v0 * w) >> 3);
    PackedSpan s;
    s.A0 = A & ~(uintptr_t)15;
    s.base_bits = (uint32_t)(A - s.A0) * 8u;
    const uint32_t nbytes = (nv * w + 7u) >> 3;
    // Each 16-B aligned chunk holds at least one byte of this array's payload, so it lies inside
    // a mapped page even when the payload itself starts or ends off a 16-B boundary.
    s.nchunks = ((uint32_t)(A - s.A0) + nbytes + 15u) >> 4;
    return s;
}

// LDS image -> values -> 16-B stores, for a tile whose packed bytes are staged in `lds`.
template <int DT, int TH, int NT>
__device__ __forceinline__ void packed_extract(const DecodePlan &p, uint64_t v0, uint32_t nv, uint32_t base_bits,
                                               const uint32_t *lds) {
    const uint32_t tid = threadIdx.x;
    const uint32_t w = p.width;
    const uint32_t mask = (1u << w) - 1u;
    if (p.kind != ZBRA_ELEM_U32) {
        ZBRA_GLOBAL int64_t *out = gptr_mut<int64_t>(p.out) + v0;
        const int64_t mid = p.midpoint;
        const bool plain = p.kind == KIND_U64_PLAIN;  // bp64_unpack alone: no zig-zag, no midpoint
#pragma unroll
        for (int it = 0; it < DT / (TH * 2); ++it) {
            const uint32_t j = it * TH * 2 + tid * 2;
            if (j < nv) {
                const uint32_t b = base_bits + j * w;
                uint32_t z0, z1;
                if (w <= 16u) {  // uniform: both values lie inside the 32 bits that start at b
                    const uint32_t x = extract_bits(lds, b, 0xFFFFFFFFu);
                    z0 = x & mask; z1 = (x >> w) & mask;
                } else {
                    z0 = extract_bits(lds, b, mask); z1 = extract_bits(lds, b + w, mask);
                }
                const int64_t o0 = plain ? (int64_t)z0 : mid + (int64_t)zigzag_decode32(z0);
                const int64_t o1 = plain ? (int64_t)z1 : mid + (int64_t)zigzag_decode32(z1);
                store_pair_i64<NT>(out, j, nv, o0, o1);
            }
        }
    } else {  // ZBRA_ELEM_U32: `as u32` keeps the low 32 bits of midpoint + delta (binary.rs:673)
        ZBRA_GLOBAL uint32_t *out = gptr_mut<uint32_t>(p.out) + v0;
        const uint32_t mid = (uint32_t)p.midpoint;
        constexpr int ITS = DT / (TH * 4) > 0 ? DT / (TH * 4) : 1;
#pragma unroll
        for (int it = 0; it < ITS; ++it) {
            const uint32_t j = it * TH * 4 + tid * 4;
            if (j < nv) {
                const uint32_t b = base_bits + j * w;
                u32x4 v;
                if (w <= 8u) {  // uniform: the lane's four values lie inside the 32 bits that start at b
                    const uint32_t x = extract_bits(lds, b, 0xFFFFFFFFu);
                    v.x = mid + (uint32_t)zigzag_decode32(x & mask);
                    v.y = mid + (uint32_t)zigzag_decode32((x >> w) & mask);
                    v.z = mid + (uint32_t)zigzag_decode32((x >> (2 * w)) & mask);
                    v.w = mid + (uint32_t)zigzag_decode32((x >> (3 * w)) & mask);
                } else {
                    v.x = mid + (uint32_t)zigzag_decode32(extract_bits(lds, b, mask));
                    v.y = mid + (uint32_t)zigzag_decode32(extract_bits(lds, b + w, mask));
                    v.z = mid + (uint32_t)zigzag_decode32(extract_bits(lds, b + 2 * w, mask));
                    v.w = mid + (uint32_t)zigzag_decode32(extract_bits(lds, b + 3 * w, mask));
                }
                if (j + 3 < nv) {
                    st_stream<NT>(v, (ZBRA_GLOBAL u32x4 *)(out + j));
                } else {
                    out[j] = v.x;
                    if (j + 1 < nv) out[j + 1] = v.y;
                    if (j + 2 < nv) out[j + 2] = v.z;
                }
            }
        }
    }
}

template <int DT, int TH, int NT>
__device__ __forceinline__ void tile_packed(const DecodePlan &p, uint64_t v0, uint32_t nv, uint32_t *lds) {
    const PackedSpan sp = packed_span(p, v0, nv);
    const ZBRA_GLOBAL u32x4 *src = gptr<u32x4>(sp.A0);
    u32x4 *dst = (u32x4 *)lds;
    for (uint32_t c = threadIdx.x; c < sp.nchunks; c += TH) dst[c] = ld_stream<NT>(src + c);
    __syncthreads();
    packed_extract<DT, TH, NT>(p, v0, nv, sp.base_bits, lds);
    __syncthreads();  // a later tile of this workgroup refills the LDS image
}

// Packed path without LDS (i64 / plain outputs): each lane reads the three dwords that hold its
// two values.  Dwords past the last payload byte are never touched.
typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
typedef u32x3 u32x3_a4 __attribute__((aligned(4)));  // three dwords at any dword address

template <int DT, int TH, int NT>
__device__ __forceinline__ void tile_packed_direct(const DecodePlan &p, uint64_t v0, uint32_t nv) {
    const uint32_t tid = threadIdx.x;
    const uint32_t w = p.width;
    const uintptr_t P = (uintptr_t)p.payload;
    const uintptr_t A = P + ((v0 * w) >> 3);       // tile start, a byte of the payload
    const uintptr_t A4 = A & ~(uintptr_t)3;
    const uint32_t base_bits = (uint32_t)(A - A4) * 8u;
    // last dword (relative to A4) that holds a payload byte
    const uint64_t last_dword = ((P + p.avail - 1) - A4) >> 2;
    const uint32_t need_dwords = ((base_bits + nv * w) >> 5) + 2u;  // >= the highest dword index any lane loads
    const bool interior = (uint64_t)need_dwords <= last_dword;             // uniform
    const ZBRA_GLOBAL uint32_t *src = gptr<uint32_t>(A4);
    const uint32_t mask = (1u << w) - 1u;
    ZBRA_GLOBAL int64_t *out = gptr_mut<int64_t>(p.out) + v0;
    const int64_t mid = p.midpoint;
    const bool plain = p.kind == KIND_U64_PLAIN;
#pragma unroll
    for (int it = 0; it < DT / (TH * 2); ++it) {
        const uint32_t j = it * TH * 2 + tid * 2;
        if (j < nv) {
            const uint32_t b = base_bits + j * w;
            const uint32_t idx = b >> 5, s = b & 31u;
            uint32_t d0, d1, d2;
            if (interior) {
                const ZBRA_GLOBAL u32x3_a4 *q = (const ZBRA_GLOBAL u32x3_a4 *)(src + idx);
                u32x3 t;
                if constexpr (NT & 1) t = __builtin_nontemporal_load(q);
                else t = *q;
                d0 = t.x; d1 = t.y; d2 = t.z;
            } else {
                d0 = src[idx];
                d1 = (uint64_t)idx + 1 <= last_dword ? src[idx + 1] : 0u;
                d2 = (uint64_t)idx + 2 <= last_dword ? src[idx + 2] : 0u;
            }
            const uint32_t z0 = __builtin_amdgcn_alignbit(d1, d0, s) & mask;
            const uint32_t s1 = s + w;
            const bool up = s1 >= 32u;
            const uint32_t z1 = __builtin_amdgcn_alignbit(up ? d2 : d1, up ? d1 : d0, s1 & 31u) & mask;
            const int64_t o0 = plain ? (int64_t)z0 : mid + (int64_t)zigzag_decode32(z0);
            const int64_t o1 = plain ? (int64_t)z1 : mid + (int64_t)zigzag_decode32(z1);
            store_pair_i64<NT>(out, j, nv, o0, o1);
        }
    }
}

// The same for u32 outputs of widths up to 8 (enum tags): a lane's four values lie inside the 32
// bits that start at its first value, which two dwords hold.
typedef u32x2 u32x2_a4 __attribute__((aligned(4)));  // two dwords at any dword address
constexpr uint32_t DIRECT_U32_MAX_W = 8;

template <int DT, int TH, int NT>
__device__ __forceinline__ void tile_packed_direct_u32(const DecodePlan &p, uint64_t v0, uint32_t nv) {
    const uint32_t tid = threadIdx.x;
    const uint32_t w = p.width;
    const uintptr_t P = (uintptr_t)p.payload;
    const uintptr_t A = P + ((v0 * w) >> 3);       // tile start, a byte of the payload
    const uintptr_t A4 = A & ~(uintptr_t)3;
    const uint32_t base_bits = (uint32_t)(A - A4) * 8u;
    const uint64_t last_dword = ((P + p.avail - 1) - A4) >> 2;  // last dword (relative to A4) that holds a payload byte
    const uint32_t need_dwords = ((base_bits + nv * w) >> 5) + 1u;  // >= the highest dword index any lane loads
    const bool interior = (uint64_t)need_dwords <= last_dword;             // uniform
    const ZBRA_GLOBAL uint32_t *src = gptr<uint32_t>(A4);
    const uint32_t mask = (1u << w) - 1u;
    ZBRA_GLOBAL uint32_t *out = gptr_mut<uint32_t>(p.out) + v0;
    const uint32_t mid = (uint32_t)p.midpoint;
    constexpr int ITS = DT / (TH * 4) > 0 ? DT / (TH * 4) : 1;
#pragma unroll
    for (int it = 0; it < ITS; ++it) {
        const uint32_t j = it * TH * 4 + tid * 4;
        if (j < nv) {
            const uint32_t b = base_bits + j * w;
            const uint32_t idx = b >> 5, s = b & 31u;
            uint32_t d0, d1;
            if (interior) {
                const ZBRA_GLOBAL u32x2_a4 *q = (const ZBRA_GLOBAL u32x2_a4 *)(src + idx);
                u32x2 t;
                if constexpr (NT & 1) t = __builtin_nontemporal_load(q);
                else t = *q;
                d0 = t.x; d1 = t.y;
            } else {
                d0 = src[idx];
                d1 = (uint64_t)idx + 1 <= last_dword ? src[idx + 1] : 0u;
            }
            const uint32_t x = __builtin_amdgcn_alignbit(d1, d0, s);  // the 32 bits that start at b
            u32x4 v;
            v.x = mid + (uint32_t)zigzag_decode32(x & mask);
            v.y = mid + (uint32_t)zigzag_decode32((x >> w) & mask);
            v.z = mid + (uint32_t)zigzag_decode32((x >> (2 * w)) & mask);
            v.w = mid + (uint32_t)zigzag_decode32((x >> (3 * w)) & mask);
            if (j + 3 < nv) {
                st_stream<NT>(v, (ZBRA_GLOBAL u32x4 *)(out + j));
            } else {
                out[j] = v.x;
                if (j + 1 < nv) out[j + 1] = v.y;
                if (j + 2 < nv) out[j + 2] = v.z;
            }
        }
    }
}

// Raw fallback (width byte >= 32): value i is the u64 at payload + 8*i when all 8 bytes are
// present, else 0 (compression.rs:195-207).
template <int DT, int TH, int NT>
__device__ __forceinline__ void tile_raw(const DecodePlan &p, uint64_t v0, uint32_t nv) {
    const uint32_t tid = threadIdx.x;
    const uintptr_t A = (uintptr_t)p.payload + v0 * 8;  // v0*8 is a multiple of 16
    const uintptr_t A0 = A & ~(uintptr_t)15;
    const uint32_t m = (uint32_t)(A - A0);
    const uint32_t q = m >> 2, r = m & 3u;
    const uintptr_t end = (uintptr_t)p.payload + p.avail;
    const int64_t mid = p.midpoint;
#pragma unroll
    for (int it = 0; it < DT / (TH * 2); ++it) {
        const uint32_t j = it * TH * 2 + tid * 2;
        if (j < nv) {
            const uintptr_t c0 = A0 + (uintptr_t)j * 8;
            u32x4 x = {0, 0, 0, 0}, y = {0, 0, 0, 0};
            if (c0 < end) x = ld_stream<NT>(gptr<u32x4>(c0));
            if (m != 0 && c0 + 16 < end) y = ld_stream<NT>(gptr<u32x4>(c0 + 16));
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
                store_pair_i64<NT>(gptr_mut<int64_t>(p.out) + v0, j, nv, o0, o1);
            } else {
                ZBRA_GLOBAL uint32_t *out = gptr_mut<uint32_t>(p.out) + v0;
                out[j] = (uint32_t)o0;
                if (j + 1 < nv) out[j + 1] = (uint32_t)o1;
            }
        }
    }
}

// Width byte 0: every unpacked value is 0 (compression.rs:187-189), so every output is the midpoint.
template <int TH>
__device__ __forceinline__ void tile_fill(const DecodePlan &p, uint64_t v0, uint32_t nv) {
    const uint32_t tid = threadIdx.x;
    for (uint32_t j = tid; j < nv; j += TH) {
        if (p.kind != ZBRA_ELEM_U32) gptr_mut<int64_t>(p.out)[v0 + j] = p.midpoint;
        else gptr_mut<uint32_t>(p.out)[v0 + j] = (uint32_t)p.midpoint;
    }
}

// One tile of one array.
template <int DT, int TH, int NT, bool DIRECT>
__device__ __forceinline__ void decode_tile(const DecodePlan &p, uint64_t tile_in_array, uint32_t *lds) {
    const bool narrow = p.kind == ZBRA_ELEM_U32;
    const uint32_t tv = narrow ? 2 * DT : DT;
    const uint64_t v0 = tile_in_array * tv;
    if (v0 >= p.count) return;  // rejected or empty array: its tiles do nothing
    const uint64_t left = p.count - v0;
    const uint32_t nv = left < (uint64_t)tv ? (uint32_t)left : tv;
    if (p.width - 1u < 31u) {
        if (narrow) {
            if (DIRECT && p.width <= DIRECT_U32_MAX_W) tile_packed_direct_u32<2 * DT, TH, NT>(p, v0, nv);
            else tile_packed<2 * DT, TH, NT>(p, v0, nv, lds);
        }
        else if (DIRECT) tile_packed_direct<DT, TH, NT>(p, v0, nv);
        else tile_packed<DT, TH, NT>(p, v0, nv, lds);
    } else if (p.width == 0) {
        tile_fill<TH>(p, v0, nv);
    } else {
        if (narrow) tile_raw<2 * DT, TH, NT>(p, v0, nv);
        else tile_raw<DT, TH, NT>(p, v0, nv);
    }
}

// Reads the packed bytes of tiles [tile_base, tile_base + ntiles) of a submission with plain loads
// and throws them away: what it leaves behind is those bytes in the memory-side cache, where the
// tile kernel that follows finds them at a fraction of the HBM latency.  Worth its time for narrow
// arrays only (max_width): their packed bytes are few and their decode is bound by how long a wave
// waits for them.  Workgroup b takes tiles_per_wg consecutive tiles; the tiles of one array are
// contiguous in memory, so each run of them is one flat sweep of 16-byte chunks (every chunk holds
// at least one payload byte, hence lies inside a mapped page).
template <int DT>
__global__ __launch_bounds__(256) void warm_tiles_kernel(const DecodePlan *__restrict__ plans, const uint64_t *__restrict__ tile_start,
                                                         const uint32_t *__restrict__ tile_map, uint32_t nd, uint64_t tile_base,
                                                         uint64_t ntiles, uint32_t tiles_per_wg, uint32_t max_width) {
    uint64_t t = tile_base + (uint64_t)blockIdx.x * tiles_per_wg;
    const uint64_t tend = t + tiles_per_wg < tile_base + ntiles ? t + tiles_per_wg : tile_base + ntiles;
    u32x4 acc = {0, 0, 0, 0};
    while (t < tend) {  // uniform
        const uint32_t a = nd > 1 ? tile_map[t] : 0;
        const DecodePlan p = plans[a];
        const uint64_t a_first = tile_start[a], a_end = tile_start[a + 1];
        const uint64_t run_end = a_end < tend ? a_end : tend;  // tiles [t, run_end) belong to array a
        const uint64_t tv = p.kind == ZBRA_ELEM_U32 ? 2 * DT : DT;
        const uint64_t v0 = (t - a_first) * tv;
        if (p.width - 1u < max_width && v0 < p.count) {
            uint64_t v1 = (run_end - a_first) * tv;
            if (v1 > p.count) v1 = p.count;
            const uintptr_t b0 = (uintptr_t)p.payload + ((v0 * p.width) >> 3), b1 = (uintptr_t)p.payload + ((v1 * p.width + 7) >> 3);
            const uintptr_t a0 = b0 & ~(uintptr_t)15;
            const uint64_t nchunks = (b1 - a0 + 15) >> 4;
            const ZBRA_GLOBAL u32x4 *src = gptr<u32x4>(a0);
            for (uint64_t c = threadIdx.x; c < nchunks; c += 256) acc ^= src[c];
        }
        t = run_end;
    }
    asm volatile("" ::"v"(acc.x), "v"(acc.y), "v"(acc.z), "v"(acc.w));  // the loads have a use
}

// Batched submission: plans come from plan_decode_batch_kernel.  A launch covers the tiles
// [tile_base, tile_base + ntiles) of the submission: the host cuts a submission whose one
// workgroup per tile would pass the 2^32 work-items of a dispatch into several launches.
template <int DT, int TH, int NT, bool DIRECT>
__global__ __launch_bounds__(TH) void decode_tiles_kernel(const DecodePlan *__restrict__ plans,
                                                          const uint64_t *__restrict__ tile_start,
                                                          const uint32_t *__restrict__ tile_map, uint32_t nd,
                                                          uint64_t tile_base, uint64_t ntiles, uint32_t xcd_chunked) {
    // u32 outputs are half as wide, so their tiles hold twice the values (same bytes written)
    __shared__ __attribute__((aligned(16))) uint32_t lds[lds_dwords<2 * DT>()];
    // xcd_chunked implies gridDim.x == ntiles: exactly one trip through the loop
    uint64_t i = xcd_chunked == 1 ? xcd_chunked_tile(blockIdx.x, ntiles) : xcd_chunked ? xcd_grouped_tile(blockIdx.x, ntiles, xcd_chunked) : blockIdx.x;
    for (; i < ntiles; i += gridDim.x) {
        const uint64_t t = tile_base + i;
        const uint32_t a = nd > 1 ? tile_map[t] : 0;
        const DecodePlan p = plans[a];
        decode_tile<DT, TH, NT, DIRECT>(p, t - tile_start[a], lds);
    }
}

// ---- experimental shape: two tiles per workgroup, packed bytes brought in by LDS-DMA ----------
//
// What holds the tile kernel above below the chip's copy rate is, by its counters, the time a wave
// waits for its few packed bytes before it can store.  This shape asks for the packed bytes of TWO
// tiles at once with global_load_lds_dwordx4 (the load writes LDS itself: no registers hold the
// tile in flight, and the wave is free to go on), decodes the first tile as soon as ITS bytes have
// landed and the second behind it.  The waits are counted by hand: vector-memory operations of a
// wave (loads, LDS-DMA, stores) retire in issue order, so `s_waitcnt vmcnt(N)` with N = the
// operations issued AFTER a tile's DMA returns when that DMA has landed and leaves the younger
// operations (the other tile's DMA, the first tile's stores) in flight.  A compiler-inserted
// vmcnt(0) would also wait for the first tile's 16 KiB of stores, which is what made every
// register-prefetch shape of this kernel slower.  N must never exceed what was really issued (a
// wait that is too weak reads stale LDS), so it is only above zero where the count is exact: this
// wave's own DMA instructions, and the two store instructions of a FULL packed tile.
// Selected with zbra_tuning_set_decode_variant(2048, 512, nt, 2, xcd); not the default: timed cold
// against the one-tile kernel it lost at every width (u32 tags at 2^29 rows 0.469 against 0.407 ms,
// width 7 0.236 / 0.209, the 2^28-row width-12 column 0.440 / 0.424, width 26 0.256 / 0.244): with 32
// waves per CU all loading, the workgroups' overlap already hides the packed bytes' latency, and a
// workgroup that holds two tiles halves the independent workgroups whose stores interleave.

// One wave-instruction: lanes [0, 64) copy 16 bytes each from their own global address to
// lds_dst + 16 * lane (lds_dst is a wave-uniform LDS byte address).  M0 carries the LDS base; it
// is saved and restored inside the statement because the compiler does not expect it to change.
template <int NT>
__device__ __forceinline__ void glds16(uintptr_t gsrc, uint32_t lds_dst) {
    unsigned keep;
    if constexpr (NT & 1)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
    else
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

// Waits until at most n of this wave's vector-memory operations are outstanding, then for the
// workgroup.  n is wave-uniform and small.
__device__ __forceinline__ void wait_vm_then_barrier(uint32_t n) {
    switch (n) {
        case 0: asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
    }
}

struct DmaTile {
    DecodePlan p;
    uint64_t v0;
    uint32_t nv, tv;     // values in this tile (0 = nothing to do), values in a full tile of its kind
    uint32_t base_bits;  // of the tile's first value from the start of its LDS image
    uint32_t issued;     // DMA instructions this wave issued for it
    bool packed;
};

template <int DT, int TH, int NT>
__global__ __launch_bounds__(TH) void decode_tiles_dma_kernel(const DecodePlan *__restrict__ plans, const uint64_t *__restrict__ tile_start,
                                                              const uint32_t *__restrict__ tile_map, uint32_t nd, uint64_t tile_base,
                                                              uint64_t ntiles, uint32_t xcd_chunked) {
    constexpr int BUF = (lds_dwords<2 * DT>() + 255) / 256 * 256;  // whole 1-KiB wave pieces
    __shared__ __attribute__((aligned(1024))) uint32_t lds[2][BUF];
    const uint64_t npairs = (ntiles + 1) >> 1;  // gridDim.x == npairs
    const uint64_t pr = xcd_chunked == 1 ? xcd_chunked_tile(blockIdx.x, npairs) : xcd_chunked ? xcd_grouped_tile(blockIdx.x, npairs, xcd_chunked) : blockIdx.x;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63u;
    DmaTile t[2];
#pragma unroll
    for (int q = 0; q < 2; q++) {
        DmaTile &x = t[q];
        x.nv = 0; x.issued = 0; x.packed = false; x.base_bits = 0; x.v0 = 0; x.tv = DT;
        const uint64_t i = 2 * pr + q;
        if (i >= ntiles) continue;  // uniform
        const uint64_t T = tile_base + i;
        const uint32_t a = nd > 1 ? tile_map[T] : 0;
        x.p = plans[a];
        x.tv = x.p.kind == ZBRA_ELEM_U32 ? 2 * DT : DT;
        x.v0 = (T - tile_start[a]) * x.tv;
        if (x.v0 >= x.p.count) continue;  // rejected or empty array
        const uint64_t left = x.p.count - x.v0;
        x.nv = left < (uint64_t)x.tv ? (uint32_t)left : x.tv;
        x.packed = x.p.width - 1u < 31u;
        if (x.packed) {
            const PackedSpan sp = packed_span(x.p, x.v0, x.nv);
            x.base_bits = sp.base_bits;
            const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t *)&lds[q][0];
            for (uint32_t c0 = wave * 64u; c0 < sp.nchunks; c0 += TH) {  // uniform trip count within the wave
                if (c0 + lane < sp.nchunks) glds16<NT>(sp.A0 + (uintptr_t)(c0 + lane) * 16, __builtin_amdgcn_readfirstlane(lds_base + c0 * 16u));
                x.issued++;
            }
        }
    }
    // first tile: its DMA is older than the second tile's, which stays in flight
    if (t[0].nv) {
        if (t[0].packed) {
            wait_vm_then_barrier(t[1].issued);
            if (t[0].p.kind == ZBRA_ELEM_U32) packed_extract<2 * DT, TH, NT>(t[0].p, t[0].v0, t[0].nv, t[0].base_bits, lds[0]);
            else packed_extract<DT, TH, NT>(t[0].p, t[0].v0, t[0].nv, t[0].base_bits, lds[0]);
        } else if (t[0].p.width == 0) {
            tile_fill<TH>(t[0].p, t[0].v0, t[0].nv);
        } else if (t[0].p.kind == ZBRA_ELEM_U32) {
            tile_raw<2 * DT, TH, NT>(t[0].p, t[0].v0, t[0].nv);
        } else {
            tile_raw<DT, TH, NT>(t[0].p, t[0].v0, t[0].nv);
        }
    }
    if (t[1].nv) {
        if (t[1].packed) {
            // behind the second tile's DMA this wave has issued exactly two stores when the first tile
            // was a full packed one; anything else is not counted on
            const bool exact = t[0].nv && t[0].packed && t[0].nv == t[0].tv;
            wait_vm_then_barrier(exact ? 2u : 0u);
            if (t[1].p.kind == ZBRA_ELEM_U32) packed_extract<2 * DT, TH, NT>(t[1].p, t[1].v0, t[1].nv, t[1].base_bits, lds[1]);
            else packed_extract<DT, TH, NT>(t[1].p, t[1].v0, t[1].nv, t[1].base_bits, lds[1]);
        } else if (t[1].p.width == 0) {
            tile_fill<TH>(t[1].p, t[1].v0, t[1].nv);
        } else if (t[1].p.kind == ZBRA_ELEM_U32) {
            tile_raw<2 * DT, TH, NT>(t[1].p, t[1].v0, t[1].nv);
        } else {
            tile_raw<DT, TH, NT>(t[1].p, t[1].v0, t[1].nv);
        }
    }
}

}  // namespace zbra
