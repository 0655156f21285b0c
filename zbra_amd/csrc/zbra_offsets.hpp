// Lengths -> offsets, fused with the decode.
//
// Binary, Array and Nested columns store their lengths through the int codec
// (read_int_array_usize_compressed, zbra-core/src/binary.rs:661-664), and the first thing every
// consumer does with them is a running sum (`offset += length`, zbra-core/src/striped.rs:523-583).
// These kernels produce that running sum straight from the packed bytes:
//   offsets[0] = 0, offsets[i + 1] = offsets[i] + lengths[i]          (wrapping, as usize adds)
// in four launches: per-tile sums (decode, reduce, store nothing), a two-level exclusive scan of
// the tile sums, and a second decode that scans inside the tile and stores the offsets.  HBM traffic is two
// reads of the packed bytes and one write of 8 B per value; decoding the lengths first and
// scanning them afterwards would write, read and write 8 B per value.
//
// A submission holds any number of independent lengths arrays (the Binary, Array and Nested
// columns of one Block, or of many): tile T of the submission belongs to array a = tile_map[T] and
// is that array's tile T - tile_start[a], exactly as in the batched decode.  The tile sums of ALL
// arrays are scanned as one sequence; array a's running sum starts at zero because its tiles
// subtract the scanned value of the array's first tile (wrapping arithmetic makes that exact).
#pragma once
#include "zbra_decode.hpp"

namespace zbra {

// Loads of the packed bytes: plain (0).  The sums pass reads them once and leaves them in the
// memory-side cache, where the write pass finds them again (2^28 lengths of width 6, caches emptied
// before the call: 0.680 -> 0.542 ms); nontemporal loads (1) leave nothing behind.
#ifndef ZBRA_OFF_NT
#define ZBRA_OFF_NT 0
#endif
constexpr int OFF_TILE = 2048;     // values per tile
constexpr int OFF_THREADS = 512;   // lanes per workgroup
constexpr int OFF_PER = OFF_TILE / OFF_THREADS;  // consecutive values per lane: one wave scan per tile
constexpr int OFF_WAVES = OFF_THREADS / 64;
static_assert(OFF_PER == 4, "a lane stores its four offsets as two 16-byte words");

// Stages the packed bytes of the tile in LDS when the width is 1..31 (nothing to do otherwise).
__device__ __forceinline__ uint32_t offsets_stage(const DecodePlan &p, uint64_t v0, uint32_t nv, uint32_t *lds) {
    uint32_t base_bits = 0;
    if (p.width - 1u < 31u) {
        const PackedSpan sp = packed_span(p, v0, nv);
        const ZBRA_GLOBAL u32x4 *src = gptr<u32x4>(sp.A0);
        u32x4 *dst = (u32x4 *)lds;
        for (uint32_t c = threadIdx.x; c < sp.nchunks; c += OFF_THREADS) dst[c] = ld_stream<ZBRA_OFF_NT>(src + c);
        base_bits = sp.base_bits;
    }
    __syncthreads();
    return base_bits;
}

// Values j .. j + 3 of the tile (0 past the last one), decoded as decompress_int_array would
// (compression.rs:295-330): un-zig-zag, add the midpoint.
__device__ __forceinline__ void offsets_values(const DecodePlan &p, uint64_t v0, uint32_t nv, uint32_t base_bits, const uint32_t *lds,
                                               uint32_t j, uint64_t (&v)[OFF_PER]) {
    const uint32_t w = p.width;
#pragma unroll
    for (int k = 0; k < OFF_PER; k++) {
        int64_t d;
        if (w - 1u < 31u) {
            d = zigzag_decode32(extract_bits(lds, base_bits + (j + k) * w, (1u << w) - 1u));
        } else if (w == 0) {
            d = 0;                                            // compression.rs:187-189
        } else {                                              // raw u64, a value with missing bytes reads as 0 (:195-207)
            const uint64_t off = (v0 + j + k) * 8;
            d = zigzag_decode64(j + k < nv && off + 8 <= p.avail ? load_le64_bytes(p.payload + off) : 0);
        }
        v[k] = j + k < nv ? (uint64_t)(p.midpoint + d) : 0;
    }
}

// The same for widths up to OFF_DIRECT_MAX_W without the LDS image: the lane's four values lie inside
// the 64 bits that start at its first value, which three dwords hold; the lane loads them itself
// (plain loads: neighbouring lanes read overlapping dwords, and the sums pass has left the packed
// bytes in the caches).  Dwords past the last payload byte are never touched.
constexpr uint32_t OFF_DIRECT_MAX_W = 16;
__device__ __forceinline__ void offsets_values_direct(const DecodePlan &p, uint64_t v0, uint32_t nv, uint32_t j, uint64_t (&v)[OFF_PER]) {
    const uint32_t w = p.width;
#pragma unroll
    for (int k = 0; k < OFF_PER; k++) v[k] = 0;
    if (j >= nv) return;
    const uintptr_t P = (uintptr_t)p.payload;
    const uint64_t bit = (v0 + j) * w;                       // of the lane's first value, from the payload's first byte
    const uintptr_t A = P + (bit >> 3);                      // the byte that holds it
    const uintptr_t A4 = A & ~(uintptr_t)3;
    const uint32_t s = (uint32_t)(A - A4) * 8u + (uint32_t)(bit & 7u);   // 0..31
    const uintptr_t last = (P + p.avail - 1) & ~(uintptr_t)3;             // dword of the last payload byte
    const uint32_t d0 = *gptr<uint32_t>(A4);
    const uint32_t d1 = A4 + 4 <= last ? *gptr<uint32_t>(A4 + 4) : 0u;
    const uint32_t d2 = A4 + 8 <= last ? *gptr<uint32_t>(A4 + 8) : 0u;
    const uint64_t x = ((uint64_t)__builtin_amdgcn_alignbit(d2, d1, s) << 32) | __builtin_amdgcn_alignbit(d1, d0, s);  // the 64 bits that start at the value
    const uint32_t mask = (1u << w) - 1u;
#pragma unroll
    for (int k = 0; k < OFF_PER; k++)
        if (j + k < nv) v[k] = (uint64_t)(p.midpoint + (int64_t)zigzag_decode32((uint32_t)(x >> (k * w)) & mask));
}

__device__ __forceinline__ uint64_t wave_sum_u64(uint64_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// Sum of the deltas of one FULL tile of packed width W, by one wave.  The pass is bound by its
// vector instructions, so the extraction is made straight-line: lane l takes the tile's values
// [32 l, 32 l + 32), which are exactly W dwords of the stream (32 * W bits), every field then sits
// at a compile-time position of the lane's registers, and a value costs two bit-field extracts, an
// exclusive-or and an add (a funnel shift more where it straddles two dwords).  The packed bytes
// come in with the same coalesced 16-byte loads as before and are dealt to the lanes through a
// wave-private LDS image; runs are one dword apart for even W (an odd stride in dwords: lanes then
// hit different banks).  The tile starts at any byte of memory: the stream is brought to a dword
// boundary by one funnel shift per dword (the shift, 0 / 8 / 16 / 24 bits, is the same for every
// tile of an array).
// Dwords of one wave's image for width w: 64 runs of w dwords, one dword apart for even w, and the dword behind them.
__host__ __device__ constexpr uint32_t offsets_image_dwords(uint32_t w) { return 64u * (w + (w % 2 == 0 ? 1u : 0u)) + 8u; }


// `ntl` consecutive FULL tiles of one array of packed width W, summed by one wave: sums[k] = sum of
// the values of the k-th of them.  The packed bytes of tile k + 1 are asked for before tile k is
// looked at, so only the first tile's load latency is waited for in full.
template <int W>
__device__ __forceinline__ void offsets_full_tiles_sums(const DecodePlan &p, uint64_t v0, uint32_t ntl, uint32_t *wl, uint64_t *__restrict__ sums) {
    constexpr int STRIDE = W + (W % 2 == 0 ? 1 : 0);
    constexpr int NCH = (16 * W + 1 + 63) / 64;  // 16-byte chunks per lane: a tile's 256 W bytes touch at most 16 W + 1 chunks
    const uint32_t lane = threadIdx.x & 63u;
    // every tile of an array starts at the same place of a 16-byte chunk (a tile is 256 W bytes)
    const PackedSpan sp0 = packed_span(p, v0, OFF_TILE);
    const uint32_t b0 = sp0.base_bits >> 5, r = sp0.base_bits & 31u;  // first dword of the tile inside its first chunk; bits into it
    const uint32_t nchunks = sp0.nchunks;
    auto load = [&](uint32_t k, u32x4 (&x)[NCH]) {
        const uintptr_t A0 = sp0.A0 + (uintptr_t)k * (256u * W);
#pragma unroll
        for (int i = 0; i < NCH; i++) {
            const uint32_t c = lane + 64u * i;
            if (c < nchunks) x[i] = ld_stream<ZBRA_OFF_NT>(gptr<u32x4>(A0 + (uintptr_t)c * 16));
        }
    };
    // D tiles' packed bytes are in flight per wave (in registers; the narrower the array, the more fit):
    // with one tile ahead only, the waves of a CU had 40 KB of loads outstanding, too little for the
    // latency of HBM (2^28 lengths of width 6: 87 us for this pass, most of it waiting).
    constexpr int D = NCH <= 3 ? 4 : NCH <= 5 ? 2 : 1;
    u32x4 buf[D][NCH];
#pragma unroll
    for (int j = 0; j < D; j++) {
#pragma unroll
        for (int i = 0; i < NCH; i++) buf[j][i] = u32x4{0, 0, 0, 0};
        if ((uint32_t)j < ntl) load((uint32_t)j, buf[j]);
    }
    for (uint32_t k0 = 0; k0 < ntl; k0 += D) {  // uniform
#pragma unroll
      for (int j = 0; j < D; j++) {
        const uint32_t k = k0 + (uint32_t)j;
        if (k >= ntl) break;  // uniform
        u32x4 cur[NCH];
#pragma unroll
        for (int i = 0; i < NCH; i++) cur[i] = buf[j][i];
        if (k + D < ntl) load(k + D, buf[j]);
        // this wave's reads of the image of the tile before are done before it is overwritten
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int i = 0; i < NCH; i++) {
            const uint32_t c = lane + 64u * i;
            const uint32_t x[4] = {cur[i].x, cur[i].y, cur[i].z, cur[i].w};
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const uint32_t jj = c * 4u + (uint32_t)t - b0;  // dword of the tile's stream; wraps for the dwords in front of it
                if (c < nchunks && jj <= 64u * W) wl[jj + (STRIDE - W) * (jj / W)] = x[t];
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        uint32_t d[W + 1];
#pragma unroll
        for (int q = 0; q < W; q++) d[q] = wl[lane * STRIDE + q];
        d[W] = wl[(lane + 1) * STRIDE];  // the next run's first dword (past the tile for the last lane: only used when r != 0, and then it was staged)
        if (r) {  // uniform
#pragma unroll
            for (int q = 0; q < W; q++) d[q] = __builtin_amdgcn_alignbit(d[q + 1], d[q], r);
        }
        // 32 deltas of at most 2^(W-1) in magnitude: a 32-bit sum holds them up to W = 26
        int32_t acc32 = 0;
        int64_t acc64 = 0;
#pragma unroll
        for (int i = 0; i < 32; i++) {
            const int bit = i * W, q = bit >> 5, off = bit & 31;
            const uint32_t x = off + W <= 32 ? d[q] >> off : __builtin_amdgcn_alignbit(d[q + 1 <= W ? q + 1 : W], d[q], (uint32_t)off);
            const uint32_t h = W > 1 ? (x >> 1) & ((1u << (W - 1)) - 1u) : 0u;
            const int32_t delta = (int32_t)h ^ -(int32_t)(x & 1u);  // zig_zag_decode, compression.rs:101-111
            if (W <= 26) acc32 += delta;
            else acc64 += delta;
        }
        const uint64_t sum = wave_sum_u64((uint64_t)(W <= 26 ? (int64_t)acc32 : acc64));
        if (lane == 0) sums[k] = sum + (uint64_t)OFF_TILE * (uint64_t)p.midpoint;
      }
    }
}
template <int W, int WMAX>
__device__ __forceinline__ bool offsets_full_tiles_sums_from(uint32_t w, const DecodePlan &p, uint64_t v0, uint32_t ntl, uint32_t *wl, uint32_t room,
                                                             uint64_t *__restrict__ sums) {
    if constexpr (W > WMAX) {
        return false;
    } else {
        if (w == (uint32_t)W) {
            if (offsets_image_dwords(W) > room) return false;  // the launch sized the images for narrower arrays: the caller takes the generic form
            offsets_full_tiles_sums<W>(p, v0, ntl, wl, sums);
            return true;
        }
        return offsets_full_tiles_sums_from<W + 1, WMAX>(w, p, v0, ntl, wl, room, sums);
    }
}

// 1. tile_sums[t] = sum of the tile's values.  The order of a sum does not matter, so nothing is
// staged or transposed: one wave per tile, a lane takes whole 16-byte chunks of the packed bytes
// (coalesced loads) and adds up the values that START in each of the chunk's four dwords, looking
// one dword ahead for the value that straddles the end.  No LDS, no barriers: a wave's life is one
// load latency and the waves of a SIMD overlap theirs freely.
constexpr int OFF_SUM_WAVES = 4;  // waves per workgroup
// Consecutive tiles one wave sums, one after the other.  A wave that lives for one tile of a narrow
// array (a kilobyte or two of packed bytes) spends most of its life being launched and fetching its
// plan: 2^28 lengths of width 6 are 131072 such waves.
// The launch says how many (tiles_per_wave): as many as leave the chip enough waves, 16 at most; a
// single array of 2^22 lengths keeps one tile per wave (its 2048 waves would otherwise be 128).
constexpr uint32_t OFF_SUM_TPW_MAX = 16;
constexpr uint64_t OFF_SUM_MIN_WAVES = 8192;  // about what the chip holds at once
// WMAX: widest array that takes the straight-line form in this instance.  The form keeps two tiles
// of packed bytes and a lane's run in registers, W-dependent, and a kernel is allotted the registers
// of its hungriest path: one instance for all 31 widths would run narrow arrays at the occupancy of
// width 31.  The host picks the instance from the widest array of the submission.
template <int WMAX>
__global__ __launch_bounds__(64 * OFF_SUM_WAVES) void offsets_sums_kernel(const DecodePlan *__restrict__ plans, const uint64_t *__restrict__ tile_start,
                                                                          const uint32_t *__restrict__ tile_map, uint32_t nd, uint64_t ntiles,
                                                                          uint64_t *__restrict__ tile_sums, uint32_t image_dwords,
                                                                          uint32_t tiles_per_wave) {
    // image_dwords: LDS dwords per wave (the launch's dynamic LDS is OFF_SUM_WAVES times that), sized by
    // the host for the widest array of the submission as its frame lengths bound it
    extern __shared__ uint32_t images[];
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t T0 = ((uint64_t)blockIdx.x * OFF_SUM_WAVES + (threadIdx.x >> 6)) * tiles_per_wave;
    if (T0 >= ntiles) return;  // uniform within the wave
    {
        // the wave's tiles are full packed tiles of ONE array (the usual case): the straight-line form
        const uint64_t Tl = T0 + tiles_per_wave <= ntiles ? T0 + tiles_per_wave - 1 : ntiles - 1;
        const uint32_t a = nd > 1 ? tile_map[T0] : 0;
        if (nd == 1 || tile_map[Tl] == a) {
            const DecodePlan p = plans[a];
            const uint64_t t0 = T0 - tile_start[a], tl = Tl - tile_start[a];
            if (p.width - 1u < 31u && (tl + 1) * OFF_TILE <= p.count &&
                offsets_full_tiles_sums_from<1, WMAX>(p.width, p, t0 * OFF_TILE, (uint32_t)(tl - t0 + 1), images + (threadIdx.x >> 6) * image_dwords, image_dwords,
                                                tile_sums + T0))
                return;
        }
    }
  for (uint64_t T = T0; T < T0 + tiles_per_wave && T < ntiles; T++) {  // uniform within the wave
    const uint32_t a = nd > 1 ? tile_map[T] : 0;
    const DecodePlan p = plans[a];
    const uint64_t t = T - tile_start[a];
    const uint64_t v0 = t * OFF_TILE;
    if (v0 >= p.count) {  // a tile of a rejected array (its frame's count gave it tiles): nothing to add
        if (lane == 0) tile_sums[T] = 0;
        continue;
    }
    const uint64_t left = p.count - v0;
    const uint32_t nv = left < (uint64_t)OFF_TILE ? (uint32_t)left : (uint32_t)OFF_TILE;
    const uint32_t w = p.width;
    uint64_t sum = 0;  // of the deltas; the midpoints are added at the end
    if (w - 1u < 31u) {  // a run of tiles with a ragged last one or of several arrays, or an image too small for this width
        const PackedSpan sp = packed_span(p, v0, nv);
        const uint32_t mask = (1u << w) - 1u;
        // n / w == (n * magic) >> 21 for n < 2^16: magic * w - 2^21 < w <= 31 and n * 31 < 2^21.
        // n is a bit offset inside the tile's chunks, below 128 * (16 * 31 + 1) + 31 < 2^16.
        const uint64_t magic = ((1u << 21) + w - 1u) / w;
        const uint32_t end_bits = sp.base_bits + nv * w;  // one past the tile's last bit, counted from A0
        const uintptr_t last = ((uintptr_t)p.payload + p.avail - 1) & ~(uintptr_t)15;  // chunk of the last payload byte
        for (uint32_t c = lane; c < sp.nchunks; c += 64) {
            const uintptr_t a = sp.A0 + (uintptr_t)c * 16;
            const u32x4 x = ld_stream<ZBRA_OFF_NT>(gptr<u32x4>(a));
            const uint32_t ahead = a + 16 <= last ? ld_stream<ZBRA_OFF_NT>(gptr<uint32_t>(a + 16)) : 0u;
            const uint32_t d[5] = {x.x, x.y, x.z, x.w, ahead};
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const uint32_t lo = c * 128u + q * 32u;  // first bit of dword q, counted from A0
                // first value that starts at or after lo, and where in the dword it starts
                const uint32_t k = lo > sp.base_bits ? (uint32_t)(((lo - sp.base_bits + w - 1u) * magic) >> 21) : 0u;
                uint32_t s = sp.base_bits + k * w - lo;
                const uint32_t lim = end_bits > lo ? (end_bits - lo < 32u ? end_bits - lo : 32u) : 0u;
                // at most two 31-bit values start in one dword, |delta| <= 2^30 each: no overflow
                int32_t acc = 0;
                for (; s < lim; s += w) acc += zigzag_decode32(__builtin_amdgcn_alignbit(d[q + 1], d[q], s) & mask);
                sum += (uint64_t)(int64_t)acc;
            }
        }
    } else if (w >= 32) {  // raw u64, a value with missing bytes reads as 0 (compression.rs:195-207)
        for (uint32_t j = lane; j < nv; j += 64) {
            const uint64_t off = (v0 + j) * 8;
            sum += (uint64_t)zigzag_decode64(off + 8 <= p.avail ? load_le64_bytes(p.payload + off) : 0);
        }
    }  // width byte 0: every delta is 0 (compression.rs:187-189)
    sum = wave_sum_u64(sum);
    if (lane == 0) tile_sums[T] = sum + (uint64_t)nv * (uint64_t)p.midpoint;
  }
}

// 2. Exclusive scan of the tile sums in two levels: every workgroup scans one chunk of SCAN_CHUNK
// consecutive sums in place (coalesced: lane t holds sum t of the chunk) and leaves the chunk's
// total in chunk_base[c]; one workgroup then turns the totals into exclusive prefixes.  A
// 2^32-value array has 2^21 tiles = 2048 chunks, which the second level takes two per lane.
constexpr int SCAN_CHUNK = 1024;
__device__ __forceinline__ uint64_t block_exclusive_scan_1024(uint64_t x, uint64_t *wtot, uint64_t *total) {
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint64_t inc = x;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint64_t up = __shfl_up(inc, o);
        if (lane >= (uint32_t)o) inc += up;
    }
    if (lane == 63) wtot[wave] = inc;
    __syncthreads();
    uint64_t before = 0, all = 0;
    for (uint32_t k = 0; k < SCAN_CHUNK / 64; k++) { if (k < wave) before += wtot[k]; all += wtot[k]; }
    __syncthreads();
    *total = all;
    return before + inc - x;
}
__global__ __launch_bounds__(SCAN_CHUNK) void offsets_scan_chunks_kernel(uint64_t ntiles, uint64_t *__restrict__ tile_sums,
                                                                          uint64_t *__restrict__ chunk_base) {
    __shared__ uint64_t wtot[SCAN_CHUNK / 64];
    const uint64_t t = (uint64_t)blockIdx.x * SCAN_CHUNK + threadIdx.x;
    if ((uint64_t)blockIdx.x * SCAN_CHUNK >= ntiles) return;  // uniform
    const uint64_t x = t < ntiles ? tile_sums[t] : 0;
    uint64_t total;
    const uint64_t ex = block_exclusive_scan_1024(x, wtot, &total);
    if (t < ntiles) tile_sums[t] = ex;
    if (threadIdx.x == 0) chunk_base[blockIdx.x] = total;
}
__global__ __launch_bounds__(SCAN_CHUNK) void offsets_scan_totals_kernel(uint64_t nchunks, uint64_t *__restrict__ chunk_base) {
    __shared__ uint64_t wtot[SCAN_CHUNK / 64];
    uint64_t carry = 0;
    for (uint64_t c0 = 0; c0 < nchunks; c0 += SCAN_CHUNK) {  // uniform trip count
        const uint64_t c = c0 + threadIdx.x;
        const uint64_t x = c < nchunks ? chunk_base[c] : 0;
        uint64_t total;
        const uint64_t ex = block_exclusive_scan_1024(x, wtot, &total);
        if (c < nchunks) chunk_base[c] = carry + ex;
        carry += total;
    }
}

// 3. offsets[v0 + j] = (sum of all earlier tiles) + (sum of the tile's values before j), i.e. the
// EXCLUSIVE prefix of value j goes to index v0 + j: a lane's four offsets then start at a multiple
// of four and its two 16-byte stores are aligned whenever `offsets` is (writing the inclusive
// prefixes to v0 + j + 1 instead straddles every 16-byte word and cost a third more time).
// offsets[0] = 0 falls out; offsets[count] is written by the lane that holds the last value.
// `chunk_base[c]` is the sum of the submission's tiles before chunk c of SCAN_CHUNK tiles and
// `tile_sums[T]` the sum of the earlier tiles of T's own chunk; the array's offsets (plan.out) start
// at zero: what the scan holds for the array's first tile is subtracted.
__global__ __launch_bounds__(OFF_THREADS) void offsets_write_kernel(const DecodePlan *__restrict__ plans, const uint64_t *__restrict__ tile_start,
                                                                     const uint32_t *__restrict__ tile_map, uint32_t nd,
                                                                     const uint64_t *__restrict__ tile_sums, const uint64_t *__restrict__ chunk_base,
                                                                     uint32_t reverse) {
    __shared__ __attribute__((aligned(16))) uint32_t lds[lds_dwords<OFF_TILE>()];
    __shared__ __attribute__((aligned(16))) u64x2 turn[OFF_WAVES][64 * OFF_PER / 2];  // a wave's 256 offsets
    __shared__ uint64_t wtot[OFF_WAVES];
    // back to front when the packed bytes are more than the cache holds: what the sums pass read
    // last is what is still there (2^29 lengths of width 6: 1.277 -> 1.122 ms)
    const uint64_t T = reverse ? gridDim.x - 1 - blockIdx.x : blockIdx.x;
    const uint32_t a = nd > 1 ? tile_map[T] : 0;
    const DecodePlan p = plans[a];
    const uint64_t S = tile_start[a];
    const uint64_t tile = T - S;
    uint64_t *offsets = (uint64_t *)p.out;
    if (p.count == 0) {  // rejected on the device (its frame's count gave it tiles): the running sum of nothing
        if (tile == 0 && threadIdx.x == 0) offsets[0] = 0;
        return;
    }
    const uint64_t v0 = tile * OFF_TILE;
    if (v0 >= p.count) return;  // uniform
    const uint64_t left = p.count - v0;
    const uint32_t nv = left < (uint64_t)OFF_TILE ? (uint32_t)left : (uint32_t)OFF_TILE;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, j = threadIdx.x * OFF_PER;
    uint64_t v[OFF_PER];
#ifndef ZBRA_OFF_NO_DIRECT
    if (p.width - 1u < OFF_DIRECT_MAX_W) {  // uniform
        offsets_values_direct(p, v0, nv, j, v);
    } else
#endif
    {
        const uint32_t base_bits = offsets_stage(p, v0, nv, lds);
        offsets_values(p, v0, nv, base_bits, lds, j, v);
    }
    const uint64_t mine = (v[0] + v[1]) + (v[2] + v[3]);
    uint64_t inc = mine;  // inclusive scan of the lanes' totals across the wave
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint64_t up = __shfl_up(inc, o);
        if (lane >= (uint32_t)o) inc += up;
    }
    if (lane == 63) wtot[wave] = inc;
    __syncthreads();
    uint64_t run = (chunk_base[T / SCAN_CHUNK] + tile_sums[T]) - (chunk_base[S / SCAN_CHUNK] + tile_sums[S]) + inc - mine;
    for (uint32_t k = 0; k < wave; k++) run += wtot[k];
    uint64_t o[OFF_PER + 1];
#pragma unroll
    for (int k = 0; k < OFF_PER; k++) { o[k] = run; run += v[k]; }
    o[OFF_PER] = run;
    ZBRA_GLOBAL uint64_t *out = gptr_mut<uint64_t>(offsets) + v0 + j;
    if ((wave + 1) * 64 * OFF_PER <= nv) {  // the wave's 256 offsets are all there (uniform within the wave)
        // A lane holds 32 contiguous bytes, so a 16-byte store per lane would leave every other
        // 16 bytes of the wave's 2 KiB to the next instruction; the wave turns its block around in
        // LDS instead and each of its two store instructions writes a dense 1 KiB.
        u64x2 *mine_t = turn[wave];
        mine_t[lane * 2] = u64x2{o[0], o[1]};
        mine_t[lane * 2 + 1] = u64x2{o[2], o[3]};
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        ZBRA_GLOBAL u64x2 *dst = (ZBRA_GLOBAL u64x2 *)(gptr_mut<uint64_t>(offsets) + v0 + wave * 64 * OFF_PER);
        __builtin_nontemporal_store(mine_t[lane], dst + lane);
        __builtin_nontemporal_store(mine_t[64 + lane], dst + 64 + lane);
    } else {
#pragma unroll
        for (int k = 0; k < OFF_PER; k++)
            if (j + k < nv) out[k] = o[k];
    }
    if (v0 + nv == p.count) {  // the array ends in this tile: the total goes one past the last value
#pragma unroll
        for (int k = 0; k < OFF_PER; k++)
            if (j + k + 1 == nv) out[k + 1] = o[k + 1];
    }
}

// Arrays whose frame says count 0 have no tile: their one offset is written here (one lane per array).
__global__ void offsets_empty_kernel(const zbra_array_desc *__restrict__ descs, uint32_t nd) {
    const uint32_t a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a < nd && descs[a].count == 0 && descs[a].out) ((uint64_t *)descs[a].out)[0] = 0;
}

}  // namespace zbra
