// Rows of an Enum column dealt to their variants.
//
// Column::from_values (zbra-core/src/striped.rs:340-387) walks the rows once: it keeps every tag
// (:347) and pushes the row's payload onto variant_data[v], v being the FIRST variant whose tag
// equals the row's (`variants.iter().position(|v| v.tag == *tag)`, :349-351); a tag that no variant
// carries ends the conversion with SchemaError::UnsupportedType("enum tag N") (:352-358).  What that
// loop computes for row i is therefore a pair (v, position of the row in variant v's column), the
// position being the number of earlier rows that went to the same variant.  These kernels produce
// that pair for every row from the tag column alone, so that any payload column already in HBM can
// be scattered to (or gathered from) its variant columns by index:
//   variant[i] = v,  rank[i] = |{ j < i : variant[j] == v }|,  counts[v] = rows of variant v.
// Three steps, like every stable partition: per-tile counts, a scan of the counts over the tiles
// (per variant), and a second read of the tags that adds the pieces up.  HBM traffic: the tags
// twice (4 B each), 4 + 8 B written per row.
//
// A submission holds the tag arrays of any number of independent Enum columns with the same variant
// list (the same column in many Blocks): tile T of the submission belongs to array a = tile_map[T]
// and is that array's tile T - tile_start[a].  The per-variant tile counts of ALL arrays are scanned
// as one sequence, and an array's ranks start at zero because its tiles subtract the scanned value
// of the array's first tile.
//
// The other direction (Column::to_values, striped.rs:617-657) needs no kernel: the reference reads
// variant column v at the ROW index (`values[row_idx]`, :631-632), not at the rank.
#pragma once
#include "zbra_encode.hpp"
#include "zbra_offsets.hpp"

namespace zbra {

constexpr int ENUM_MAX_VARIANTS = 32;  // five ballots tell a lane which lanes hold its variant
constexpr int ENUM_TILE = 2048;        // rows per tile
constexpr int ENUM_THREADS = 256;
constexpr int ENUM_WAVES = ENUM_THREADS / 64;
constexpr int ENUM_PER = ENUM_TILE / ENUM_THREADS;  // rows per lane: a wave owns 512 consecutive rows, 64 per step
constexpr uint32_t ENUM_LUT = 256;     // variant tags below this are looked up, not searched
constexpr uint32_t ENUM_NONE = 0xFFFFFFFFu;
constexpr int ENUM_FEW = 8;            // up to this many variants the counting pass keeps byte counters in two registers

// The arrays of one submission (device memory, except nd / ntiles).
struct EnumBatch {
    const zbra_enum_desc *arrays;  // tags, n, variant, rank, counts of each array
    const uint64_t *tile_start;    // nd + 1: first tile of each array
    const uint32_t *tile_map;      // tile -> array (not read when nd == 1)
    uint32_t nd;
    uint64_t ntiles;
};
// Array and first tile of that array for tile T of the submission.
__device__ __forceinline__ void enum_locate(const EnumBatch &b, uint64_t T, uint32_t &a, uint64_t &S) {
    a = b.nd > 1 ? b.tile_map[T] : 0u;
    S = b.tile_start[a];
}

struct EnumShared {
    uint32_t vtag[ENUM_MAX_VARIANTS];
    uint8_t lut[ENUM_LUT];                             // tag -> variant, 0xFF = none
    uint32_t wcnt[ENUM_WAVES][ENUM_MAX_VARIANTS];      // rows of each variant in each wave's 512 rows
};

// Fills vtag and lut; every lane of the workgroup calls it.  Ends with a barrier.
__device__ __forceinline__ void enum_setup(EnumShared &sh, const uint32_t *__restrict__ variant_tags, uint32_t nv) {
    for (uint32_t i = threadIdx.x; i < ENUM_LUT; i += ENUM_THREADS) sh.lut[i] = 0xFF;
    if (threadIdx.x < (uint32_t)ENUM_MAX_VARIANTS) sh.vtag[threadIdx.x] = threadIdx.x < nv ? variant_tags[threadIdx.x] : 0;
    for (uint32_t i = threadIdx.x; i < ENUM_WAVES * ENUM_MAX_VARIANTS; i += ENUM_THREADS) (&sh.wcnt[0][0])[i] = 0;
    __syncthreads();
    if (threadIdx.x == 0) {  // descending, so that the FIRST variant with a tag wins (striped.rs:349)
        for (uint32_t v = nv; v-- > 0;)
            if (sh.vtag[v] < ENUM_LUT) sh.lut[sh.vtag[v]] = (uint8_t)v;
    }
    __syncthreads();
}

__device__ __forceinline__ uint32_t enum_variant_of(const EnumShared &sh, uint32_t tag, uint32_t nv) {
    if (tag < ENUM_LUT) {
        const uint32_t v = sh.lut[tag];
        return v == 0xFF ? ENUM_NONE : v;
    }
    for (uint32_t v = 0; v < nv; v++)
        if (sh.vtag[v] == tag) return v;
    return ENUM_NONE;
}

// The wave's 512 rows of the tile that starts at row r0, 64 consecutive rows per step: variant of
// this lane's row of every step and its rank among the WAVE's rows of that variant so far.  A lane
// finds the lanes that hold its variant with five ballots (one per bit of v); the lowest of them
// adds the group's size to the wave's counter of that variant in LDS and hands the old value on.
// first_bad = lowest row whose tag is no variant's (or ~0).
__device__ __forceinline__ void enum_wave_rows(EnumShared &sh, const uint32_t *__restrict__ tags, uint64_t r0, uint64_t n, uint32_t nv,
                                               uint32_t (&v)[ENUM_PER], uint32_t (&r)[ENUM_PER], uint64_t &first_bad) {
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint64_t below = lane ? (~0ull >> (64 - lane)) : 0ull;
    uint32_t t[ENUM_PER];
#pragma unroll
    for (int e = 0; e < ENUM_PER; e++) {
        const uint64_t row = r0 + wave * (64u * ENUM_PER) + (uint32_t)e * 64u + lane;
        t[e] = row < n ? tags[row] : 0u;
    }
#pragma unroll
    for (int e = 0; e < ENUM_PER; e++) {
        const uint64_t row = r0 + wave * (64u * ENUM_PER) + (uint32_t)e * 64u + lane;
        const bool valid = row < n;
        const uint32_t var = valid ? enum_variant_of(sh, t[e], nv) : ENUM_NONE;
        if (valid && var == ENUM_NONE && row < first_bad) first_bad = row;
        const bool counted = var != ENUM_NONE;
        uint64_t peers = __ballot(counted);
#pragma unroll
        for (int b = 0; b < 5; b++) {
            const uint64_t set = __ballot(counted && ((var >> b) & 1u));
            peers &= ((var >> b) & 1u) ? set : ~set;
        }
        uint32_t base = 0;
        if (counted) {
            const uint32_t leader = (uint32_t)__builtin_ctzll(peers);
            if (lane == leader) base = atomicAdd(&sh.wcnt[wave][var], (uint32_t)__popcll(peers));
            base = __shfl(base, (int)leader);
        }
        v[e] = var;
        r[e] = base + (uint32_t)__popcll(peers & below);
    }
}

// 1. tile_counts[v * ntiles + t] = rows of variant v in tile t.
__global__ __launch_bounds__(ENUM_THREADS) void enum_counts_kernel(EnumBatch batch, const uint32_t *__restrict__ variant_tags, uint32_t nv,
                                                                    uint64_t *__restrict__ tile_counts, unsigned long long *first_bad_row) {
    __shared__ EnumShared sh;
    enum_setup(sh, variant_tags, nv);
    const uint64_t T = blockIdx.x;
    uint32_t a;
    uint64_t S;
    enum_locate(batch, T, a, S);
    const zbra_enum_desc d = batch.arrays[a];
    uint64_t bad = ~0ull;
    uint32_t v[ENUM_PER], r[ENUM_PER];
    enum_wave_rows(sh, d.tags, (T - S) * ENUM_TILE, d.n, nv, v, r, bad);
    bad = wave_min_u64(bad);
    if ((threadIdx.x & 63u) == 0 && bad != ~0ull) atomicMin(first_bad_row + a, (unsigned long long)bad);
    __syncthreads();
    if (threadIdx.x < nv) {
        uint32_t c = 0;
#pragma unroll
        for (int w = 0; w < ENUM_WAVES; w++) c += sh.wcnt[w][threadIdx.x];
        tile_counts[(uint64_t)threadIdx.x * batch.ntiles + T] = c;
    }
}

// 1b. The same for at most ENUM_FEW variants, which is what enums have: counting needs neither
// the rows' order nor LDS.  A wave takes whole tiles on its own (no barrier, no set-up), a lane
// takes 32 of a tile's rows with 16-byte loads and keeps one byte counter per variant in two
// registers (at most 32 per byte); the wave adds the bytes up as 16-bit fields.  2^28 rows, 4
// variants: 530 us for the general kernel above (131072 workgroups, each a chain of set-up and
// load latencies for 8 KiB of tags), 300 us for this one.
constexpr int ENUM_COUNT_TILES = 2;  // tiles per wave
// A row whose tag no variant carries is not looked for here: it shows afterwards, as a tile whose
// counts do not add up to its rows, and only then is the tile read again for the row (the common
// case pays nothing per row for it).
__device__ __forceinline__ void enum_count_tag(uint32_t tag, const uint32_t (&vt)[ENUM_FEW], uint32_t live, uint32_t &lo, uint32_t &hi) {
#pragma unroll
    for (int b = 0; b < 4; b++) {
        const bool hit = ((live >> b) & 1u) && tag == vt[b];
        lo += hit ? 1u << (8 * b) : 0u;
    }
    if (live >> 4) {  // uniform
#pragma unroll
        for (int b = 4; b < ENUM_FEW; b++) {
            const bool hit = ((live >> b) & 1u) && tag == vt[b];
            hi += hit ? 1u << (8 * (b - 4)) : 0u;
        }
    }
}
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__global__ __launch_bounds__(ENUM_THREADS) void enum_counts_few_kernel(EnumBatch batch, const uint32_t *__restrict__ variant_tags, uint32_t nv,
                                                                        uint64_t *__restrict__ tile_counts, unsigned long long *first_bad_row) {
    const uint64_t ntiles = batch.ntiles;
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    uint32_t vt[ENUM_FEW], live = 0;  // uniform; a variant whose tag an earlier one carries never gets a row (striped.rs:349)
#pragma unroll
    for (int b = 0; b < ENUM_FEW; b++) {
        vt[b] = (uint32_t)b < nv ? variant_tags[b] : 0u;
        bool first = (uint32_t)b < nv;
        for (int a = 0; a < b; a++) first = first && vt[a] != vt[b];
        live |= first ? 1u << b : 0u;
    }
    for (int q = 0; q < ENUM_COUNT_TILES; q++) {
        const uint64_t tile = ((uint64_t)blockIdx.x * ENUM_WAVES + wave) * ENUM_COUNT_TILES + q;  // of the submission
        if (tile >= ntiles) break;  // uniform within the wave
        uint32_t a;
        uint64_t S;
        enum_locate(batch, tile, a, S);
        const zbra_enum_desc d = batch.arrays[a];
        const uint32_t *__restrict__ tags = d.tags;
        const uint64_t n = d.n;
        const bool aligned = ((uintptr_t)tags & 15u) == 0;  // tiles start at multiples of 8 KiB from `tags`
        uint64_t bad = ~0ull;
        const uint64_t r0 = (tile - S) * ENUM_TILE;
        uint32_t lo = 0, hi = 0;
        if (aligned && r0 + ENUM_TILE <= n) {
            const ZBRA_GLOBAL u32x4 *src = (const ZBRA_GLOBAL u32x4 *)(gptr<uint32_t>(tags) + r0) + lane;
            u32x4 x[ENUM_TILE / 256];
#pragma unroll
#ifdef ZBRA_ENUM_NT
            for (int st = 0; st < ENUM_TILE / 256; st++) x[st] = __builtin_nontemporal_load(src + st * 64);
#else
            for (int st = 0; st < ENUM_TILE / 256; st++) x[st] = src[st * 64];  // plain: what is read here is read again by the write pass
#endif
            // variant by variant over the lane's 32 tags: a compare and an add-with-carry per tag
            // and live variant (all tags against all variants with byte-packed counters took
            // three times the vector instructions and bound the kernel: 222 us for 2^28 rows)
#pragma unroll
            for (int b = 0; b < ENUM_FEW; b++) {
                if ((live >> b) & 1u) {  // uniform
                    uint32_t c = 0;
#pragma unroll
                    for (int st = 0; st < ENUM_TILE / 256; st++)
                        c += (uint32_t)(x[st].x == vt[b]) + (uint32_t)(x[st].y == vt[b]) + (uint32_t)(x[st].z == vt[b]) + (uint32_t)(x[st].w == vt[b]);
                    if (b < 4) lo += c << (8 * b);  // at most 32 per byte
                    else hi += c << (8 * (b - 4));
                }
            }
        } else {
            for (uint32_t st = 0; st < (uint32_t)ENUM_TILE / 64; st++) {
                const uint64_t row = r0 + st * 64u + lane;
                if (row < n) enum_count_tag(tags[row], vt, live, lo, hi);
            }
        }
        // bytes -> 16-bit fields (a wave's 2048 rows fit), summed over the lanes
        const uint32_t f02 = wave_sum_u32(lo & 0x00FF00FFu), f13 = wave_sum_u32((lo >> 8) & 0x00FF00FFu);
        const uint32_t f46 = wave_sum_u32(hi & 0x00FF00FFu), f57 = wave_sum_u32((hi >> 8) & 0x00FF00FFu);
        const uint32_t c[ENUM_FEW] = {f02 & 0xFFFFu, f13 & 0xFFFFu, f02 >> 16, f13 >> 16, f46 & 0xFFFFu, f57 & 0xFFFFu, f46 >> 16, f57 >> 16};
#pragma unroll
        for (int b = 0; b < ENUM_FEW; b++)
            if (lane == (uint32_t)b && (uint32_t)b < nv) tile_counts[(uint64_t)b * ntiles + tile] = c[b];
        uint32_t counted = 0;
#pragma unroll
        for (int b = 0; b < ENUM_FEW; b++) counted += c[b];
        const uint64_t rows = n - r0 < (uint64_t)ENUM_TILE ? n - r0 : (uint64_t)ENUM_TILE;
        if ((uint64_t)counted != rows) {  // uniform, rare: some row's tag is carried by no variant; find the first
            for (uint32_t st = 0; st < (uint32_t)ENUM_TILE / 64; st++) {
                const uint64_t row = r0 + st * 64u + lane;
                if (row < n && row < bad) {
                    const uint32_t tag = tags[row];
                    bool any = false;
#pragma unroll
                    for (int b = 0; b < ENUM_FEW; b++) any = any || (((live >> b) & 1u) && tag == vt[b]);
                    if (!any) bad = row;
                }
            }
            bad = wave_min_u64(bad);
            if (lane == 0 && bad != ~0ull) atomicMin(first_bad_row + a, (unsigned long long)bad);
        }
    }
}

// 2. Exclusive scan of every variant's tile counts, two levels, as for the offsets: blockIdx.y is
// the variant.  chunk_base holds nchunks + 1 words per variant: the second level leaves the
// variant's total over the whole submission in the last one.
__global__ __launch_bounds__(SCAN_CHUNK) void enum_scan_chunks_kernel(uint64_t ntiles, uint64_t nchunks, uint64_t *__restrict__ tile_counts,
                                                                       uint64_t *__restrict__ chunk_base) {
    __shared__ uint64_t wtot[SCAN_CHUNK / 64];
    uint64_t *col = tile_counts + (uint64_t)blockIdx.y * ntiles;
    const uint64_t t = (uint64_t)blockIdx.x * SCAN_CHUNK + threadIdx.x;
    const uint64_t x = t < ntiles ? col[t] : 0;
    uint64_t total;
    const uint64_t ex = block_exclusive_scan_1024(x, wtot, &total);
    if (t < ntiles) col[t] = ex;
    if (threadIdx.x == 0) chunk_base[(uint64_t)blockIdx.y * (nchunks + 1) + blockIdx.x] = total;
}
__global__ __launch_bounds__(SCAN_CHUNK) void enum_scan_totals_kernel(uint64_t nchunks, uint64_t *__restrict__ chunk_base) {
    __shared__ uint64_t wtot[SCAN_CHUNK / 64];
    uint64_t *col = chunk_base + (uint64_t)blockIdx.x * (nchunks + 1);  // blockIdx.x is the variant
    uint64_t carry = 0;
    for (uint64_t c0 = 0; c0 < nchunks; c0 += SCAN_CHUNK) {  // uniform trip count
        const uint64_t c = c0 + threadIdx.x;
        const uint64_t x = c < nchunks ? col[c] : 0;
        uint64_t total;
        const uint64_t ex = block_exclusive_scan_1024(x, wtot, &total);
        if (c < nchunks) col[c] = carry + ex;
        carry += total;
    }
    if (threadIdx.x == 0) col[nchunks] = carry;
}

// Rows of variant v in the submission's tiles before tile T (T == ntiles: in all of them).
__device__ __forceinline__ uint64_t enum_prefix(const uint64_t *__restrict__ tile_counts, const uint64_t *__restrict__ chunk_base, uint64_t ntiles,
                                                uint64_t nchunks, uint32_t v, uint64_t T) {
    const uint64_t *col = chunk_base + (uint64_t)v * (nchunks + 1);
    return T < ntiles ? col[T / SCAN_CHUNK] + tile_counts[(uint64_t)v * ntiles + T] : col[nchunks];
}

// 2b. counts[v] of every array: rows of variant v in its tiles.  One lane per (array, variant).
__global__ void enum_array_counts_kernel(EnumBatch batch, uint32_t nv, uint64_t nchunks, const uint64_t *__restrict__ tile_counts,
                                         const uint64_t *__restrict__ chunk_base) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (uint64_t)batch.nd * nv) return;
    const uint32_t a = (uint32_t)(i / nv), v = (uint32_t)(i % nv);
    const uint64_t S = batch.tile_start[a], E = batch.tile_start[a + 1];
    batch.arrays[a].counts[v] = E == S ? 0 : enum_prefix(tile_counts, chunk_base, batch.ntiles, nchunks, v, E) -
                                             enum_prefix(tile_counts, chunk_base, batch.ntiles, nchunks, v, S);
}

// 3. variant[i] and rank[i]: rows of the variant in earlier tiles + in earlier waves of the tile +
// earlier in the wave.  A submission with an unknown tag is reported, its outputs are not defined.
__global__ __launch_bounds__(ENUM_THREADS) void enum_write_kernel(EnumBatch batch, const uint32_t *__restrict__ variant_tags, uint32_t nv,
                                                                   uint64_t nchunks, const uint64_t *__restrict__ tile_counts,
                                                                   const uint64_t *__restrict__ chunk_base, uint32_t reverse) {
    __shared__ EnumShared sh;
    __shared__ uint64_t start[ENUM_WAVES][ENUM_MAX_VARIANTS];  // rank of the wave's first row of each variant
    enum_setup(sh, variant_tags, nv);
    // back to front when the tags are more than the memory-side cache holds: what the counting pass
    // read last is what is still there
    const uint64_t T = reverse ? gridDim.x - 1 - blockIdx.x : blockIdx.x;
    uint32_t a;
    uint64_t S;
    enum_locate(batch, T, a, S);
    const zbra_enum_desc d = batch.arrays[a];
    const uint64_t n = d.n, tile = T - S;
    uint32_t *__restrict__ variant = d.variant;
    uint64_t *__restrict__ rank = d.rank;
    uint32_t v[ENUM_PER], r[ENUM_PER];
    uint64_t bad = ~0ull;
    enum_wave_rows(sh, d.tags, tile * ENUM_TILE, n, nv, v, r, bad);
    __syncthreads();
    if (threadIdx.x < nv) {
        uint64_t run = enum_prefix(tile_counts, chunk_base, batch.ntiles, nchunks, threadIdx.x, T) -
                       enum_prefix(tile_counts, chunk_base, batch.ntiles, nchunks, threadIdx.x, S);
#pragma unroll
        for (int w = 0; w < ENUM_WAVES; w++) { start[w][threadIdx.x] = run; run += sh.wcnt[w][threadIdx.x]; }
    }
    __syncthreads();
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
#pragma unroll
    for (int e = 0; e < ENUM_PER; e++) {
        const uint64_t row = tile * ENUM_TILE + wave * (64u * ENUM_PER) + (uint32_t)e * 64u + lane;
        if (row < n) {
            __builtin_nontemporal_store(v[e], variant + row);
            __builtin_nontemporal_store(v[e] != ENUM_NONE ? start[wave][v[e]] + r[e] : 0ull, rank + row);
        }
    }
}

// Status words: ZBRA_ERR_UNKNOWN_TAG with out_count = the first offending row and midpoint = its
// tag, or ZBRA_OK with out_count = n.
// One lane per array.
__global__ void enum_report_kernel(EnumBatch batch, const unsigned long long *first_bad_row, zbra_dev_result *results) {
    const uint32_t a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= batch.nd || !results) return;
    const zbra_enum_desc d = batch.arrays[a];
    zbra_dev_result res;
    res.status = ZBRA_OK; res.width = 0; res.out_count = d.n; res.midpoint = 0; res.packed_len = 0;
    const unsigned long long bad = first_bad_row[a];
    if (bad != ~0ull) { res.status = ZBRA_ERR_UNKNOWN_TAG; res.out_count = bad; res.midpoint = (int64_t)d.tags[bad]; }
    results[a] = res;
}

// The three small steps between the counting and the writing pass as ONE launch, for submissions of
// few arrays: the scan of the chunk totals of every variant, every array's counts, the status words.
// One workgroup: the steps depend on each other only through what this workgroup itself wrote
// (a barrier orders that), and each of the three launches it replaces cost a kernel boundary.
__global__ __launch_bounds__(SCAN_CHUNK) void enum_totals_kernel(EnumBatch batch, uint32_t nv, uint64_t nchunks, const uint64_t *__restrict__ tile_counts,
                                                                 uint64_t *__restrict__ chunk_base, const unsigned long long *first_bad_row,
                                                                 zbra_dev_result *results) {
    __shared__ uint64_t wtot[SCAN_CHUNK / 64];
    if (batch.ntiles) {
        for (uint32_t v = 0; v < nv; v++) {  // as enum_scan_totals_kernel, variant after variant
            uint64_t *col = chunk_base + (uint64_t)v * (nchunks + 1);
            uint64_t carry = 0;
            for (uint64_t c0 = 0; c0 < nchunks; c0 += SCAN_CHUNK) {  // uniform trip count
                const uint64_t c = c0 + threadIdx.x;
                const uint64_t x = c < nchunks ? col[c] : 0;
                uint64_t total;
                const uint64_t ex = block_exclusive_scan_1024(x, wtot, &total);
                if (c < nchunks) col[c] = carry + ex;
                carry += total;
            }
            if (threadIdx.x == 0) col[nchunks] = carry;
        }
    }
    __syncthreads();  // the scanned columns, written by this workgroup, are read below
    for (uint64_t i = threadIdx.x; i < (uint64_t)batch.nd * nv; i += SCAN_CHUNK) {  // as enum_array_counts_kernel
        const uint32_t a = (uint32_t)(i / nv), v = (uint32_t)(i % nv);
        const uint64_t S = batch.tile_start[a], E = batch.tile_start[a + 1];
        batch.arrays[a].counts[v] = E == S ? 0 : enum_prefix(tile_counts, chunk_base, batch.ntiles, nchunks, v, E) -
                                                 enum_prefix(tile_counts, chunk_base, batch.ntiles, nchunks, v, S);
    }
    if (results) {
        for (uint32_t a = threadIdx.x; a < batch.nd; a += SCAN_CHUNK) {  // as enum_report_kernel
            const zbra_enum_desc d = batch.arrays[a];
            zbra_dev_result res;
            res.status = ZBRA_OK; res.width = 0; res.out_count = d.n; res.midpoint = 0; res.packed_len = 0;
            const unsigned long long bad = first_bad_row[a];
            if (bad != ~0ull) { res.status = ZBRA_ERR_UNKNOWN_TAG; res.out_count = bad; res.midpoint = (int64_t)d.tags[bad]; }
            results[a] = res;
        }
    }
}

// tile -> array table of a submission of several arrays.  One workgroup per array.
__global__ __launch_bounds__(256) void enum_map_kernel(EnumBatch batch, uint32_t *__restrict__ tile_map) {
    const uint32_t a = blockIdx.x;
    if (a >= batch.nd) return;
    for (uint64_t t = batch.tile_start[a] + threadIdx.x; t < batch.tile_start[a + 1]; t += blockDim.x) tile_map[t] = a;
}

}  // namespace zbra
