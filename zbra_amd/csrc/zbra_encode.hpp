// Encode kernels: exact median by radix select -> width -> fused (v - mid) -> zig-zag -> bit-pack.
//
// Replaces compress_int_array (zbra-core/src/compression.rs:271-292) and its stages
// frame_of_reference_encode (:52-74, clone + full sort for the median), zig_zag_encode (:88-98)
// and bp64_pack (:117-178, max scan + byte-at-a-time bit buffer).
//
// The bytes are a function of (midpoint, width) only, so the device needs, per array:
//   1. min / max, and the (n-1)/2-th smallest value a (for even n also the next one, b), exactly;
//   2. midpoint = a, or a + trunc((b - a)/2)        (compression.rs:60-68)
//   3. width = bit length of max zigzag(v - mid)    from the two extremes when span < 2^63 (no
//                                                   wrap-around possible), else from an OR pass
//   4. the packed stream.
// Selection reads the array ONCE in the common case:
//   sample   4096 values at hashed positions (one per n/4096 stride) stay in the registers of one
//            workgroup, which finds the two values 128 sample ranks (4 sigma) either side of the
//            median's expected sample rank by a two-rank radix select: they bracket the median.
//            Arrays of up to 4096 values are selected whole: a, b, min and max fall out directly.
//   bracket  one pass counts the values below the bracket, finds min / max and keeps what is
//            inside the bracket: as a histogram when it spans fewer than 4096 distinct keys (one or
//            two counters for a bracket of one or two keys), else as a compacted list (about 6 %
//            of the array) together with a histogram of the keys' first digit.
//   verify   the exact counts say whether ranks (n-1)/2 and n/2 are inside the bracket.  If so the
//            radix select below runs on the histogram / the compacted list; if not (a sample that
//            misled, more than n/4 candidates) it runs on the whole array, as it would without
//            the sample.  Either way the result is exact.
//   select   radix select on key = v - base, 12 bits a digit, from the top.  A short source is
//            finished by one workgroup in one launch; a long one gets up to three launches of many
//            workgroups, each of which either counts the next digit or, as soon as at most 2048
//            keys share the resolved digits, gathers those keys for the finishing workgroup.
// Every kernel works on a whole submission (many arrays) through the same tile -> array map the
// decoder uses, and every pass that an array does not need skips that array's memory.
#pragma once
#include "zbra_common.hpp"

namespace zbra {

constexpr int DIGIT_BITS = 12;
constexpr int BINS = 1 << DIGIT_BITS;
constexpr int MAX_PASSES = 6;  // ceil(64 / 12)
constexpr uint64_t SIGN = 0x8000000000000000ull;
constexpr uint32_t WIDTH_SKIP = 0xFFFFFFFFu;  // array rejected (capacity): pack nothing

// One array of an encode submission: inputs (filled by the host) and select state (device).
struct EncodeJob {
    const void *values;  // n elements of `kind`
    uint8_t *out;        // E goes here
    uint64_t n;
    uint64_t cap;
    uint32_t kind;
    uint32_t passes;     // digits in a key of the radix select: ceil(key_bits / DIGIT_BITS)
    uint64_t min_b;      // min / max of (v ^ SIGN): unsigned order of the biased value = signed order
    uint64_t max_b;
    uint64_t next_above; // smallest key above the bracket (in key space once the bracket is verified), ~0 if none
    unsigned long long above_key;  // smallest key above the keys that share the resolved digits, from the select's last read of the source
    uint64_t or_bits;    // OR of zigzag(v - mid), wide spans only
    int64_t midpoint;
    uint32_t width;
    uint32_t need_or;
    // bracket select
    uint32_t mode;       // SelectMode
    uint32_t frame_dropped;  // framed output: the array's frame did not fit (set by the layout kernel)
    uint64_t lo_b, hi_b; // bracket, inclusive, in biased (v ^ SIGN) space
    uint64_t below, inb; // exact counts of values below / inside the bracket (summed by the setup kernel)
    uint64_t key_base;   // keys of the radix select are (v ^ SIGN) - key_base
    const void *sel_values;  // what the radix select reads: the array itself or its candidates
    uint64_t sel_n;
    uint32_t sel_kind;
    uint32_t key_bits;   // bit length of the largest key of the radix select; digit i is bits [digit_shift(i), +DIGIT_BITS)
    // Candidate list (values inside the bracket): CAND_SEGS segments of seg_cap entries, segment s
    // filled from its start with counters->slot[s].cand_count entries.  Appends rotate over the
    // segments, so that they (and the below / inside counts, which live in the same slots) queue
    // on CAND_SEGS different cache lines instead of one.
    int64_t *cand;
    uint64_t seg_cap;    // a multiple of TILE; 0 = no list (short array or bracket select off)
    struct BracketCounters *counters;
    // Radix select.  What is read: the candidate list (mode SEL_BRACKET_COMPACT) or the array itself,
    // of which only values with flo <= (v ^ SIGN) <= fhi count (the verified bracket whose list
    // overflowed, or everything).
    uint64_t flo, fhi;
    uint32_t *hist;      // BINS counters: the bracket pass's histogram (exact keys of a narrow bracket, first digit of a wide one)
    uint32_t *lvl_hist;  // SEL_LEVELS x BINS counters of the many-workgroup levels (long arrays only)
    // Keys gathered by a level for the finishing workgroup (long arrays only): GATHER_SEGS segments of
    // GATHER_CAP keys, each with its own fill count (the level's workgroups take their room with a
    // returning atomic, and hundreds of them on one word queue: 7 of that level's 25 us)
    uint64_t *small;
    uint32_t small_n[8];
    uint32_t pending;    // 1 = on the list of the many-workgroup levels
    uint32_t pad_small;
    // smallest / largest key among those whose next digit level L counted (all equal: a run of one
    // value, which needs no further digit)
    unsigned long long match_min[3], match_max[3];
    struct SelState {
        uint64_t prefix;  // resolved high digits of the key of a
        uint64_t k;       // rank of a among the keys that share `prefix`
        uint64_t cnt;     // how many keys share `prefix`
        uint32_t lv;      // digits resolved
        uint32_t action;  // SelAction: what the launch that wrote this state left behind
    } S[4];               // S[0]: after set-up; S[L]: after level L (SEL_LEVELS of them)
};
constexpr int SEL_LEVELS = 3;
static_assert(SEL_LEVELS == sizeof(EncodeJob::match_min) / sizeof(unsigned long long), "one min / max pair per level");
constexpr uint32_t GATHER_CAP = 2048;  // keys a level gathers: what the finishing workgroup holds in LDS
constexpr uint32_t GATHER_SEGS = 8;   // segments of the gather buffer, one fill count each
static_assert(GATHER_SEGS == sizeof(EncodeJob::small_n) / sizeof(uint32_t), "one count per segment");
enum SelAction : uint32_t {
    ACT_NONE = 0,    // nothing counted yet: the next step starts from this state
    ACT_HIST = 1,    // digit `lv` of the keys that share `prefix` is counted: in `hist` for S[0], else in lvl_hist[L - 1]
    ACT_GATHER = 2,  // the `cnt` keys that share `prefix` are in `small`
    ACT_DONE = 3     // the midpoint is known
};

constexpr int CAND_SEGS = 16;
struct BracketCounters {
    struct Slot {
        unsigned long long below, inb, cand_count;  // cand_count may exceed seg_cap: the list is then unusable
        unsigned long long first;                    // two-key bracket: values equal to its lower key
        unsigned long long pad[4];
    } slot[CAND_SEGS];
};
static_assert(sizeof(BracketCounters::Slot) == 64, "one slot per 64-byte line");

enum SelectMode : uint32_t {
    SEL_FULL = 0,            // no bracket: min/max pass, then radix select over the whole array
    SEL_BRACKET_HIST = 1,    // bracket narrower than BINS keys: histogram filled by the bracket pass
    SEL_BRACKET_COMPACT = 2, // wide bracket: candidates compacted by the bracket pass
    SEL_DONE = 3             // midpoint and width already known (small array sorted whole, constant, ...)
};

constexpr uint32_t MICRO_KEYS = 8;     // brackets of fewer keys than this are counted in per-lane columns (see SEL_BRACKET_MICRO)
#ifndef ZBRA_SAMPLE  // tuning builds: -DZBRA_SAMPLE=16384 -DZBRA_SAMPLE_MARGIN=256 -DZBRA_SAMPLE_WAVES=4
#define ZBRA_SAMPLE 4096
#define ZBRA_SAMPLE_MARGIN 128
#define ZBRA_SAMPLE_WAVES 8
#endif
constexpr int SAMPLE = ZBRA_SAMPLE;   // sample keys of an array (four per lane of the sampling workgroup)
constexpr int SAMPLE_THREADS = 1024;  // lanes of the sampling workgroup
constexpr int SAMPLE_MARGIN = ZBRA_SAMPLE_MARGIN;  // sample ranks either side: 4 sigma, sigma = sqrt(SAMPLE)/2 = 32
constexpr int CAND_DIV = 4;           // candidate list capacity = n / CAND_DIV (a bracket holds ~6 % of n)
#ifndef ZBRA_CAND_LDS
#define ZBRA_CAND_LDS 2048
#endif
constexpr int CAND_LDS = ZBRA_CAND_LDS;  // candidates a workgroup stages in LDS between flushes (a quarter of them per wave)
constexpr int CAND_PARTS = CAND_LDS >= 2048 ? 2 : 4;  // a tile's values are appended in this many parts, with a look at the room before each

__device__ __forceinline__ int64_t load_value(const EncodeJob &j, uint64_t i) {
    return j.kind != ZBRA_ELEM_U32 ? gptr<int64_t>(j.values)[i] : (int64_t)gptr<uint32_t>(j.values)[i];
}

// Value that goes into the bit stream: zigzag(v - mid), or v itself for bp64_pack alone.
__device__ __forceinline__ uint64_t to_stream(const EncodeJob &j, int64_t x, uint64_t mid) {
    return j.kind == KIND_U64_PLAIN ? (uint64_t)x : zigzag_encode64((int64_t)((uint64_t)x - mid));
}

constexpr int PAIRS = TILE / (THREADS * 2);  // 16-B loads of int64 per thread per tile

// Which value of its tile a lane holds in x[e]: every load is 16 B per lane, so int64 tiles are
// dealt in pairs (x[2*it], x[2*it+1] = values it*2*THREADS + 2*tid, +1) and u32 tiles in fours
// (x[4*q .. 4*q+3] = values q*4*THREADS + 4*tid .. +3).
__device__ __forceinline__ uint32_t tile_slot(uint32_t kind, int e) {
    return kind != ZBRA_ELEM_U32 ? (uint32_t)(e >> 1) * THREADS * 2 + threadIdx.x * 2 + (uint32_t)(e & 1)
                                 : (uint32_t)(e >> 2) * THREADS * 4 + threadIdx.x * 4 + (uint32_t)(e & 3);
}

// This thread's 2*PAIRS values of the tile that starts at value v0, dealt as tile_slot says.  A
// tile that lies wholly inside the array issues all its loads back to back; values past n get
// `fill`.
__device__ __forceinline__ void load_tile_from(const void *values, uint32_t kind, uint64_t v0, uint64_t n, int64_t fill,
                                               int64_t (&x)[2 * PAIRS]) {
    const uint32_t tid = threadIdx.x;
    if (v0 + TILE <= n) {  // uniform
        if (kind != ZBRA_ELEM_U32) {
            const ZBRA_GLOBAL u64x2 *src = (const ZBRA_GLOBAL u64x2 *)(gptr<int64_t>(values) + v0) + tid;
            u64x2 v[PAIRS];
#pragma unroll
            for (int it = 0; it < PAIRS; ++it) v[it] = __builtin_nontemporal_load(src + it * THREADS);
#pragma unroll
            for (int it = 0; it < PAIRS; ++it) { x[2 * it] = (int64_t)v[it].x; x[2 * it + 1] = (int64_t)v[it].y; }
        } else {
            static_assert(PAIRS % 2 == 0, "u32 tiles are dealt in fours");
            const ZBRA_GLOBAL u32x4 *src = (const ZBRA_GLOBAL u32x4 *)(gptr<uint32_t>(values) + v0) + tid;
            u32x4 v[PAIRS / 2];
#pragma unroll
            for (int q = 0; q < PAIRS / 2; ++q) v[q] = __builtin_nontemporal_load(src + q * THREADS);
#pragma unroll
            for (int q = 0; q < PAIRS / 2; ++q) {
                x[4 * q] = (int64_t)v[q].x; x[4 * q + 1] = (int64_t)v[q].y; x[4 * q + 2] = (int64_t)v[q].z; x[4 * q + 3] = (int64_t)v[q].w;
            }
        }
    } else {
#pragma unroll
        for (int e = 0; e < 2 * PAIRS; ++e) {
            const uint64_t i = v0 + tile_slot(kind, e);
            x[e] = fill;
            if (i < n) x[e] = kind != ZBRA_ELEM_U32 ? gptr<int64_t>(values)[i] : (int64_t)gptr<uint32_t>(values)[i];
        }
    }
}
__device__ __forceinline__ void load_tile(const EncodeJob &j, uint64_t v0, uint64_t n, int64_t fill, int64_t (&x)[2 * PAIRS]) {
    load_tile_from(j.values, j.kind, v0, n, fill, x);
}

__device__ __forceinline__ uint64_t wave_min_u64(uint64_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { uint64_t t = __shfl_xor(v, o); v = t < v ? t : v; }
    return v;
}
__device__ __forceinline__ uint64_t wave_max_u64(uint64_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { uint64_t t = __shfl_xor(v, o); v = t > v ? t : v; }
    return v;
}
__device__ __forceinline__ uint64_t wave_or_u64(uint64_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v |= __shfl_xor(v, o);
    return v;
}

// Walks the contiguous tile range of this workgroup array by array.  F(a, job, first_tile_of_a,
// t0, t1) is called once per array with tiles [t0, t1) of the submission.
template <typename F>
__device__ __forceinline__ void for_each_array_run(const uint64_t *__restrict__ tile_start, uint32_t nd, uint64_t ntiles,
                                                   F &&f) {
    const uint64_t tpw = (ntiles + gridDim.x - 1) / gridDim.x;
    uint64_t t = (uint64_t)blockIdx.x * tpw;
    uint64_t tend = t + tpw < ntiles ? t + tpw : ntiles;
    if (t >= tend) return;
    uint32_t a = nd > 1 ? find_array(tile_start, nd, t) : 0;
    while (t < tend) {
        const uint64_t a_first = tile_start[a];
        const uint64_t a_last = tile_start[a + 1];
        const uint64_t t1 = a_last < tend ? a_last : tend;
        if (t1 > t) f(a, a_first, t, t1);
        t = t1;
        a++;
    }
}

// ---- helpers ---------------------------------------------------------------------------------

__device__ __forceinline__ uint32_t bit_length64(uint64_t x) { return x ? 64u - (uint32_t)__builtin_clzll(x) : 0u; }

// Digits of a select key of `bits` bits are taken from the top: digit i starts at this bit, so the
// first digit always spans a full DIGIT_BITS bits of the keys' range (never a few leftover top
// bits) and only the last one may be narrower: it then overlaps bits of the digit before it, which
// are the same for all keys that share the earlier digits.
__device__ __forceinline__ uint32_t digit_shift(uint32_t bits, uint32_t i) {
    const uint32_t top = DIGIT_BITS * (i + 1);
    return bits > top ? bits - top : 0u;
}

// bp64_pack's width rule (compression.rs:123-128) from max zig-zag value.
__device__ __forceinline__ uint32_t width_of(uint64_t zmax) { return zmax ? bit_length64(zmax) : 1u; }

// Needs j.min_b / j.max_b / j.need_or.
__device__ __forceinline__ void set_midpoint(EncodeJob &j, int64_t a, int64_t b, bool even) {
    int64_t mid = a;
    if (even) {
        const int64_t diff = (int64_t)((uint64_t)b - (uint64_t)a);  // b.wrapping_sub(a)
        mid = (int64_t)((uint64_t)a + (uint64_t)(diff / 2));        // a.wrapping_add(diff / 2), `/` truncates
    }
    j.midpoint = mid;
    const int64_t vmin = (int64_t)(j.min_b ^ SIGN), vmax = (int64_t)(j.max_b ^ SIGN);
    const uint64_t z0 = zigzag_encode64((int64_t)((uint64_t)vmin - (uint64_t)mid));
    const uint64_t z1 = zigzag_encode64((int64_t)((uint64_t)vmax - (uint64_t)mid));
    const uint64_t z = z0 > z1 ? z0 : z1;
    if (!j.need_or) {
        j.width = width_of(z);
    } else if (z >= SIGN) {
        // A span of 2^63 or more lets v - mid wrap, so the extremes need not give the largest
        // zig-zag value; but they are values of the array, and when one of them already needs all
        // 64 bits nothing can need more: the OR pass over the array is not needed (a full-range
        // column, f64 bit patterns).  [i64::MIN, i64::MAX] is the case that still takes it.
        j.width = 64;
        j.need_or = 0;
    }
}

// ---- 1. sample: bracket the median (or finish small arrays outright) --------------------------

// Two order statistics of up to SAMPLE keys, by one workgroup of SAMPLE_THREADS lanes that keep
// their keys in registers: radix select on (key - smallest key), DIGIT_BITS bits a pass from the
// top, both ranks at once (two histograms in LDS, one scan).  ceil(bits(largest - smallest) / 12)
// passes of about 2 us, against 31 us for sorting 4096 keys in LDS, of which only two were looked at.
constexpr int SAMPLE_PER = SAMPLE / SAMPLE_THREADS;  // keys per lane
struct SampleShared {
    uint32_t hist[2][BINS];
    uint64_t wave_lo[SAMPLE_THREADS / 64], wave_hi[SAMPLE_THREADS / 64];
    uint32_t wave_sum[2][SAMPLE_THREADS / 64];
    uint32_t bin[2], before[2], cnt[2];
    uint64_t found[2];
};
// key[i] valid for i < nvalid (this lane's keys); ranks r0 <= r1 < total valid keys.  Returns the
// keys of rank r0 and r1 in out[0], out[1], the smallest and largest key in out[2], out[3].  Every
// lane gets the same answer.
__device__ __forceinline__ void select_two_ranks(const uint64_t (&key)[SAMPLE_PER], uint32_t nvalid, uint32_t r0, uint32_t r1,
                                                 SampleShared *sh, uint64_t (&out)[4]) {
    const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63u;
    constexpr uint32_t WAVES = SAMPLE_THREADS / 64;
    uint64_t lo = ~0ull, hi = 0;
#pragma unroll
    for (int i = 0; i < SAMPLE_PER; i++)
        if ((uint32_t)i < nvalid) { lo = min(lo, key[i]); hi = max(hi, key[i]); }
    lo = wave_min_u64(lo); hi = wave_max_u64(hi);
    if (lane == 0) { sh->wave_lo[wave] = lo; sh->wave_hi[wave] = hi; }
    __syncthreads();
    for (uint32_t w = 0; w < WAVES; w++) { lo = min(lo, sh->wave_lo[w]); hi = max(hi, sh->wave_hi[w]); }
    out[2] = lo; out[3] = hi;
    const uint32_t bits = bit_length64(hi - lo), passes = (bits + DIGIT_BITS - 1) / DIGIT_BITS;
    uint64_t prefix[2] = {0, 0};
    uint32_t k[2] = {r0, r1};
    for (uint32_t lv = 0; lv < passes; lv++) {  // uniform
        for (uint32_t b = tid; b < 2u * BINS; b += SAMPLE_THREADS) (&sh->hist[0][0])[b] = 0;
        __syncthreads();
        const uint32_t shift = digit_shift(bits, lv), up = lv ? digit_shift(bits, lv - 1) : 0u;
#pragma unroll
        for (int i = 0; i < SAMPLE_PER; i++) {
            if ((uint32_t)i < nvalid) {
                const uint64_t rel = key[i] - lo;
                const uint32_t digit = (uint32_t)(rel >> shift) & (BINS - 1);
                if (lv == 0 || (rel >> up) == prefix[0]) atomicAdd(&sh->hist[0][digit], 1u);
                if (lv == 0 || (rel >> up) == prefix[1]) atomicAdd(&sh->hist[1][digit], 1u);
            }
        }
        __syncthreads();
        // lane t brings bins [4t, 4t + 4) of both histograms; exclusive scan over the workgroup
        uint32_t c[2][BINS / SAMPLE_THREADS], sum[2], incl[2];
#pragma unroll
        for (int h = 0; h < 2; h++) {
            sum[h] = 0;
#pragma unroll
            for (int q = 0; q < BINS / SAMPLE_THREADS; q++) { c[h][q] = sh->hist[h][tid * (BINS / SAMPLE_THREADS) + q]; sum[h] += c[h][q]; }
            incl[h] = sum[h];
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const uint32_t t = __shfl_up(incl[h], o); if (lane >= (uint32_t)o) incl[h] += t; }
            if (lane == 63) sh->wave_sum[h][wave] = incl[h];
        }
        __syncthreads();
#pragma unroll
        for (int h = 0; h < 2; h++) {
            uint32_t before = incl[h] - sum[h];
            for (uint32_t w = 0; w < wave; w++) before += sh->wave_sum[h][w];
            if (k[h] >= before && k[h] < before + sum[h]) {  // exactly one lane
                uint32_t run = before;
#pragma unroll
                for (int q = 0; q < BINS / SAMPLE_THREADS; q++) {
                    if (k[h] >= run && k[h] < run + c[h][q]) { sh->bin[h] = tid * (BINS / SAMPLE_THREADS) + q; sh->before[h] = run; sh->cnt[h] = c[h][q]; }
                    run += c[h][q];
                }
            }
        }
        __syncthreads();
        const uint32_t d = lv == 0 ? (uint32_t)DIGIT_BITS : up - shift;
#pragma unroll
        for (int h = 0; h < 2; h++) {
            prefix[h] = (lv == 0 ? 0 : prefix[h] << d) | (sh->bin[h] & ((1u << d) - 1u));
            k[h] -= sh->before[h];
        }
        // Keys spread over a wide range are alone in their bin after a pass or two (4096 keys in
        // 4096 bins): the one key that carries each prefix is the answer, its lower digits need
        // no passes of their own.
        const bool alone = sh->cnt[0] == 1 && sh->cnt[1] == 1 && lv + 1 < passes;  // uniform
        if (alone) {
#pragma unroll
            for (int i = 0; i < SAMPLE_PER; i++) {
                if ((uint32_t)i < nvalid) {
                    const uint64_t rel = key[i] - lo;
                    if ((rel >> shift) == prefix[0]) sh->found[0] = rel;
                    if ((rel >> shift) == prefix[1]) sh->found[1] = rel;
                }
            }
        }
        // the next pass's zeroing of the histograms comes after every lane has read bin / before:
        // the barrier at the top of its counting loop is behind the zeroing, so add one here
        __syncthreads();
        if (alone) { prefix[0] = sh->found[0]; prefix[1] = sh->found[1]; break; }
    }
    out[0] = lo + prefix[0];
    out[1] = lo + prefix[1];
}

// One workgroup per array.  Held at 64 VGPRs (it wants 72 and spills 4) so that two workgroups of 1024
// lanes fit a CU: three thousand 2^14-row arrays 121 -> 98 us, a single array the same 14 us.
__global__ __launch_bounds__(SAMPLE_THREADS, ZBRA_SAMPLE_WAVES) void enc_sample_kernel(EncodeJob *__restrict__ jobs, uint32_t nd, uint32_t use_bracket) {
    __shared__ SampleShared sh;
    const uint32_t a = blockIdx.x;
    if (a >= nd) return;
    EncodeJob &job = jobs[a];
    const uint64_t n = job.n;
    if (n == 0 || job.kind == KIND_U64_PLAIN || !use_bracket) return;  // mode stays SEL_FULL; setup deals with it

    const bool whole = n <= (uint64_t)SAMPLE;
    // An array too short for a candidate list (seg_cap == 0) only asks the sample whether its
    // bracket is narrow enough for the histogram route: a quarter of the keys answers that.
    // The sample of a longer array is a power of two, and counts stay below 2^32 (the frame's u32):
    // strides, offsets and the expected rank below need no 128-bit product and one 64-bit division.
    static_assert((SAMPLE & (SAMPLE - 1)) == 0 && SAMPLE >= 4, "power-of-two sample");
    const uint32_t sample_log2 = (uint32_t)__builtin_ctz((unsigned)SAMPLE) - (job.seg_cap ? 0u : 2u);
    const uint32_t sample = whole ? (uint32_t)n : 1u << sample_log2;
    const uint32_t margin = job.seg_cap ? (uint32_t)SAMPLE_MARGIN : (uint32_t)SAMPLE_MARGIN / 2;  // 4 sigma of sqrt(sample) / 2
    const uint64_t stride = whole ? 1 : n >> sample_log2;  // below 2^22
    uint64_t key[SAMPLE_PER];
    uint32_t nvalid = 0;  // this lane holds keys j = tid + i * SAMPLE_THREADS, i < nvalid
#pragma unroll
    for (int i = 0; i < SAMPLE_PER; i++) {
        const uint32_t j = threadIdx.x + (uint32_t)i * SAMPLE_THREADS;
        key[i] = 0;
        if (j < sample) {
            uint64_t at = j;
            if (!whole) {
                // one value from each of `sample` equal strides, at a hashed offset inside it, so that
                // periodic data does not alias with the sampling step; the offset is the hash's upper
                // half scaled to the stride (a multiply and a shift: no division)
                uint64_t h = (j + 1) * 0x9E3779B97F4A7C15ull;
                h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
                at = (uint64_t)j * stride + (((h >> 32) * stride) >> 32);
            }
            key[i] = (uint64_t)load_value(job, at) ^ SIGN;
            nvalid = (uint32_t)i + 1;
        }
    }
    const uint64_t k = (n - 1) / 2;  // rank of a: n/2 - 1 for even n, n/2 for odd n
    uint64_t got[4];
    if (whole) {  // every value is in the sample: a, b, min and max fall out directly
        const bool even = (n & 1) == 0;
        select_two_ranks(key, nvalid, (uint32_t)k, (uint32_t)(even ? k + 1 : k), &sh, got);
        if (threadIdx.x != 0) return;
        job.min_b = got[2]; job.max_b = got[3];
        job.need_or = job.max_b - job.min_b >= SIGN;
        job.passes = 0;
        job.mode = SEL_DONE;
        set_midpoint(job, (int64_t)(got[0] ^ SIGN), (int64_t)(got[1] ^ SIGN), even);
        return;
    }
    const uint64_t r = (k << sample_log2) / n;  // expected sample rank of a (k < 2^31)
    const bool has_lo = r >= (uint64_t)margin, has_hi = r + margin < (uint64_t)sample;
    select_two_ranks(key, nvalid, has_lo ? (uint32_t)(r - margin) : 0u, has_hi ? (uint32_t)(r + margin) : sample - 1, &sh, got);
    const uint64_t lo_b = has_lo ? got[0] : 0, hi_b = has_hi ? got[1] : ~0ull;
    if (threadIdx.x != 0) return;
    job.min_b = got[2]; job.max_b = got[3];  // of the sample: where the bracket pass's minimum and maximum start
    job.lo_b = lo_b;
    job.hi_b = hi_b;
    job.key_base = job.lo_b;  // keys of a bracket select: (v ^ SIGN) - lo_b
    job.key_bits = bit_length64(job.hi_b - job.lo_b);
    job.passes = (job.key_bits + DIGIT_BITS - 1) / DIGIT_BITS;
    if (job.hi_b - job.lo_b < (uint64_t)BINS) { job.mode = SEL_BRACKET_HIST; job.passes = 1; }
    else if (job.seg_cap) job.mode = SEL_BRACKET_COMPACT;  // else stays SEL_FULL: no room for candidates
}

// ---- 2. bracket pass: min / max, counts, histogram or candidates — one read of the array ------

// Per-lane accumulators of one run of tiles of one array.
struct BracketAcc {
    uint64_t lo = ~0ull, hi = 0, above = ~0ull;
    uint32_t below = 0, inb = 0;
    uint32_t first = 0;  // two-key bracket: values equal to its lower key
};
// Brackets of at most MICRO_KEYS keys (enum tags, booleans): a shared LDS histogram would take
// every lane's atomic on the same few addresses, which serialise.  Each lane counts in its own
// column of the pool instead (key k of lane t at lh[k * THREADS + t]: no two lanes ever meet), and
// the columns are summed when the run ends: per wave, then per workgroup in LDS, so that each
// workgroup sends ONE atomic per key to the array's histogram (atomics on one word from all over
// the chip queue behind each other: one per wave made this pass twice as long).
static_assert(MICRO_KEYS * THREADS + MICRO_KEYS <= (uint32_t)BINS, "the private columns and their sums live in the histogram pool");
constexpr uint32_t SEL_BRACKET_MICRO = 100;  // template tag only: a histogram-mode array with a tiny bracket
// A bracket of ONE key (more than 6 % of the values are the median: a popular tag, a boolean, a
// default value) needs no histogram at all, its one bin is the count of values inside; a bracket
// of TWO keys needs one more count.  These are the usual tiny brackets, and without the per-value
// LDS access the pass runs at the rate of the plain histogram route.
constexpr uint32_t SEL_BRACKET_ONE = 101, SEL_BRACKET_TWO = 102;  // template tags only

// Values x[E0..E1) of a tile.  MODE and WHOLE (the tile lies wholly inside the array) are
// compile-time so that the loop body is straight-line code with one branch (the append).
//
// Minimum and maximum: a 64-bit min / max is a compare and two selects each, per value, although
// after a lane's first few values hardly any value improves on what it holds (it starts from the
// sample's extremes).  So the wave first asks whether ANY lane has such a value (two compares and
// a scalar branch) and only then updates.
template <uint32_t MODE, bool WHOLE, int E0, int E1>
__device__ __forceinline__ void bracket_values(const int64_t (&x)[2 * PAIRS], uint32_t kind, uint64_t v0, uint64_t n, uint64_t lo_b,
                                               uint64_t hi_b, BracketAcc &acc, uint32_t *lh, int64_t *lcand, uint32_t *lcount,
                                               uint32_t digit_shift = 0, uint32_t *wave_cnt = nullptr) {
    // lanes past the end hold a value of the array: harmless here
    uint64_t improves = 0;  // uniform: lanes with a value beyond what they hold
#pragma unroll
    for (int e = E0; e < E1; ++e) {
        const uint64_t b = (uint64_t)x[e] ^ SIGN;
        improves |= __builtin_amdgcn_ballot_w64(b < acc.lo) | __builtin_amdgcn_ballot_w64(b > acc.hi);
    }
    if (improves != 0) {
        asm volatile("" ::: "memory");  // keeps this a branch: as straight-line selects it costs more than it saves
#pragma unroll
        for (int e = E0; e < E1; ++e) {
            const uint64_t b = (uint64_t)x[e] ^ SIGN;
            acc.lo = min(acc.lo, b);
            acc.hi = max(acc.hi, b);
        }
    }
#pragma unroll
    for (int e = E0; e < E1; ++e) {
        const uint64_t b = (uint64_t)x[e] ^ SIGN;
        if (MODE == SEL_BRACKET_COMPACT) {
            // Wide bracket.  Counts are kept per WAVE, from the ballots (scalar unit), and the
            // smallest value above the bracket is not looked for: an upper median that is not
            // inside the bracket sends the array to the select over all its values.
            const bool valid = WHOLE || v0 + tile_slot(kind, e) < n;
            const uint64_t r = b - lo_b;
            const bool inside = valid && r <= hi_b - lo_b;
            const uint64_t m = __builtin_amdgcn_ballot_w64(inside);
            wave_cnt[0] += (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(valid && b < lo_b));
            wave_cnt[1] += (uint32_t)__popcll(m);
            if (inside) {
                // wave-private staging: the ballot gives every lane its slot, no atomic, no barrier
                lcand[*lcount + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))] = x[e];
                atomicAdd(&lh[(uint32_t)(r >> digit_shift)], 1u);  // first digit of the select's key
            }
            *lcount += (uint32_t)__popcll(m);
        } else if (MODE != SEL_FULL) {
            const bool valid = WHOLE || v0 + tile_slot(kind, e) < n;
            const bool is_below = valid && b < lo_b, is_above = valid && b > hi_b;
            const bool inside = valid && !is_below && !is_above;
            acc.below += is_below;
            acc.inb += inside;
            acc.above = is_above && b < acc.above ? b : acc.above;
            if (MODE == SEL_BRACKET_ONE) {
                // nothing: the bin is acc.inb
            } else if (MODE == SEL_BRACKET_TWO) {
                acc.first += inside && b == lo_b;
            } else if (MODE == SEL_BRACKET_MICRO) {
                if (inside) lh[((uint32_t)(b - lo_b) & (MICRO_KEYS - 1)) * THREADS + threadIdx.x] += 1u;
            } else if (inside) {
                atomicAdd(&lh[(uint32_t)(b - lo_b)], 1u);
            }
        }
    }
}

#ifndef ZBRA_BRACKET_WAVES
#define ZBRA_BRACKET_WAVES 4
#endif
__global__ __launch_bounds__(THREADS, ZBRA_BRACKET_WAVES) void enc_bracket_kernel(EncodeJob *__restrict__ jobs,
                                                               const uint64_t *__restrict__ tile_start, uint32_t nd,
                                                               uint64_t ntiles) {
    // Histogram (exact keys of a narrow bracket; first digit of the keys of a wide one, whose
    // values are also staged as candidates).  Invariant between runs: the histogram is all zero.
    __shared__ uint32_t lh[BINS];
    __shared__ __attribute__((aligned(16))) int64_t lcand[CAND_LDS];
    __shared__ unsigned long long lbelow, linb; // this run's counts
    __shared__ uint32_t lfirst;                 // two-key bracket: this run's values equal to the lower key
    for (uint32_t b = threadIdx.x; b < BINS; b += THREADS) lh[b] = 0;
    if (threadIdx.x == 0) { lbelow = 0; linb = 0; lfirst = 0; }
    __syncthreads();
    // Tiles t0, t0 + step, ... below t1 of array a (whose first tile is a_first).
    auto run = [&](uint32_t a, uint64_t a_first, uint64_t t0, uint64_t t1, uint64_t step) {
        const EncodeJob &job = jobs[a];
        if (job.kind == KIND_U64_PLAIN || job.mode == SEL_DONE) return;  // uniform
        // a histogram-mode array with a tiny bracket counts in per-lane columns (see MICRO_KEYS)
        const uint64_t bspan = job.hi_b - job.lo_b;
        const uint32_t mode = job.mode != SEL_BRACKET_HIST || bspan >= (uint64_t)MICRO_KEYS ? job.mode
                              : bspan == 0 ? SEL_BRACKET_ONE : bspan == 1 ? SEL_BRACKET_TWO : SEL_BRACKET_MICRO;
        const void *values = job.values;
        const uint32_t kind = job.kind;
        const uint64_t n = job.n, seg_cap = job.seg_cap;
        // counts go to the slot of this workgroup; candidate flushes rotate over the segments, so
        // that a short array that only two or three workgroups see still fills all of them evenly
        BracketCounters::Slot *ctr = &job.counters->slot[blockIdx.x % CAND_SEGS];
        uint32_t flushes = 0;
        const uint64_t lo_b = job.lo_b, hi_b = job.hi_b;
        const int64_t fill = load_value(job, 0);  // a value of the array: a no-op for min and max
        BracketAcc acc;
        // the sample's extremes, or what other workgroups have found since: values of the array
        acc.lo = __hip_atomic_load(&job.min_b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        acc.hi = __hip_atomic_load(&job.max_b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // Candidates are staged per WAVE: a quarter of the pool each, filled in ballot order and
        // flushed (one returning atomic on the segment's counter, then a copy by the wave's 64
        // lanes) when less than half a tile's worth of room is left.  No lane of another wave is
        // involved, so the tile loop has no workgroup barrier.
        constexpr uint32_t WAVES = THREADS / 64, WAVE_CAP = CAND_LDS / WAVES;
        static_assert(WAVE_CAP >= 2 * (TILE / CAND_PARTS / WAVES), "a wave appends at most TILE / CAND_PARTS / WAVES values between two looks");
        static_assert(PAIRS == 4, "a tile is handled in halves or quarters of its eight values per lane");
        const uint32_t dsh = digit_shift(job.key_bits, 0);  // wide bracket: the key's first digit
        const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
        int64_t *wstage = lcand + wave * WAVE_CAP;
        uint32_t wcount = 0;  // uniform within the wave
        uint32_t wcnt[2] = {0, 0};  // wide bracket: the wave's values below / inside the bracket
        auto flush_wave = [&]() {
            if (wcount == 0) return;
            const uint32_t sgm = (blockIdx.x * WAVES + wave + flushes++) % CAND_SEGS;
            unsigned long long base = 0;
            if (lane == 0) base = atomicAdd(&job.counters->slot[sgm].cand_count, (unsigned long long)wcount);
            base = __shfl(base, 0);
            // the wave's own LDS writes, read back by other lanes of the same wave
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            if (base + wcount <= seg_cap) {
                ZBRA_GLOBAL int64_t *dst = gptr_mut<int64_t>(job.cand) + (size_t)sgm * seg_cap + base;
                for (uint32_t i = lane; i < wcount; i += 64) dst[i] = wstage[i];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            wcount = 0;
        };
        // A workgroup that holds a contiguous run of a long array (step 1) reads it in a scattered
        // order instead: position p of the array is tile (p * S) mod T, S a prime that does not
        // divide T.  The values inside the bracket of a sorted column are one contiguous stretch,
        // and whoever reads that stretch does all the appending; scattered, every workgroup that
        // shares the array gets its part of it (96 timestamp columns of 2^22 rows: 798 us against
        // 630 us for shuffled values before).
        const uint64_t T = (n + TILE - 1) / TILE;
        uint64_t S = 0;
        if (step == 1 && T > 512) S = T % 127 ? 127 : T % 131 ? 131 : T % 137 ? 137 : 139;  // T < 127 * 131 * 137 * 139
        uint64_t q = S ? ((t0 - a_first) * S) % T : t0 - a_first;  // tile of the array, uniform
        for (uint64_t t = t0; t < t1; t += step) {
            const uint64_t v0 = q * TILE;
            q = S ? (q + S >= T ? q + S - T : q + S) : q + step;
            const bool whole = v0 + TILE <= n;  // uniform
            int64_t x[2 * PAIRS];
            load_tile_from(values, kind, v0, n, fill, x);
            if (mode == SEL_BRACKET_HIST) {
                if (whole) bracket_values<SEL_BRACKET_HIST, true, 0, 2 * PAIRS>(x, kind, v0, n, lo_b, hi_b, acc, lh, nullptr, nullptr);
                else bracket_values<SEL_BRACKET_HIST, false, 0, 2 * PAIRS>(x, kind, v0, n, lo_b, hi_b, acc, lh, nullptr, nullptr);
            } else if (mode == SEL_FULL) {
                bracket_values<SEL_FULL, true, 0, 2 * PAIRS>(x, kind, v0, n, lo_b, hi_b, acc, lh, nullptr, nullptr);
            } else if (mode == SEL_BRACKET_MICRO) {
                if (whole) bracket_values<SEL_BRACKET_MICRO, true, 0, 2 * PAIRS>(x, kind, v0, n, lo_b, hi_b, acc, lh, nullptr, nullptr);
                else bracket_values<SEL_BRACKET_MICRO, false, 0, 2 * PAIRS>(x, kind, v0, n, lo_b, hi_b, acc, lh, nullptr, nullptr);
            } else if (mode == SEL_BRACKET_ONE) {
                if (whole) bracket_values<SEL_BRACKET_ONE, true, 0, 2 * PAIRS>(x, kind, v0, n, lo_b, hi_b, acc, lh, nullptr, nullptr);
                else bracket_values<SEL_BRACKET_ONE, false, 0, 2 * PAIRS>(x, kind, v0, n, lo_b, hi_b, acc, lh, nullptr, nullptr);
            } else if (mode == SEL_BRACKET_TWO) {
                if (whole) bracket_values<SEL_BRACKET_TWO, true, 0, 2 * PAIRS>(x, kind, v0, n, lo_b, hi_b, acc, lh, nullptr, nullptr);
                else bracket_values<SEL_BRACKET_TWO, false, 0, 2 * PAIRS>(x, kind, v0, n, lo_b, hi_b, acc, lh, nullptr, nullptr);
            } else {
                // a wave appends at most TILE / CAND_PARTS / WAVES values per part of a tile
#define ZBRA_COMPACT_PART(E0, E1)                                                                                                  \
    if (wcount + TILE / CAND_PARTS / WAVES > WAVE_CAP) flush_wave();                                                                \
    if (whole) bracket_values<SEL_BRACKET_COMPACT, true, E0, E1>(x, kind, v0, n, lo_b, hi_b, acc, lh, wstage, &wcount, dsh, wcnt); \
    else bracket_values<SEL_BRACKET_COMPACT, false, E0, E1>(x, kind, v0, n, lo_b, hi_b, acc, lh, wstage, &wcount, dsh, wcnt);
                if (CAND_PARTS == 4) {
                    ZBRA_COMPACT_PART(0, 2)
                    ZBRA_COMPACT_PART(2, 4)
                    ZBRA_COMPACT_PART(4, 6)
                    ZBRA_COMPACT_PART(6, 8)
                } else {
                    ZBRA_COMPACT_PART(0, 4)
                    ZBRA_COMPACT_PART(4, 8)
                }
#undef ZBRA_COMPACT_PART
            }
        }
        // min / max: most waves cannot improve on what is already there: look before queueing an
        // atomic on these two words (a stale look only costs an atomic that changes nothing)
        const uint64_t lo = wave_min_u64(acc.lo), hi = wave_max_u64(acc.hi);
        if ((threadIdx.x & 63) == 0) {
            unsigned long long *pmin = (unsigned long long *)&jobs[a].min_b, *pmax = (unsigned long long *)&jobs[a].max_b;
            if (lo < __hip_atomic_load(pmin, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(pmin, (unsigned long long)lo);
            if (hi > __hip_atomic_load(pmax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(pmax, (unsigned long long)hi);
        }
        if (mode != SEL_FULL) {
            if (mode == SEL_BRACKET_COMPACT && lane == 0) { acc.below = wcnt[0]; acc.inb = wcnt[1]; }
            if (acc.below) atomicAdd(&lbelow, (unsigned long long)acc.below);
            if (acc.inb) atomicAdd(&linb, (unsigned long long)acc.inb);
            const uint64_t above = wave_min_u64(acc.above);
            if ((threadIdx.x & 63) == 0) {
                unsigned long long *pabove = (unsigned long long *)&jobs[a].next_above;
                if (above < __hip_atomic_load(pabove, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(pabove, (unsigned long long)above);
            }
            if (mode == SEL_BRACKET_COMPACT) flush_wave();
            if (mode == SEL_BRACKET_TWO) {
                uint32_t c = acc.first;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
                if ((threadIdx.x & 63) == 0 && c) atomicAdd(&lfirst, c);
            }
            if (mode == SEL_BRACKET_MICRO) {  // sum the lanes' private columns: per wave, then per workgroup in LDS
#pragma unroll
                for (uint32_t q = 0; q < MICRO_KEYS; q++) {
                    uint32_t c = lh[q * THREADS + threadIdx.x];
                    lh[q * THREADS + threadIdx.x] = 0;  // the invariant: an all-zero pool between runs
#pragma unroll
                    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
                    if ((threadIdx.x & 63) == 0 && c) atomicAdd(&lh[MICRO_KEYS * THREADS + q], c);
                }
            }
            __syncthreads();
            if (mode == SEL_BRACKET_MICRO && threadIdx.x < MICRO_KEYS) {
                // ONE atomic per key and workgroup on the array's histogram: atomics of every workgroup on
                // the same few words queue behind each other (one per wave made this pass twice as long)
                const uint32_t c = lh[MICRO_KEYS * THREADS + threadIdx.x];
                if (c) { atomicAdd(&job.hist[threadIdx.x], c); lh[MICRO_KEYS * THREADS + threadIdx.x] = 0; }
            }
            if (mode == SEL_BRACKET_HIST || mode == SEL_BRACKET_COMPACT) {
                uint32_t *gh = job.hist;
                for (uint32_t b = threadIdx.x; b < BINS; b += THREADS) {
                    const uint32_t c = lh[b];
                    if (c) { atomicAdd(&gh[b], c); lh[b] = 0; }
                }
            }
            if (threadIdx.x == 0) {
                if (lbelow) atomicAdd(&ctr->below, lbelow);
                if (linb) atomicAdd(&ctr->inb, linb);
                // one- and two-key brackets: the histogram's bins are these counts, which the set-up kernel
                // takes from the sixteen slots (a thousand workgroups adding to hist[0] queue behind each other)
                if (mode == SEL_BRACKET_TWO && lfirst) atomicAdd(&ctr->first, (unsigned long long)lfirst);
                lbelow = 0; linb = 0; lfirst = 0;
            }
            __syncthreads();
        }
    };
    if (nd == 1) {
        // One array: the workgroups take its tiles in turn.  With a contiguous run each, the
        // values inside the bracket of a SORTED column (one stretch of 6 % of the array) would all
        // be appended by the few workgroups whose runs hold it (a 2^27-row timestamp column:
        // 263 us against 199 us for shuffled values).
        if (blockIdx.x < ntiles) run(0, 0, blockIdx.x, ntiles, gridDim.x);
    } else {
        for_each_array_run(tile_start, nd, ntiles, [&](uint32_t a, uint64_t a_first, uint64_t t0, uint64_t t1) { run(a, a_first, t0, t1, 1); });
    }
}

// ---- 2b. the same pass for submissions of u32 arrays only (enum tags) --------------------------
//
// A tile of 2048 tags is 8 KiB: at the four waves per SIMD the pass above runs with, a CU has half
// the bytes in flight that int64 columns give it, and per value it does 64-bit compares on numbers
// that have 32 bits (2^29 tags were read at 4.1 TB/s against 6.3-6.6 for int64 columns).  When the
// host sees that every array of a submission is u32 it launches this kernel instead: the same
// routes, counters and epilogues, but values stay 32 bits wide (the bracket's bounds are brought
// into the u32 domain once per run: they are biased array elements, or "no bound"), a trip of the
// tile loop loads TWO tiles before it looks at either, and candidates are staged as 32-bit words.
struct BracketAcc32 {
    uint32_t lo = ~0u, hi = 0, above = ~0u;
    uint32_t any_above = 0;
    uint32_t below = 0, inb = 0, first = 0;
};
constexpr int TAGS_PER_LANE = TILE / THREADS;  // 8: two 16-byte loads
__device__ __forceinline__ void load_tile_u32(const void *values, uint64_t v0, uint64_t n, uint32_t fill, uint32_t (&x)[TAGS_PER_LANE]) {
    const uint32_t tid = threadIdx.x;
    static_assert(TAGS_PER_LANE == 8, "two 16-byte loads per lane");
    if (v0 + TILE <= n) {
        const ZBRA_GLOBAL u32x4 *src = (const ZBRA_GLOBAL u32x4 *)(gptr<uint32_t>(values) + v0) + tid;
        const u32x4 a = __builtin_nontemporal_load(src), b = __builtin_nontemporal_load(src + THREADS);
        x[0] = a.x; x[1] = a.y; x[2] = a.z; x[3] = a.w; x[4] = b.x; x[5] = b.y; x[6] = b.z; x[7] = b.w;
    } else {
#pragma unroll
        for (int e = 0; e < TAGS_PER_LANE; ++e) {
            const uint64_t i = v0 + tile_slot(ZBRA_ELEM_U32, e);
            x[e] = i < n ? gptr<uint32_t>(values)[i] : fill;
        }
    }
}
// bracket_values for u32 values: lo32 / hi32 are the bracket in the u32 domain.
template <uint32_t MODE, bool WHOLE, int E0, int E1>
__device__ __forceinline__ void bracket_values_u32(const uint32_t (&x)[TAGS_PER_LANE], uint64_t v0, uint64_t n, uint32_t lo32, uint32_t hi32,
                                                   BracketAcc32 &acc, uint32_t *lh, uint32_t *lcand, uint32_t *lcount,
                                                   uint32_t digit_shift = 0, uint32_t *wave_cnt = nullptr, uint64_t lo_b = 0) {
    uint64_t improves = 0;
#pragma unroll
    for (int e = E0; e < E1; ++e) improves |= __builtin_amdgcn_ballot_w64(x[e] < acc.lo) | __builtin_amdgcn_ballot_w64(x[e] > acc.hi);
    if (improves != 0) {
        asm volatile("" ::: "memory");
#pragma unroll
        for (int e = E0; e < E1; ++e) { acc.lo = min(acc.lo, x[e]); acc.hi = max(acc.hi, x[e]); }
    }
#pragma unroll
    for (int e = E0; e < E1; ++e) {
        const uint32_t b = x[e];
        if (MODE == SEL_BRACKET_COMPACT) {
            const bool valid = WHOLE || v0 + tile_slot(ZBRA_ELEM_U32, e) < n;
            const uint32_t r = b - lo32;
            const bool inside = valid && r <= hi32 - lo32;
            const uint64_t m = __builtin_amdgcn_ballot_w64(inside);
            wave_cnt[0] += (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(valid && b < lo32));
            wave_cnt[1] += (uint32_t)__popcll(m);
            if (inside) {
                lcand[*lcount + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))] = b;
                // first digit of the select's 64-bit key, (v ^ SIGN) - lo_b (a bracket without a lower bound has lo_b = 0)
                atomicAdd(&lh[(uint32_t)((((uint64_t)b | SIGN) - lo_b) >> digit_shift)], 1u);
            }
            *lcount += (uint32_t)__popcll(m);
        } else if (MODE != SEL_FULL) {
            const bool valid = WHOLE || v0 + tile_slot(ZBRA_ELEM_U32, e) < n;
            const bool is_below = valid && b < lo32, is_above = valid && b > hi32;
            const bool inside = valid && !is_below && !is_above;
            acc.below += is_below;
            acc.inb += inside;
            acc.any_above |= is_above;
            acc.above = is_above && b < acc.above ? b : acc.above;
            if (MODE == SEL_BRACKET_ONE) {
            } else if (MODE == SEL_BRACKET_TWO) {
                acc.first += inside && b == lo32;
            } else if (MODE == SEL_BRACKET_MICRO) {
                if (inside) lh[((b - lo32) & (MICRO_KEYS - 1)) * THREADS + threadIdx.x] += 1u;
            } else if (inside) {
                atomicAdd(&lh[b - lo32], 1u);
            }
        }
    }
}

#ifndef ZBRA_BRACKET_U32_WAVES
#define ZBRA_BRACKET_U32_WAVES 4
#endif
__global__ __launch_bounds__(THREADS, ZBRA_BRACKET_U32_WAVES) void enc_bracket_u32_kernel(EncodeJob *__restrict__ jobs,
                                                                                       const uint64_t *__restrict__ tile_start, uint32_t nd,
                                                                                       uint64_t ntiles) {
    __shared__ uint32_t lh[BINS];
    __shared__ __attribute__((aligned(16))) uint32_t lcand[CAND_LDS];
    __shared__ unsigned long long lbelow, linb;
    __shared__ uint32_t lfirst;
    for (uint32_t b = threadIdx.x; b < BINS; b += THREADS) lh[b] = 0;
    if (threadIdx.x == 0) { lbelow = 0; linb = 0; lfirst = 0; }
    __syncthreads();
    auto run = [&](uint32_t a, uint64_t a_first, uint64_t t0, uint64_t t1, uint64_t step) {
        const EncodeJob &job = jobs[a];
        if (job.mode == SEL_DONE) return;  // uniform
        const uint64_t bspan = job.hi_b - job.lo_b;
        const uint32_t mode = job.mode != SEL_BRACKET_HIST || bspan >= (uint64_t)MICRO_KEYS ? job.mode
                              : bspan == 0 ? SEL_BRACKET_ONE : bspan == 1 ? SEL_BRACKET_TWO : SEL_BRACKET_MICRO;
        const void *values = job.values;
        const uint64_t n = job.n, seg_cap = job.seg_cap;
        BracketCounters::Slot *ctr = &job.counters->slot[blockIdx.x % CAND_SEGS];
        uint32_t flushes = 0;
        // the bracket in the u32 domain: a bound is the biased image of an array element, or 0 / ~0 for "none"
        const uint32_t lo32 = job.lo_b >= SIGN ? (uint32_t)(job.lo_b - SIGN) : 0u;
        const uint32_t hi32 = job.hi_b < SIGN + 0x100000000ull ? (uint32_t)(job.hi_b - SIGN) : 0xFFFFFFFFu;
        const uint64_t lo_b64 = job.lo_b;
        const uint32_t fill = gptr<uint32_t>(values)[0];
        BracketAcc32 acc;
        acc.lo = (uint32_t)(__hip_atomic_load(&job.min_b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - SIGN);
        acc.hi = (uint32_t)(__hip_atomic_load(&job.max_b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - SIGN);
        constexpr uint32_t WAVES = THREADS / 64, WAVE_CAP = CAND_LDS / WAVES;
        const uint32_t dsh = digit_shift(job.key_bits, 0);
        const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
        uint32_t *wstage = lcand + wave * WAVE_CAP;
        uint32_t wcount = 0;
        uint32_t wcnt[2] = {0, 0};
        auto flush_wave = [&]() {
            if (wcount == 0) return;
            const uint32_t sgm = (blockIdx.x * WAVES + wave + flushes++) % CAND_SEGS;
            unsigned long long base = 0;
            if (lane == 0) base = atomicAdd(&job.counters->slot[sgm].cand_count, (unsigned long long)wcount);
            base = __shfl(base, 0);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            if (base + wcount <= seg_cap) {
                ZBRA_GLOBAL int64_t *dst = gptr_mut<int64_t>(job.cand) + (size_t)sgm * seg_cap + base;
                for (uint32_t i = lane; i < wcount; i += 64) dst[i] = (int64_t)wstage[i];  // the list holds int64 values
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            wcount = 0;
        };
        auto tile = [&](const uint32_t (&x)[TAGS_PER_LANE], uint64_t v0) {
            const bool whole = v0 + TILE <= n;  // uniform
            if (mode == SEL_BRACKET_HIST) {
                if (whole) bracket_values_u32<SEL_BRACKET_HIST, true, 0, TAGS_PER_LANE>(x, v0, n, lo32, hi32, acc, lh, nullptr, nullptr);
                else bracket_values_u32<SEL_BRACKET_HIST, false, 0, TAGS_PER_LANE>(x, v0, n, lo32, hi32, acc, lh, nullptr, nullptr);
            } else if (mode == SEL_FULL) {
                bracket_values_u32<SEL_FULL, true, 0, TAGS_PER_LANE>(x, v0, n, lo32, hi32, acc, lh, nullptr, nullptr);
            } else if (mode == SEL_BRACKET_MICRO) {
                if (whole) bracket_values_u32<SEL_BRACKET_MICRO, true, 0, TAGS_PER_LANE>(x, v0, n, lo32, hi32, acc, lh, nullptr, nullptr);
                else bracket_values_u32<SEL_BRACKET_MICRO, false, 0, TAGS_PER_LANE>(x, v0, n, lo32, hi32, acc, lh, nullptr, nullptr);
            } else if (mode == SEL_BRACKET_ONE) {
                if (whole) bracket_values_u32<SEL_BRACKET_ONE, true, 0, TAGS_PER_LANE>(x, v0, n, lo32, hi32, acc, lh, nullptr, nullptr);
                else bracket_values_u32<SEL_BRACKET_ONE, false, 0, TAGS_PER_LANE>(x, v0, n, lo32, hi32, acc, lh, nullptr, nullptr);
            } else if (mode == SEL_BRACKET_TWO) {
                if (whole) bracket_values_u32<SEL_BRACKET_TWO, true, 0, TAGS_PER_LANE>(x, v0, n, lo32, hi32, acc, lh, nullptr, nullptr);
                else bracket_values_u32<SEL_BRACKET_TWO, false, 0, TAGS_PER_LANE>(x, v0, n, lo32, hi32, acc, lh, nullptr, nullptr);
            } else {
                // a wave appends at most TILE / 2 / WAVES values per half of a tile
                static_assert(WAVE_CAP >= 2 * (TILE / 2 / WAVES), "room for two halves between looks");
#define ZBRA_COMPACT_PART32(E0, E1)                                                                                                   \
    if (wcount + TILE / 2 / WAVES > WAVE_CAP) flush_wave();                                                                            \
    if (whole) bracket_values_u32<SEL_BRACKET_COMPACT, true, E0, E1>(x, v0, n, lo32, hi32, acc, lh, wstage, &wcount, dsh, wcnt, lo_b64); \
    else bracket_values_u32<SEL_BRACKET_COMPACT, false, E0, E1>(x, v0, n, lo32, hi32, acc, lh, wstage, &wcount, dsh, wcnt, lo_b64);
                ZBRA_COMPACT_PART32(0, 4)
                ZBRA_COMPACT_PART32(4, 8)
#undef ZBRA_COMPACT_PART32
            }
        };
        // the same walk over the array's tiles as in enc_bracket_kernel (scattered inside a contiguous run)
        const uint64_t T = (n + TILE - 1) / TILE;
        uint64_t S = 0;
        if (step == 1 && T > 512) S = T % 127 ? 127 : T % 131 ? 131 : T % 137 ? 137 : 139;
        uint64_t q = S ? ((t0 - a_first) * S) % T : t0 - a_first;
        auto next_q = [&]() { q = S ? (q + S >= T ? q + S - T : q + S) : q + step; };
        for (uint64_t t = t0; t < t1;) {
            // two tiles per trip, both tiles' loads issued before either is looked at
            const uint64_t va = q * TILE;
            next_q();
            const bool pair = t + step < t1;  // uniform
            const uint64_t vb = q * TILE;
            uint32_t xa[TAGS_PER_LANE], xb[TAGS_PER_LANE];
            load_tile_u32(values, va, n, fill, xa);
            if (pair) { load_tile_u32(values, vb, n, fill, xb); next_q(); }
            tile(xa, va);
            if (pair) tile(xb, vb);
            t += pair ? 2 * step : step;
        }
        uint32_t lo = acc.lo, hi = acc.hi;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { lo = min(lo, (uint32_t)__shfl_xor(lo, o)); hi = max(hi, (uint32_t)__shfl_xor(hi, o)); }
        if (lane == 0) {
            unsigned long long *pmin = (unsigned long long *)&jobs[a].min_b, *pmax = (unsigned long long *)&jobs[a].max_b;
            const unsigned long long lo_b = SIGN + lo, hi_b = SIGN + hi;
            if (lo_b < __hip_atomic_load(pmin, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(pmin, lo_b);
            if (hi_b > __hip_atomic_load(pmax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(pmax, hi_b);
        }
        if (mode != SEL_FULL) {
            if (mode == SEL_BRACKET_COMPACT && lane == 0) { acc.below = wcnt[0]; acc.inb = wcnt[1]; }
            if (acc.below) atomicAdd(&lbelow, (unsigned long long)acc.below);
            if (acc.inb) atomicAdd(&linb, (unsigned long long)acc.inb);
            const uint64_t above = wave_min_u64(acc.any_above ? SIGN + acc.above : ~0ull);
            if (lane == 0) {
                unsigned long long *pabove = (unsigned long long *)&jobs[a].next_above;
                if (above < __hip_atomic_load(pabove, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(pabove, (unsigned long long)above);
            }
            if (mode == SEL_BRACKET_COMPACT) flush_wave();
            if (mode == SEL_BRACKET_TWO) {
                uint32_t c = acc.first;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
                if (lane == 0 && c) atomicAdd(&lfirst, c);
            }
            if (mode == SEL_BRACKET_MICRO) {
#pragma unroll
                for (uint32_t k = 0; k < MICRO_KEYS; k++) {
                    uint32_t c = lh[k * THREADS + threadIdx.x];
                    lh[k * THREADS + threadIdx.x] = 0;
#pragma unroll
                    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
                    if (lane == 0 && c) atomicAdd(&lh[MICRO_KEYS * THREADS + k], c);
                }
            }
            __syncthreads();
            if (mode == SEL_BRACKET_MICRO && threadIdx.x < MICRO_KEYS) {
                const uint32_t c = lh[MICRO_KEYS * THREADS + threadIdx.x];
                if (c) { atomicAdd(&job.hist[threadIdx.x], c); lh[MICRO_KEYS * THREADS + threadIdx.x] = 0; }
            }
            if (mode == SEL_BRACKET_HIST || mode == SEL_BRACKET_COMPACT) {
                uint32_t *gh = job.hist;
                for (uint32_t b = threadIdx.x; b < BINS; b += THREADS) {
                    const uint32_t c = lh[b];
                    if (c) { atomicAdd(&gh[b], c); lh[b] = 0; }
                }
            }
            if (threadIdx.x == 0) {
                if (lbelow) atomicAdd(&ctr->below, lbelow);
                if (linb) atomicAdd(&ctr->inb, linb);
                if (mode == SEL_BRACKET_TWO && lfirst) atomicAdd(&ctr->first, (unsigned long long)lfirst);
                lbelow = 0; linb = 0; lfirst = 0;
            }
            __syncthreads();
        }
    };
    if (nd == 1) {
        if (blockIdx.x < ntiles) run(0, 0, blockIdx.x, ntiles, gridDim.x);
    } else {
        for_each_array_run(tile_start, nd, ntiles, [&](uint32_t a, uint64_t a_first, uint64_t t0, uint64_t t1) { run(a, a_first, t0, t1, 1); });
    }
}

// ---- 3. verify the bracket, set up the radix select -------------------------------------------

// The bin that holds rank k of a BINS-bin histogram, found by one workgroup of THREADS lanes: lane
// t brings bins [t * SEL_PER, +SEL_PER).  Every lane gets the same answer.
constexpr int SEL_PER = BINS / THREADS;  // 16 consecutive bins per thread
struct BinPick {
    uint32_t bin, cnt;   // the bin of rank k and how many keys it holds
    uint32_t next_bin;   // the next bin above it that holds a key (BINS if none)
    uint32_t pad;
    uint64_t before;     // keys in the bins below
};
struct BinPickShared {
    uint64_t part[THREADS / 64];  // the waves' totals
    BinPick pick;
};
__device__ __forceinline__ BinPick pick_bin(const uint32_t (&c)[SEL_PER], uint64_t k, BinPickShared *sh) {
    const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63u;
    uint64_t sum = 0;
#pragma unroll
    for (int i = 0; i < SEL_PER; i++) sum += c[i];
    // inclusive prefix of the per-thread sums: inside the wave by shuffles, across the waves through
    // their totals in LDS (three barriers in all; a Hillis-Steele scan over the 256 sums in LDS took 16)
    uint64_t incl = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint64_t add = __shfl_up(incl, o);
        if (lane >= (uint32_t)o) incl += add;
    }
    __syncthreads();  // the previous use of `sh` is over
    if (lane == 63) sh->part[wave] = incl;
    if (tid == 0) { sh->pick.next_bin = BINS; sh->pick.bin = 0; sh->pick.cnt = 0; sh->pick.before = 0; }
    __syncthreads();
    for (uint32_t w = 0; w < wave; w++) incl += sh->part[w];
    const uint64_t before = incl - sum;
    if (k >= before && k < incl) {  // exactly one thread
        uint64_t run = before;
#pragma unroll
        for (int i = 0; i < SEL_PER; i++) {
            if (k >= run && k < run + c[i]) { sh->pick.bin = tid * SEL_PER + i; sh->pick.before = run; sh->pick.cnt = c[i]; }
            run += c[i];
        }
    }
    __syncthreads();
    const uint32_t d = sh->pick.bin;
#pragma unroll
    for (int i = 0; i < SEL_PER; i++) {
        const uint32_t b = tid * SEL_PER + i;
        if (b > d && c[i]) { atomicMin(&sh->pick.next_bin, b); break; }
    }
    __syncthreads();
    return sh->pick;
}

// a and b (the keys of rank k and, for an even count, of rank k + 1) are known: the midpoint.
__device__ __forceinline__ void finish_select(EncodeJob &job, uint64_t key_a, uint64_t key_b) {
    const int64_t va = (int64_t)((key_a + job.key_base) ^ SIGN);
    const int64_t vb = (int64_t)((key_b + job.key_base) ^ SIGN);
    set_midpoint(job, va, vb, (job.n & 1) == 0);
}

// State s says digit s.lv of the keys that share s.prefix is counted in `gh`: picks the bin of rank
// s.k and descends into it.  By the whole workgroup; every lane gets the same state.
// `next_key` = the smallest key above the picked bin as far as this histogram knows it: meaningful
// only when this was the key's last digit (then a bin is one key), ~0 if no higher bin holds a key.
__device__ __forceinline__ EncodeJob::SelState descend(EncodeJob::SelState s, const uint32_t *gh, BinPickShared *sh, uint64_t *next_key,
                                                       uint32_t bits) {
    uint32_t c[SEL_PER];
#pragma unroll
    for (int i = 0; i < SEL_PER; i++) c[i] = gh[threadIdx.x * SEL_PER + i];
    const BinPick pk = pick_bin(c, s.k, sh);
    // the digit's new bits: all DIGIT_BITS of them, except for a narrower last digit, whose bins
    // repeat bits of the prefix in their upper part
    const uint32_t d = s.lv == 0 ? (uint32_t)DIGIT_BITS : digit_shift(bits, s.lv - 1) - digit_shift(bits, s.lv);
    const uint64_t low = (1ull << d) - 1;
    const uint64_t high = s.lv == 0 ? 0 : s.prefix << d;
    *next_key = pk.next_bin < (uint32_t)BINS ? (high | (pk.next_bin & low)) : ~0ull;
    s.prefix = high | (pk.bin & low);
    s.k -= pk.before;
    s.cnt = pk.cnt;
    s.lv += 1;
    s.action = ACT_NONE;
    return s;
}

// Every digit of the key of a is resolved (s.lv == passes): a = s.prefix, and b, for an even
// count, is the same key (multiplicity), the next key of the last histogram, or the smallest key
// above everything that shared the earlier digits.  One lane.
__device__ __forceinline__ void finish_digits(EncodeJob &job, const EncodeJob::SelState &s, uint64_t next_key) {
    uint64_t key_b = s.prefix;
    if ((job.n & 1) == 0 && s.k + 1 >= s.cnt) {
        key_b = next_key;
        if (key_b == ~0ull) key_b = (uint64_t)job.above_key < job.next_above ? (uint64_t)job.above_key : job.next_above;
    }
    finish_select(job, s.prefix, key_b);
}

// The s.cnt keys that share s.prefix are all `key` (the level that counted their next digit saw
// min == max): a = key, and b is the same key or the smallest key above the run.  One lane.
__device__ __forceinline__ void finish_run(EncodeJob &job, const EncodeJob::SelState &s, uint64_t key) {
    uint64_t key_b = key;
    if ((job.n & 1) == 0 && s.k + 1 >= s.cnt) key_b = (uint64_t)job.above_key < job.next_above ? (uint64_t)job.above_key : job.next_above;
    finish_select(job, key, key_b);
}

// One workgroup per array.  Arrays whose select has more than `small_max` keys to read (and that
// are long enough for that: a list holds at most n / CAND_DIV values) are put on the `pending`
// list ({count, arrays...}, zero on entry) of the many-workgroup levels: one workgroup reads at
// about 11 GB/s, so a long source is better spread over many.  Everything else is left to the one
// workgroup per array of enc_finish_select_kernel.
__global__ __launch_bounds__(THREADS) void enc_setup_kernel(EncodeJob *__restrict__ jobs, uint32_t nd,
                                                             uint32_t *__restrict__ pending, uint64_t small_max,
                                                             const uint64_t *__restrict__ tile_start, uint32_t *__restrict__ tile_map,
                                                             uint32_t *__restrict__ or_flag) {
    const uint32_t a = blockIdx.x;
    if (a >= nd) return;
    // tile -> array table of the pack pass (one workgroup per tile there: a binary search over
    // tile_start would be a dozen dependent loads in front of every tile)
    if (tile_map)
        for (uint64_t t = tile_start[a] + threadIdx.x; t < tile_start[a + 1]; t += THREADS) tile_map[t] = a;
    EncodeJob &j = jobs[a];
    __shared__ EncodeJob::SelState s0;
    __shared__ uint64_t src_n;
    __shared__ BinPickShared sh;
    if (threadIdx.x == 0) {
        s0.prefix = 0; s0.k = 0; s0.cnt = 0; s0.lv = 0; s0.action = ACT_DONE;
        src_n = 0;
        if (j.n == 0) {
            j.passes = 0;
        } else if (j.kind == KIND_U64_PLAIN) {
            j.passes = 0; j.need_or = 1; j.midpoint = 0;
        } else if (j.mode != SEL_DONE) {
            const uint64_t span = j.max_b - j.min_b;
            j.need_or = span >= SIGN;
            const uint64_t k = (j.n - 1) / 2;  // rank of a: n/2 - 1 for even n, n/2 for odd n
            bool whole_list = true;
            if (j.mode != SEL_FULL) {
                // summed in registers: through `j` every step would be a load and a store that wait for each other
                const BracketCounters *ctr = j.counters;
                const uint64_t seg_cap = j.seg_cap;
                uint64_t below = 0, inb = 0, first = 0;
#pragma unroll
                for (int sgm = 0; sgm < CAND_SEGS; sgm++) {
                    below += ctr->slot[sgm].below;
                    inb += ctr->slot[sgm].inb;
                    first += ctr->slot[sgm].first;
                    whole_list = whole_list && ctr->slot[sgm].cand_count <= seg_cap;
                }
                j.below = below; j.inb = inb;
                // a bracket of one or two keys: the bracket pass left its bins in the slots' counts
                if (j.mode == SEL_BRACKET_HIST && j.hi_b - j.lo_b <= 1) {
                    j.hist[0] = j.hi_b == j.lo_b ? (uint32_t)inb : (uint32_t)first;
                    if (j.hi_b != j.lo_b) j.hist[1] = (uint32_t)(inb - first);
                }
            }
            // Rank k must fall inside the bracket (rank k+1, for even n, may also be the smallest
            // value above it).  The bracket pass has then counted the first digit of the keys inside
            // it (every digit there is, for a narrow bracket), whether or not the candidate list is
            // whole.
            // A wide bracket's pass does not look for the smallest value above the bracket: there
            // both ranks must be inside.
            const uint64_t k_hi = j.mode == SEL_BRACKET_COMPACT && (j.n & 1) == 0 ? k + 1 : k;
            const bool ok = j.mode != SEL_FULL && k >= j.below && k_hi < j.below + j.inb;
            if (ok) {
                // key_base = lo_b and passes were set with the bracket, by the sample kernel
                if (j.next_above != ~0ull) j.next_above -= j.key_base;  // into key space
                s0.k = k - j.below; s0.cnt = j.inb; s0.action = ACT_HIST;
                j.flo = j.lo_b; j.fhi = j.hi_b;
                if (j.mode == SEL_BRACKET_HIST) {
                    src_n = 0;  // nothing left to read
                } else if (whole_list) {
                    src_n = j.inb;
                } else {  // more than n / CAND_DIV values inside a wide bracket: the select reads the array itself
                    j.mode = SEL_FULL;
                    src_n = j.n;
                }
            } else {
                j.mode = SEL_FULL;
                j.key_base = j.min_b;
                j.next_above = ~0ull;
                j.key_bits = bit_length64(span);
                j.passes = (j.key_bits + DIGIT_BITS - 1) / DIGIT_BITS;
                j.flo = 0; j.fhi = ~0ull;
                s0.k = k; s0.cnt = j.n; s0.action = ACT_NONE;
                src_n = j.n;
                if (j.passes == 0) {  // constant column
                    const int64_t v = (int64_t)(j.min_b ^ SIGN);
                    set_midpoint(j, v, v, false);
                    s0.action = ACT_DONE;
                }
            }
        }
    }
    __syncthreads();
    EncodeJob::SelState s = s0;
    if (s.action == ACT_HIST) {
        // the first digit is counted: pick its bin here, once, rather than in every workgroup of
        // the first level; a narrow bracket is finished by this
        uint64_t next_key = ~0ull;
        s = descend(s, j.hist, &sh, &next_key, j.key_bits);
        if (threadIdx.x == 0 && s.lv == j.passes) {
            finish_digits(j, s, next_key);
            s.action = ACT_DONE;
        }
        if (s.lv == j.passes) s.action = ACT_DONE;
    }
    if (threadIdx.x == 0) {
        // one word for the whole submission: the OR pass ends at its first load unless some array's span wraps
        if (j.n && j.need_or) *or_flag = 1u;
        j.S[0] = s;
        // A candidate list holds at most n / CAND_DIV values, so an array of up to CAND_DIV *
        // small_max values can only exceed small_max by reading all its values; those stay with
        // the one workgroup too, which lets the host skip the many-workgroup levels for a
        // submission without longer arrays.
        if (s.action != ACT_DONE && src_n > small_max && j.n > (uint64_t)CAND_DIV * small_max && j.lvl_hist) {
            j.pending = 1;
            pending[1 + atomicAdd(&pending[0], 1u)] = a;
        }
    }
}

// ---- 4. radix select ---------------------------------------------------------------------

// Keys that share s.prefix (s.lv digits resolved) in a key of `passes` digits.
__device__ __forceinline__ bool shares_prefix(uint64_t key, const EncodeJob::SelState &s, uint32_t bits) {
    return s.lv == 0 || (key >> digit_shift(bits, s.lv - 1)) == s.prefix;
}
__device__ __forceinline__ bool above_prefix(uint64_t key, const EncodeJob::SelState &s, uint32_t bits) {
    return s.lv != 0 && (key >> digit_shift(bits, s.lv - 1)) > s.prefix;
}

// ---- 4a. long select sources: up to SEL_LEVELS launches of many workgroups ----------------------
//
// Level L (1..SEL_LEVELS), for every pending array: every workgroup that takes part first brings the
// array's state up to date by itself (if the launch before counted a digit, it picks the bin of
// rank k from that histogram: 16 KiB read, one scan; all workgroups get the same answer, the first
// one records it as S[L]), then
//   - if every digit is resolved, the first workgroup derives the midpoint;
//   - if at most GATHER_CAP keys share the resolved digits, the workgroups gather them (and the
//     smallest key above them) for the finishing workgroup;
//   - else they count the next digit of the keys that share the resolved digits.
// No workgroup waits for another one: kernel boundaries are the only ordering.

// One workgroup's share (tiles w, w + eff, ...) of one array at level L.
__device__ __forceinline__ void level_one(EncodeJob *__restrict__ jobs, uint32_t a, uint32_t w, uint32_t per_array, uint32_t tiles_per_wg,
                                          uint32_t L, uint32_t *lh, BinPickShared *sh, uint32_t sparse_tiles) {
    EncodeJob &job = jobs[a];
    const uint32_t tid = threadIdx.x;
    const EncodeJob::SelState s0 = job.S[L - 1];
    if (s0.action == ACT_DONE || s0.action == ACT_GATHER) {  // finished, or waiting for the finishing workgroup
        if (w == 0 && tid == 0) job.S[L] = s0;
        return;
    }
    // Every workgroup that counts merges up to BINS counters into the array's histogram with
    // device-scope atomics, which cost more than reading a tile: only as many workgroups take part
    // as have tiles_per_wg tiles each.
    const bool from_list = job.mode == SEL_BRACKET_COMPACT;
    const uint64_t work_tiles = ((from_list ? job.inb : job.n) + TILE - 1) / TILE;
    // A digit counted among keys that share earlier digits, and a gather, touch few counters:
    // nothing to merge, so every workgroup is worth having.
    const bool sparse = s0.action == ACT_HIST || s0.lv > 0;
    if (sparse) tiles_per_wg = sparse_tiles ? sparse_tiles : tiles_per_wg == 0 ? 8u : 2u;  // few arrays / many (measured: 1 to 32 tiles; many arrays do not care)
    else if (tiles_per_wg == 0) tiles_per_wg = work_tiles >= 4096 ? 16u : work_tiles < 512 ? 2u : (uint32_t)(work_tiles / 256);
    const uint64_t want = work_tiles / tiles_per_wg;
    const uint32_t eff = want < 1 ? 1u : want < per_array ? (uint32_t)want : per_array;
    if (w >= eff) return;
    const uint32_t passes = job.passes, bits = job.key_bits;
    EncodeJob::SelState s = s0;
    uint64_t next_key = ~0ull;
    if (s0.action == ACT_HIST && L >= 2 && s0.lv > 0 && job.match_min[L - 2] == job.match_max[L - 2]) {
        // the level before counted a digit of keys that are all the same: nothing left to resolve
        if (w == 0 && tid == 0) {
            finish_run(job, s0, job.match_min[L - 2]);
            s.action = ACT_DONE;
            job.S[L] = s;
        }
        return;
    }
    if (s0.action == ACT_HIST) s = descend(s0, L == 1 ? job.hist : job.lvl_hist + (size_t)(L - 2) * BINS, sh, &next_key, bits);
    if (s.lv == passes) {
        if (w == 0 && tid == 0) {
            finish_digits(job, s, next_key);
            s.action = ACT_DONE;
            job.S[L] = s;
        }
        return;
    }
    const bool gather = s.cnt <= (uint64_t)GATHER_CAP;
    if (w == 0 && tid == 0) {
        s.action = gather ? ACT_GATHER : ACT_HIST;
        job.S[L] = s;
    }
    // LDS: the digit's histogram, or the keys this workgroup gathers (at most GATHER_CAP in all)
    static_assert(GATHER_CAP * sizeof(uint64_t) <= BINS * sizeof(uint32_t), "gathered keys are staged in the histogram's LDS");
    uint64_t *stage = (uint64_t *)lh;
    __shared__ uint32_t stage_n, stage_base;
    // How far each segment of the candidate list is filled.  The loop below asks for every slot it
    // looks at, most of them empty: from global memory that was a dependent load per slot, and
    // noticeable when a thousand short lists are pending.
    __shared__ uint64_t seg_fill[CAND_SEGS];
    if (from_list && tid < (uint32_t)CAND_SEGS) seg_fill[tid] = job.counters->slot[tid].cand_count;
    __syncthreads();  // the previous item's merge, and this item's pick, have read LDS
    if (gather) { if (tid == 0) stage_n = 0; }
    else for (uint32_t b = tid; b < BINS; b += THREADS) lh[b] = 0;
    __syncthreads();
    const uint32_t shift = digit_shift(bits, s.lv);  // of the digit to count
    const uint64_t base = job.key_base;
    // `above`: the smallest key above the keys that share the resolved digits.  It is what b falls
    // back on when this turns out to be the array's last read (a gather, the last digit, a run of
    // one value); a later level's set of such keys contains this one's, so one running minimum
    // per array serves them all.
    uint64_t above = ~0ull, kmin = ~0ull, kmax = 0;
    auto visit = [&](uint64_t key) {
        if (shares_prefix(key, s, bits)) {
            if (gather) {
                const uint32_t slot = atomicAdd(&stage_n, 1u);
                if (slot < GATHER_CAP) stage[slot] = key;  // always true while nobody writes the source: s.cnt keys in all
            } else {
                atomicAdd(&lh[(uint32_t)(key >> shift) & (BINS - 1)], 1u);
                kmin = min(kmin, key);
                kmax = max(kmax, key);
            }
        } else if (above_prefix(key, s, bits)) {
            above = min(above, key);
        }
    };
    // A workgroup that looked at one tile at a time would pay a full memory latency per tile: the
    // loads of up to LEVEL_AHEAD tiles are issued before the first is looked at.
    constexpr int LEVEL_AHEAD = 4;
    int64_t x[LEVEL_AHEAD][2 * PAIRS];
    if (!from_list) {
        const void *values = job.values;
        const uint32_t kind = job.kind;
        const uint64_t n = job.n, flo = job.flo, fhi = job.fhi;
        for (uint64_t v0 = (uint64_t)w * TILE; v0 < n; v0 += (uint64_t)LEVEL_AHEAD * eff * TILE) {
#pragma unroll
            for (int u = 0; u < LEVEL_AHEAD; u++) {
                const uint64_t t0 = v0 + (uint64_t)u * eff * TILE;
                if (t0 < n) load_tile_from(values, kind, t0, n, 0, x[u]);
            }
#pragma unroll
            for (int u = 0; u < LEVEL_AHEAD; u++) {
                const uint64_t t0 = v0 + (uint64_t)u * eff * TILE;
                if (t0 < n) {
#pragma unroll
                    for (int e = 0; e < 2 * PAIRS; ++e) {
                        const uint64_t i = t0 + tile_slot(kind, e);
                        const uint64_t b = (uint64_t)x[u][e] ^ SIGN;
                        if (i < n && b >= flo && b <= fhi) visit(b - base);
                    }
                }
            }
        }
    } else {
        // the segmented list, seen as one space of tile slots (slot u = tile u / CAND_SEGS of segment
        // u % CAND_SEGS), of which those below each segment's fill hold values
        const uint64_t seg_cap = job.seg_cap, slots = seg_cap / TILE * CAND_SEGS;
        uint64_t u = w;
        while (u < slots) {  // uniform
            uint64_t t0[LEVEL_AHEAD], end[LEVEL_AHEAD];
            int m = 0;
#pragma unroll
            for (int q = 0; q < LEVEL_AHEAD; q++) { t0[q] = 0; end[q] = 0; }
            while (m < LEVEL_AHEAD && u < slots) {
                const uint64_t sgm = u % CAND_SEGS, first = sgm * seg_cap + (u / CAND_SEGS) * TILE;
                const uint64_t last = sgm * seg_cap + seg_fill[sgm];  // values exist below this index
                u += eff;
                if (first >= last) continue;
#pragma unroll
                for (int q = 0; q < LEVEL_AHEAD; q++)
                    if (q == m) { t0[q] = first; end[q] = last; }
                m++;
            }
#pragma unroll
            for (int q = 0; q < LEVEL_AHEAD; q++)
                // every tile slot lies wholly inside its segment's memory (seg_cap is a multiple of
                // TILE): load it whole, with 16-byte loads, and leave out what is past the fill when
                // looking at the values, instead of the guarded element loads of a ragged tile
                if (q < m) load_tile_from(job.cand, ZBRA_ELEM_I64, t0[q], t0[q] + TILE, 0, x[q]);
#pragma unroll
            for (int q = 0; q < LEVEL_AHEAD; q++) {
                if (q < m) {
#pragma unroll
                    for (int e = 0; e < 2 * PAIRS; ++e) {
                        const uint64_t i = t0[q] + tile_slot(ZBRA_ELEM_I64, e);
                        if (i < end[q]) visit(((uint64_t)x[q][e] ^ SIGN) - base);
                    }
                }
            }
        }
    }
    __syncthreads();
    if (gather) {
        const uint32_t cnt = stage_n < GATHER_CAP ? stage_n : GATHER_CAP;
        const uint32_t seg = w % GATHER_SEGS;
        if (tid == 0 && cnt) stage_base = atomicAdd(&job.small_n[seg], cnt);
        __syncthreads();
        if (cnt) {
            const uint32_t at = stage_base;
            for (uint32_t i = tid; i < cnt; i += THREADS)
                if (at + i < GATHER_CAP) job.small[(size_t)seg * GATHER_CAP + at + i] = stage[i];  // always true while nobody writes the source
        }
    } else {
        uint32_t *gh = job.lvl_hist + (size_t)(L - 1) * BINS;
        for (uint32_t b = tid; b < BINS; b += THREADS) {
            const uint32_t c = lh[b];
            if (c) atomicAdd(&gh[b], c);
        }
    }
    // Thousands of waves hold a key above the prefix and all would queue on one word: look first
    // (a stale look only costs an atomic that changes nothing; the minimum settles after a few).
    above = wave_min_u64(above);
    if ((tid & 63) == 0 && above < __hip_atomic_load(&job.above_key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
        atomicMin(&job.above_key, (unsigned long long)above);
    if (!gather) {
        kmin = wave_min_u64(kmin);
        kmax = wave_max_u64(kmax);
        if ((tid & 63) == 0 && kmin <= kmax) {
            if (kmin < __hip_atomic_load(&job.match_min[L - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
                atomicMin(&job.match_min[L - 1], (unsigned long long)kmin);
            if (kmax > __hip_atomic_load(&job.match_max[L - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
                atomicMax(&job.match_max[L - 1], (unsigned long long)kmax);
        }
    }
}

// `pending` = {count, arrays...}, written by enc_setup_kernel (the host cannot know it).  The grid
// is fixed; its workgroups are dealt evenly to the pending arrays, workgroup w of an array's
// `per_array` taking tiles w, w + per_array, ..., so that an array is never left to the two or
// three workgroups that happen to own its stretch of the submission.  With nothing pending the
// launch ends at its first load.
// 221 VGPRs, two waves per SIMD.  Capped at 168 or 128 VGPRs it spills, and a kernel with scratch
// memory pays for its set-up in every launch, also the ones that find nothing to do (9 -> 18-38 us):
// the first level of a thousand-array batch got 20 % faster, the encode as a whole did not.
__global__ __launch_bounds__(THREADS) void enc_level_kernel(EncodeJob *__restrict__ jobs, const uint32_t *__restrict__ pending, uint32_t L,
                                                             uint32_t sparse_tiles) {
    const uint32_t n_list = pending[0];
    if (n_list == 0) return;
    const uint32_t *list = pending + 1;
    uint32_t per_array = gridDim.x / n_list;
    if (per_array == 0) per_array = 1;
    // Tiles a workgroup reads before it merges its 4096 counters into the array's histogram
    // (device-scope atomics).  One 2^27-value array with a wide bracket: a counting pass takes
    // 142 us at 1.5 tiles per workgroup (4096 workgroups), 62 us at 4, 44 us at 16, 115 us at 64; a
    // thousand arrays keep all their workgroups busy at 4 and slow down beyond.  0 = by the array's
    // length: about 256 workgroups, 2 to 16 tiles each.
    const uint32_t tiles_per_wg = n_list <= 4 ? 0u : 4u;
    __shared__ uint32_t lh[BINS];
    __shared__ BinPickShared sh;
    for (uint64_t item = blockIdx.x; item < (uint64_t)n_list * per_array; item += gridDim.x)  // uniform
        level_one(jobs, list[item / per_array], (uint32_t)(item % per_array), per_array, tiles_per_wg, L, lh, &sh, sparse_tiles);
}

// ---- 4b. the finishing workgroup: one per array, every remaining digit in one launch -------------

// f(key) for every key of what the array's select reads, by the whole workgroup.  One workgroup
// reading tile after tile would pay a full memory latency per tile, so four tiles (of the array,
// or of four segments of the candidate list) are loaded before the first is looked at.
constexpr int SEL_AHEAD = 4;
static_assert(CAND_SEGS % SEL_AHEAD == 0, "segments are read in groups");
template <typename F>
__device__ __forceinline__ void for_each_select_key(const EncodeJob &job, uint32_t mode, F &&f) {
    const uint64_t base = job.key_base;
    int64_t x[SEL_AHEAD][2 * PAIRS];
    if (mode == SEL_FULL) {
        const void *values = job.values;
        const uint32_t kind = job.kind;
        const uint64_t n = job.n, flo = job.flo, fhi = job.fhi;
        for (uint64_t v0 = 0; v0 < n; v0 += (uint64_t)SEL_AHEAD * TILE) {
#pragma unroll
            for (int u = 0; u < SEL_AHEAD; u++)
                if (v0 + (uint64_t)u * TILE < n) load_tile_from(values, kind, v0 + (uint64_t)u * TILE, n, 0, x[u]);
#pragma unroll
            for (int u = 0; u < SEL_AHEAD; u++) {
                const uint64_t t0 = v0 + (uint64_t)u * TILE;
                if (t0 < n) {
#pragma unroll
                    for (int e = 0; e < 2 * PAIRS; ++e) {
                        const uint64_t b = (uint64_t)x[u][e] ^ SIGN;
                        if (t0 + tile_slot(kind, e) < n && b >= flo && b <= fhi) f(b - base);
                    }
                }
            }
        }
    } else {
        for (uint32_t g = 0; g < (uint32_t)CAND_SEGS; g += SEL_AHEAD) {
            uint64_t end[SEL_AHEAD], longest = 0;  // setup made sure that no segment overflowed
#pragma unroll
            for (int u = 0; u < SEL_AHEAD; u++) {
                end[u] = job.counters->slot[g + u].cand_count;
                longest = end[u] > longest ? end[u] : longest;
            }
            for (uint64_t v0 = 0; v0 < longest; v0 += TILE) {
#pragma unroll
                for (int u = 0; u < SEL_AHEAD; u++)
                    if (v0 < end[u]) load_tile_from(job.cand + (size_t)(g + u) * job.seg_cap, ZBRA_ELEM_I64, v0, v0 + TILE, 0, x[u]);  // whole slot, masked below
#pragma unroll
                for (int u = 0; u < SEL_AHEAD; u++) {
                    if (v0 < end[u]) {
#pragma unroll
                        for (int e = 0; e < 2 * PAIRS; ++e)
                            if (v0 + tile_slot(ZBRA_ELEM_I64, e) < end[u]) f(((uint64_t)x[u][e] ^ SIGN) - base);
                    }
                }
            }
        }
    }
}

// One workgroup per listed array finishes its select, from wherever the earlier launches left it:
//   - the bracket pass counted every digit there is (narrow bracket): one pick;
//   - a level gathered the keys that share the resolved digits: their remaining digits are resolved in LDS;
//   - otherwise the workgroup reads the source itself, digit by digit (histogram in LDS, no
//     device-scope atomic), and as soon as KEEP or fewer keys share the resolved digits it gathers
//     them into LDS and finishes there, so a short source is read twice, not once per digit.  This is also the
//     (slow, exact) way out for a long source that SEL_LEVELS levels did not resolve: more than
//     GATHER_CAP different keys of one array inside 2^-36 of its bracket.
// Measured thresholds for "short" (2^12 .. 2^19 keys): a thousand 2^20-row arrays encode in
// 4.73-4.85 ms whatever it is; 1152 ragged columns of up to 2^21 rows 6.31 ms up to 2^14 against
// 6.55 ms at 2^19; 96 columns of 2^22 rows 2.37 against 2.54 ms.
constexpr uint64_t SMALL_SRC_MAX = 1ull << 14;  // default of the `small_max` argument
constexpr uint32_t KEEP = GATHER_CAP;  // 16 KiB: with the histogram and the scan, four workgroups fit a CU
__global__ __launch_bounds__(THREADS) void enc_finish_select_kernel(EncodeJob *__restrict__ jobs, const uint32_t *__restrict__ list,
                                                                     uint32_t n_list) {
    if (blockIdx.x >= n_list) return;
    const uint32_t a = list[blockIdx.x];
    EncodeJob &job = jobs[a];
    const uint32_t passes = job.passes, bits = job.key_bits, mode = job.mode;
    if (passes == 0 || mode == SEL_DONE) return;
    __shared__ uint32_t lh[BINS];
    __shared__ uint64_t keep[KEEP];
    __shared__ BinPickShared sh;
    __shared__ uint32_t keep_n;
    __shared__ unsigned long long above_all;
    const uint32_t tid = threadIdx.x;
    const bool even = (job.n & 1) == 0;
    const uint32_t at = job.pending ? (uint32_t)SEL_LEVELS : 0u;  // the last state written
    EncodeJob::SelState s = job.S[at];
    if (s.action == ACT_DONE) return;
    if (tid == 0) above_all = ~0ull;
    uint64_t next_key = ~0ull;
    if (s.action == ACT_HIST && at > 0 && s.lv > 0 && job.match_min[at - 1] == job.match_max[at - 1]) {
        if (tid == 0) finish_run(job, s, job.match_min[at - 1]);  // the last level counted a digit of keys that are all the same
        return;
    }
    if (s.action == ACT_HIST) s = descend(s, at == 0 ? job.hist : job.lvl_hist + (size_t)(at - 1) * BINS, &sh, &next_key, bits);
    // the smallest key above the keys that share the resolved digits, as far as earlier launches know
    auto above_so_far = [&]() -> uint64_t {
        const uint64_t x = (uint64_t)job.above_key < job.next_above ? (uint64_t)job.above_key : job.next_above;
        return (uint64_t)above_all < x ? (uint64_t)above_all : x;
    };
    // keep[0, cnt) are the keys that share the s.lv resolved digits, s.k the rank of a among them:
    // the remaining digits are resolved by the same count-and-descend steps, on the keys in LDS
    // (2-3 us a digit; sorting the up to 2048 keys took 20-40 us, for two of them).
    __shared__ unsigned long long lone;
    auto finish_kept = [&](uint32_t cnt) {
        __syncthreads();  // keep[] is written
        uint64_t key_a = 0;
        bool have_a = false;
        while (!have_a) {  // uniform
            if (s.lv == passes) { key_a = s.prefix; have_a = true; break; }
            if (s.cnt == 1) {  // one key left under the prefix: it is a, whatever its lower digits
                for (uint32_t i = tid; i < cnt; i += THREADS)
                    if (shares_prefix(keep[i], s, bits)) lone = keep[i];
                __syncthreads();
                key_a = lone; have_a = true; break;
            }
            const uint32_t shift = digit_shift(bits, s.lv);
            for (uint32_t b = tid; b < (uint32_t)BINS; b += THREADS) lh[b] = 0;
            __syncthreads();
            for (uint32_t i = tid; i < cnt; i += THREADS) {
                const uint64_t key = keep[i];
                if (shares_prefix(key, s, bits)) atomicAdd(&lh[(uint32_t)(key >> shift) & (BINS - 1)], 1u);
            }
            __syncthreads();
            uint64_t unused;
            s.action = ACT_HIST;
            s = descend(s, lh, &sh, &unused, bits);
        }
        // b for an even count: a again if rank k + 1 is still one of the keys equal to a, else the
        // smallest key above a, inside the kept keys or above everything that shared their digits
        const bool b_is_a = !even || (s.lv == passes && s.k + 1 < s.cnt);
        if (!b_is_a) {
            uint64_t above = ~0ull;
            for (uint32_t i = tid; i < cnt; i += THREADS) {
                const uint64_t key = keep[i];
                if (key > key_a && key < above) above = key;
            }
            above = wave_min_u64(above);
            if ((tid & 63) == 0 && above != ~0ull) atomicMin(&above_all, (unsigned long long)above);
            __syncthreads();
        }
        if (tid == 0) finish_select(job, key_a, b_is_a ? key_a : above_so_far());
    };
    if (s.action == ACT_GATHER) {  // a level left the keys in `small`
        const uint32_t cnt = (uint32_t)s.cnt;  // == the sum of job.small_n
        uint32_t m[GATHER_SEGS], at = 0;
#pragma unroll
        for (uint32_t seg = 0; seg < GATHER_SEGS; seg++) m[seg] = job.small_n[seg];  // one load for all of them
        const uint64_t *small = job.small;
#pragma unroll
        for (uint32_t seg = 0; seg < GATHER_SEGS; seg++) {  // uniform
            for (uint32_t i = tid; i < m[seg]; i += THREADS)
                if (at + i < KEEP) keep[at + i] = small[(size_t)seg * GATHER_CAP + i];
            at += m[seg];
        }
        finish_kept(cnt);
        return;
    }
    while (true) {  // uniform
        if (s.lv == passes) {
            // a wide last digit's `above` (keys above the earlier digits) was collected with its histogram
            __syncthreads();
            if (tid == 0) {
                if (next_key == ~0ull) next_key = above_so_far();
                uint64_t key_b = s.prefix;
                if (even && s.k + 1 >= s.cnt) key_b = next_key;
                finish_select(job, s.prefix, key_b);
            }
            return;
        }
        if (s.cnt <= (uint64_t)KEEP) {
            // gather the keys that share the resolved digits and finish on them
            if (tid == 0) keep_n = 0;
            __syncthreads();
            uint64_t above = ~0ull;
            for_each_select_key(job, mode, [&](uint64_t key) {
                if (shares_prefix(key, s, bits)) {
                    const uint32_t slot = atomicAdd(&keep_n, 1u);
                    if (slot < KEEP) keep[slot] = key;  // always true while nobody writes the source
                } else if (above_prefix(key, s, bits)) {
                    above = min(above, key);
                }
            });
            above = wave_min_u64(above);
            if ((tid & 63) == 0 && above != ~0ull) atomicMin(&above_all, (unsigned long long)above);
            finish_kept((uint32_t)s.cnt);  // == keep_n
            return;
        }
        // count the next digit of the keys that share the resolved ones
        const uint32_t shift = digit_shift(bits, s.lv);
        const bool last_digit = s.lv + 1 == passes;
        __syncthreads();
        for (uint32_t b = tid; b < (uint32_t)BINS; b += THREADS) lh[b] = 0;
        __syncthreads();
        uint64_t above = ~0ull;
        for_each_select_key(job, mode, [&](uint64_t key) {
            if (shares_prefix(key, s, bits)) atomicAdd(&lh[(uint32_t)(key >> shift) & (BINS - 1)], 1u);
            else if (last_digit && above_prefix(key, s, bits)) above = min(above, key);
        });
        if (last_digit) {
            above = wave_min_u64(above);
            if ((tid & 63) == 0 && above != ~0ull) atomicMin(&above_all, (unsigned long long)above);
        }
        __syncthreads();
        s.action = ACT_HIST;
        s = descend(s, lh, &sh, &next_key, bits);
    }
}

// ---- 4. width of wide-span arrays -------------------------------------------------------------

__global__ __launch_bounds__(THREADS) void enc_or_kernel(EncodeJob *__restrict__ jobs,
                                                          const uint64_t *__restrict__ tile_start, uint32_t nd,
                                                          uint64_t ntiles, const uint32_t *__restrict__ or_flag) {
    // set by the set-up kernel when an array may need this pass: without it the tens of thousands of
    // workgroups of a large batch each looked up their arrays to find that none does (41 us)
    if (*or_flag == 0) return;
    for_each_array_run(tile_start, nd, ntiles, [&](uint32_t a, uint64_t a_first, uint64_t t0, uint64_t t1) {
        const EncodeJob &job = jobs[a];
        if (!job.need_or) return;
        const uint64_t n = job.n;
        const uint64_t mid = (uint64_t)job.midpoint;
        uint64_t acc = 0;
        for (uint64_t t = t0; t < t1; ++t) {
            const uint64_t v0 = (t - a_first) * TILE;
            int64_t x[2 * PAIRS];
            load_tile(job, v0, n, (int64_t)mid, x);  // fill = mid -> contributes 0 (mid = 0 when plain)
#pragma unroll
            for (int e = 0; e < 2 * PAIRS; ++e) acc |= to_stream(job, x[e], mid);
        }
        acc = wave_or_u64(acc);
        // every wave would queue on this one word: look first (the bits settle after a few waves, and
        // a stale look only costs an atomic that changes nothing)
        if ((threadIdx.x & 63) == 0 && (acc & ~__hip_atomic_load(&jobs[a].or_bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)))
            atomicOr((unsigned long long *)&jobs[a].or_bits, (unsigned long long)acc);
    });
}

// ---- 5. header + pack ----------------------------------------------------------------------

// Framed output (write_int_array_compressed, binary.rs:639-645): the arrays of a submission become
// back-to-back frames `u32 count | u32 clen | E` in one buffer.  Sizes are known once the widths
// are, so one workgroup scans them, points every job's `out` at its place behind its frame header,
// writes the 8 header bytes and the offsets (offsets[nd] = total).  A frame that would pass `cap`
// gets capacity 0, which the finalize kernel turns into ZBRA_ERR_CAPACITY.  Frame ends only grow,
// so every frame after the first that does not fit fails too: all of them are reported as the
// empty range [stop, stop) at the end of the last frame that fits, and offsets[nd] = stop is the
// number of bytes actually laid out.
__global__ __launch_bounds__(1024) void enc_frame_layout_kernel(EncodeJob *__restrict__ jobs, uint32_t nd, uint8_t *out_base,
                                                                uint64_t cap, uint64_t *__restrict__ offsets) {
    // One workgroup of 256 or 1024 lanes; lane t owns the consecutive arrays [t * per, t * per + per):
    // it adds up their frame lengths, the lanes' totals are scanned once, and it then lays its arrays
    // out from its base.  What bounds it for tens of thousands of arrays is one CU reading and writing
    // that many job records (0.6 ms for 65536), not the scan.
    __shared__ uint64_t scan[1024];
    __shared__ uint64_t stop;
    const uint32_t tid = threadIdx.x, nt = blockDim.x;
    const uint32_t per = (nd + nt - 1) / nt;
    const uint32_t a0 = tid * per < nd ? tid * per : nd, a1 = a0 + per < nd ? a0 + per : nd;
    auto e_len_of = [&](const EncodeJob &j) -> uint64_t {
        if (!j.n) return 0;
        const uint32_t w = j.need_or ? width_of(j.or_bits) : j.width;
        return 13 + (w >= 32 ? j.n * 8 : (j.n * w + 7) / 8);
    };
    if (tid == 0) stop = ~0ull;
    uint64_t sum = 0;
    for (uint32_t a = a0; a < a1; a++) sum += 8 + e_len_of(jobs[a]);
    scan[tid] = sum;
    __syncthreads();
    for (uint32_t d = 1; d < nt; d <<= 1) {  // inclusive scan of the lanes' totals
        const uint64_t add = tid >= d ? scan[tid - d] : 0;
        __syncthreads();
        scan[tid] += add;
        __syncthreads();
    }
    uint64_t off = scan[tid] - sum;
    for (uint32_t a = a0; a < a1; a++) {
        EncodeJob &j = jobs[a];
        const uint64_t e_len = e_len_of(j), end = off + 8 + e_len;
        const bool fits = end <= cap;
        offsets[a] = off;
        if (a == nd - 1) offsets[nd] = end;
        if (!fits && off <= cap) stop = off;  // the first frame that does not fit: the one that starts inside cap
        if (fits) {
            uint8_t *f = out_base + off;
            const uint32_t cnt = (uint32_t)j.n, clen = (uint32_t)e_len;  // `as u32`, binary.rs:640-642
            for (int b = 0; b < 4; b++) { f[b] = (uint8_t)(cnt >> (8 * b)); f[4 + b] = (uint8_t)(clen >> (8 * b)); }
            j.out = f + 8;
            j.cap = e_len;
        } else {
            j.out = nullptr;
            j.cap = 0;
            j.frame_dropped = 1;  // an empty array has no E whose capacity could fail: its 8-byte frame did
        }
        off = end;
    }
    __syncthreads();
    const uint64_t s = stop;
    if (s != ~0ull)
        for (uint32_t a = tid; a <= nd; a += nt)
            if (offsets[a] > s) offsets[a] = s;
}

// Writes midpoint | packed_len | width (compression.rs:131, 287-288) and the result words.
__global__ void enc_finalize_kernel(EncodeJob *__restrict__ jobs, uint32_t nd, zbra_dev_result *results) {
    const uint32_t a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= nd) return;
    EncodeJob &j = jobs[a];
    zbra_dev_result r;
    r.status = ZBRA_OK; r.width = 0; r.out_count = 0; r.midpoint = 0; r.packed_len = 0;
    if (j.n == 0) {                       // compression.rs:272-274: empty in, empty out
        j.width = WIDTH_SKIP;
        if (j.frame_dropped) r.status = ZBRA_ERR_CAPACITY;
    } else {
        if (j.need_or) j.width = width_of(j.or_bits);
        const uint32_t w = j.width;
        const uint64_t payload = w >= 32 ? j.n * 8 : (j.n * w + 7) / 8;
        const uint64_t packed_len = 1 + payload;
        r.width = w; r.midpoint = j.midpoint; r.packed_len = packed_len; r.out_count = 12 + packed_len;
        if (j.cap < 12 + packed_len) {
            r.status = ZBRA_ERR_CAPACITY; r.out_count = 0;
            j.width = WIDTH_SKIP;
        } else {
            const uint64_t m = (uint64_t)j.midpoint;
            for (int b = 0; b < 8; b++) j.out[b] = (uint8_t)(m >> (8 * b));
            const uint32_t pl = (uint32_t)packed_len;  // `as u32`, compression.rs:288
            for (int b = 0; b < 4; b++) j.out[8 + b] = (uint8_t)(pl >> (8 * b));
            j.out[12] = (uint8_t)w;
        }
    }
    if (results) results[a] = r;
}

// Bits [lo, lo + 32) of a tile's packed stream, lo possibly negative or past the end; zz holds
// the tile's LEN zig-zag values (0 past the last value).  inv = ceil(2^32 / w).
template <int LEN = TILE>
__device__ __forceinline__ uint32_t stream_dword(const uint32_t *zz, int32_t lo, uint32_t w, uint64_t inv) {
    int32_t i, pos;  // value i starts at bit pos relative to lo
    if (lo <= 0) { i = 0; pos = -lo; }
    else { i = (int32_t)(((uint64_t)(uint32_t)lo * inv) >> 32); pos = i * (int32_t)w - lo; }
    uint32_t acc = 0;
    while (pos < 32 && i < LEN) {
        const uint32_t z = zz[i];
        acc |= pos >= 0 ? z << pos : z >> (-pos);
        pos += (int32_t)w;
        i++;
    }
    return acc;
}

// Stages the tile's zig-zag values in LDS (all below 2^31 because w <= 31; 0 past the last value).
__device__ __forceinline__ void stage_zigzag(const EncodeJob &job, uint64_t v0, uint32_t nv, uint32_t *zz) {
    const uint32_t tid = threadIdx.x;
    const uint64_t mid = (uint64_t)job.midpoint;
    int64_t x[2 * PAIRS];
    load_tile(job, v0, v0 + nv, (int64_t)mid, x);
    if (job.kind != ZBRA_ELEM_U32) {
#pragma unroll
        for (int it = 0; it < PAIRS; ++it) {
            uint2 z;
            z.x = (uint32_t)to_stream(job, x[2 * it], mid);
            z.y = (uint32_t)to_stream(job, x[2 * it + 1], mid);
            *(uint2 *)(zz + it * THREADS * 2 + tid * 2) = z;
        }
    } else {
#pragma unroll
        for (int q = 0; q < PAIRS / 2; ++q) {
            uint4 z;
            z.x = (uint32_t)to_stream(job, x[4 * q], mid);
            z.y = (uint32_t)to_stream(job, x[4 * q + 1], mid);
            z.z = (uint32_t)to_stream(job, x[4 * q + 2], mid);
            z.w = (uint32_t)to_stream(job, x[4 * q + 3], mid);
            *(uint4 *)(zz + q * THREADS * 4 + tid * 4) = z;
        }
    }
    __syncthreads();
}

// Widths up to NARROW_W: the tile's output is at most 2 KiB.  One dword per lane keeps every lane
// busy (16-B chunks would leave most of them idle while the rest walk 32 / w values per dword,
// four dwords each): the u32 tag column of the sum-typed table packs in 0.30 ms instead of 0.49.
constexpr uint32_t NARROW_W = 8;
// The staged tile (LEN values in zz, the first nv of them real) written out at width w <= NARROW_W.
template <int LEN>
__device__ __forceinline__ void emit_tile_narrow(const EncodeJob &job, uint32_t w, uint64_t v0, uint32_t nv, const uint32_t *zz) {
    const uint32_t tid = threadIdx.x;
    const uintptr_t G = (uintptr_t)job.out + 13 + ((v0 * w) >> 3);
    const uint32_t nbytes = (nv * w + 7u) >> 3;
    const uintptr_t Gend = G + nbytes;
    const uintptr_t G4 = G & ~(uintptr_t)3;
    const uint32_t off4 = (uint32_t)(G - G4);
    const uint32_t ndw = (off4 + nbytes + 3u) >> 2;
    const uint64_t inv = (0x100000000ull + w - 1) / w;
    for (uint32_t d = tid; d < ndw; d += THREADS) {
        const int32_t lo = (int32_t)(d * 32u) - (int32_t)(off4 * 8u);
        uint32_t v;
        if ((w & (w - 1)) == 0 && lo >= 0 && (uint32_t)lo + 32u <= (uint32_t)LEN * w) {
            // Widths 1, 2, 4 and 8 (booleans, constant columns, enum tags): the dword's 32 / w values lie
            // wholly inside the tile and start where the dword does (lo is a multiple of 8), so they are
            // read sixteen or eight bytes at a time, without the loop: few lanes have a dword at the
            // narrowest widths, and the rest of the workgroup waits for them.
            const uint32_t i0 = (uint32_t)lo / w;
            v = 0;
            if (w == 1) {
                const uint4 *z4 = (const uint4 *)(zz + i0);  // a multiple of 8
#pragma unroll
                for (int q = 0; q < 8; q++) { const uint4 z = z4[q]; v |= (z.x | z.y << 1 | z.z << 2 | z.w << 3) << (4 * q); }
            } else if (w == 2) {
                const uint4 *z4 = (const uint4 *)(zz + i0);  // a multiple of 4
#pragma unroll
                for (int q = 0; q < 4; q++) { const uint4 z = z4[q]; v |= (z.x | z.y << 2 | z.z << 4 | z.w << 6) << (8 * q); }
            } else if (w == 4) {
                const uint2 *z2 = (const uint2 *)(zz + i0);  // a multiple of 2
#pragma unroll
                for (int q = 0; q < 4; q++) { const uint2 z = z2[q]; v |= (z.x | z.y << 4) << (8 * q); }
            } else {
                v = zz[i0] | zz[i0 + 1] << 8 | zz[i0 + 2] << 16 | zz[i0 + 3] << 24;
            }
        } else {
            v = stream_dword<LEN>(zz, lo, w, inv);
        }
        const uintptr_t addr = G4 + (uintptr_t)d * 4;
        if (addr >= G && addr + 4 <= Gend) {
            __builtin_nontemporal_store(v, gptr_mut<uint32_t>(addr));
        } else {  // head / tail dword shared with a neighbour: only this tile's bytes
#pragma unroll
            for (int b = 0; b < 4; b++) {
                const uintptr_t p = addr + b;
                if (p >= G && p < Gend) *gptr_mut<uint8_t>(p) = (uint8_t)(v >> (8 * b));
            }
        }
    }
    __syncthreads();  // a later tile of this workgroup restages
}
__device__ __forceinline__ void pack_tile_narrow(const EncodeJob &job, uint32_t w, uint64_t v0, uint32_t nv, uint32_t *zz) {
    stage_zigzag(job, v0, nv, zz);
    emit_tile_narrow<TILE>(job, w, v0, nv, zz);
}

// The staged tile written out at width NARROW_W < w < 32.
template <int LEN>
__device__ __forceinline__ void emit_tile_packed(const EncodeJob &job, uint32_t w, uint64_t v0, uint32_t nv, const uint32_t *zz) {
    const uint32_t tid = threadIdx.x;
    const uintptr_t G = (uintptr_t)job.out + 13 + ((v0 * w) >> 3);
    const uint32_t nbytes = (nv * w + 7u) >> 3;
    const uintptr_t Gend = G + nbytes;
    const uintptr_t G0 = G & ~(uintptr_t)15;
    const uint32_t off = (uint32_t)(G - G0);
    const uint32_t nchunks = (off + nbytes + 15u) >> 4;
    const uint64_t inv = (0x100000000ull + w - 1) / w;
    for (uint32_t c = tid; c < nchunks; c += THREADS) {
        const int32_t lo = (int32_t)(c * 128u) - (int32_t)(off * 8u);
        u32x4 v;
        v.x = stream_dword<LEN>(zz, lo, w, inv);
        v.y = stream_dword<LEN>(zz, lo + 32, w, inv);
        v.z = stream_dword<LEN>(zz, lo + 64, w, inv);
        v.w = stream_dword<LEN>(zz, lo + 96, w, inv);
        const uintptr_t addr = G0 + (uintptr_t)c * 16;
        if (addr >= G && addr + 16 <= Gend) {
            __builtin_nontemporal_store(v, gptr_mut<u32x4>(addr));  // streamed out once: -2 % on the whole encode
        } else {  // head / tail chunk shared with a neighbour: only this tile's bytes
            const uint32_t d[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int b = 0; b < 16; b++) {
                const uintptr_t p = addr + b;
                if (p >= G && p < Gend) *gptr_mut<uint8_t>(p) = (uint8_t)(d[b >> 2] >> (8 * (b & 3)));
            }
        }
    }
    __syncthreads();  // a later tile of this workgroup restages
}
__device__ __forceinline__ void pack_tile_packed(const EncodeJob &job, uint32_t w, uint64_t v0, uint32_t nv, uint32_t *zz) {
    stage_zigzag(job, v0, nv, zz);
    emit_tile_packed<TILE>(job, w, v0, nv, zz);
}

// Raw fallback: zig-zag value i as u64 LE at payload + 8*i (compression.rs:143-149).  The tile's
// values are loaded once with coalesced 16-byte loads, staged in LDS as 64-bit zig-zag values, and
// each lane then assembles whole 16-byte chunks of the OUTPUT memory image (which starts 13 bytes
// into E, wherever E is) from the three staged values that overlap it.  Measured against two
// variants without LDS on a 2^27-row full-range column: 367 us for this one, 403 us with two aligned
// 16-byte loads per store and a byte funnel in registers (every input chunk is loaded twice, for
// itself and for its neighbour), 428 us with three 8-byte loads per store.
__device__ __forceinline__ void pack_tile_raw(const EncodeJob &job, uint64_t v0, uint32_t nv, uint64_t *zz64) {
    const uint32_t tid = threadIdx.x;
    const uint64_t mid = (uint64_t)job.midpoint;
    int64_t x[2 * PAIRS];
    load_tile(job, v0, v0 + nv, (int64_t)mid, x);
    static_assert(sizeof(int64_t) == 8, "");
    if (job.kind != ZBRA_ELEM_U32) {
#pragma unroll
        for (int it = 0; it < PAIRS; ++it) {
            u64x2 z = {to_stream(job, x[2 * it], mid), to_stream(job, x[2 * it + 1], mid)};
            *(u64x2 *)(zz64 + it * THREADS * 2 + tid * 2) = z;
        }
    } else {
#pragma unroll
        for (int q = 0; q < PAIRS / 2; ++q)
#pragma unroll
            for (int e = 0; e < 4; e++) zz64[q * THREADS * 4 + tid * 4 + e] = to_stream(job, x[4 * q + e], mid);
    }
    __syncthreads();
    const uintptr_t G = (uintptr_t)job.out + 13 + v0 * 8;
    const uint32_t nbytes = nv * 8u;
    const uintptr_t Gend = G + nbytes;
    const uintptr_t G0 = G & ~(uintptr_t)15;
    const uint32_t off = (uint32_t)(G - G0);
    const uint32_t nchunks = (off + nbytes + 15u) >> 4;
    const int32_t first = -(int32_t)((off + 7u) >> 3);       // value that holds the chunk row's first byte: floor(-off / 8)
    const uint32_t sub = ((8u - (off & 7u)) & 7u) * 8u;      // bit offset of that byte inside it
    for (uint32_t c = tid; c < nchunks; c += THREADS) {
        const int32_t i0 = first + 2 * (int32_t)c;
        uint64_t z[3];
#pragma unroll
        for (int e = 0; e < 3; e++) {
            const int32_t i = i0 + e;
            z[e] = (i >= 0 && (uint32_t)i < nv) ? zz64[i] : 0;
        }
        uint64_t o0 = z[0], o1 = z[1];
        if (sub) {
            o0 = (z[0] >> sub) | (z[1] << (64 - sub));
            o1 = (z[1] >> sub) | (z[2] << (64 - sub));
        }
        const uintptr_t addr = G0 + (uintptr_t)c * 16;
        if (addr >= G && addr + 16 <= Gend) {
            u64x2 v = {o0, o1};
            __builtin_nontemporal_store(v, gptr_mut<u64x2>(addr));
        } else {  // head / tail chunk shared with a neighbour: only this tile's bytes
#pragma unroll
            for (int b = 0; b < 16; b++) {
                const uintptr_t p = addr + b;
                const uint64_t src = b < 8 ? o0 : o1;
                if (p >= G && p < Gend) *gptr_mut<uint8_t>(p) = (uint8_t)(src >> (8 * (b & 7)));
            }
        }
    }
    __syncthreads();  // a later tile of this workgroup restages
}

// frame_of_reference_encode alone: copies each array's midpoint to its result words.
__global__ void enc_report_median_kernel(const EncodeJob *__restrict__ jobs, uint32_t nd, zbra_dev_result *results) {
    const uint32_t a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= nd) return;
    zbra_dev_result r;
    r.status = ZBRA_OK; r.width = jobs[a].need_or ? 0 : jobs[a].width; r.out_count = jobs[a].n;
    r.midpoint = jobs[a].midpoint; r.packed_len = 0;
    results[a] = r;
}

// FINALIZE (a single array with values, not framed): no enc_finalize_kernel ran before this launch.
// Every workgroup derives the width and the capacity check itself (a few scalar operations on the
// job it loads anyway), and the workgroup of tile 0 writes the 13 header bytes and the result words:
// one dependent launch less per encode.
template <bool FINALIZE>
__global__ __launch_bounds__(THREADS) void enc_pack_kernel(const EncodeJob *__restrict__ jobs,
                                                            const uint64_t *__restrict__ tile_start,
                                                            const uint32_t *__restrict__ tile_map, uint32_t nd,
                                                            uint64_t ntiles, uint32_t xcd_chunked, zbra_dev_result *results) {
    __shared__ __attribute__((aligned(16))) uint64_t zz64[TILE];  // 64-bit zig-zag values of a raw tile; the packed paths use half of it
    uint32_t *zz = (uint32_t *)zz64;
    // xcd_chunked implies gridDim.x == ntiles: exactly one trip through the loop
    uint64_t t = xcd_chunked ? xcd_chunked_tile(blockIdx.x, ntiles) : blockIdx.x;
    for (; t < ntiles; t += gridDim.x) {
        const uint32_t a = nd > 1 ? tile_map[t] : 0;
        const EncodeJob &job = jobs[a];
        uint32_t w = job.width;
        if (FINALIZE) {  // what enc_finalize_kernel does for an array with values
            if (job.need_or) w = width_of(job.or_bits);
            const uint64_t packed_len = 1 + (w >= 32 ? job.n * 8 : (job.n * w + 7) / 8);
            const bool fits = job.cap >= 12 + packed_len;
            if (t == 0 && threadIdx.x == 0) {
                zbra_dev_result r;
                r.status = fits ? ZBRA_OK : ZBRA_ERR_CAPACITY; r.width = w; r.out_count = fits ? 12 + packed_len : 0;
                r.midpoint = job.midpoint; r.packed_len = packed_len;
                if (results) results[0] = r;
                if (fits) {
                    const uint64_t m = (uint64_t)job.midpoint;
                    for (int b = 0; b < 8; b++) job.out[b] = (uint8_t)(m >> (8 * b));
                    const uint32_t pl = (uint32_t)packed_len;  // `as u32`, compression.rs:288
                    for (int b = 0; b < 4; b++) job.out[8 + b] = (uint8_t)(pl >> (8 * b));
                    job.out[12] = (uint8_t)w;
                }
            }
            if (!fits) return;
        } else if (w == WIDTH_SKIP) {
            continue;
        }
        const uint64_t v0 = (t - tile_start[a]) * TILE;
        const uint64_t left = job.n - v0;
        const uint32_t nv = left < (uint64_t)TILE ? (uint32_t)left : (uint32_t)TILE;
        if (w - (NARROW_W + 1) < 31u - NARROW_W) pack_tile_packed(job, w, v0, nv, zz);  // NARROW_W < w < 32
        else if (w <= NARROW_W) pack_tile_narrow(job, w, v0, nv, zz);
        else pack_tile_raw(job, v0, nv, zz64);
    }
}

// The pack pass of a submission whose arrays are all u32 (enum tags) and equally long, or a single
// one.  A tile of 2048 tags is 8 KiB of input: with one such tile per workgroup a CU has half the
// bytes in flight that an int64 column gives it, and the pass read 2^29 tags at 4.2 TB/s where
// int64 columns reach 6.  Here a workgroup takes 4096 consecutive tags of one array (16 KiB, four
// 16-byte loads per lane) and works in 32 bits: below width 32 the delta to the midpoint fits an
// int32, so zig-zag(v - mid) is computed on the low halves.  Tiles are numbered per array
// (per_array of them each): no table.  Raw widths (a tag column spanning 2^31 or more) take the
// generic raw routine, half a tile at a time.
constexpr int TILE32 = 2 * TILE;
template <bool FINALIZE>
__global__ __launch_bounds__(THREADS) void enc_pack_u32_kernel(const EncodeJob *__restrict__ jobs, uint32_t per_array, uint64_t per_array_inv,
                                                                uint64_t ntiles, uint32_t xcd_chunked, zbra_dev_result *results) {
    __shared__ __attribute__((aligned(16))) uint64_t zz64[TILE];  // 4096 32-bit zig-zag values, or a raw half tile's 64-bit ones
    uint32_t *zz = (uint32_t *)zz64;
    const uint32_t tid = threadIdx.x;
    const uint64_t t = xcd_chunked ? xcd_chunked_tile(blockIdx.x, ntiles) : blockIdx.x;
    if (t >= ntiles) return;
    // t / per_array by a multiply-high with ceil(2^64 / per_array) (exact below 2^32 tiles); one tile per array: t itself
    const uint32_t a = FINALIZE ? 0u : per_array == 1 ? (uint32_t)t : (uint32_t)(((unsigned __int128)(uint32_t)t * per_array_inv) >> 64);
    const EncodeJob &job = jobs[a];
    uint32_t w = job.width;
    const uint64_t ta = t - (uint64_t)a * per_array;
    if (FINALIZE) {  // as in enc_pack_kernel<true>
        if (job.need_or) w = width_of(job.or_bits);
        const uint64_t packed_len = 1 + (w >= 32 ? job.n * 8 : (job.n * w + 7) / 8);
        const bool fits = job.cap >= 12 + packed_len;
        if (ta == 0 && tid == 0) {
            zbra_dev_result r;
            r.status = fits ? ZBRA_OK : ZBRA_ERR_CAPACITY; r.width = w; r.out_count = fits ? 12 + packed_len : 0;
            r.midpoint = job.midpoint; r.packed_len = packed_len;
            if (results) results[0] = r;
            if (fits) {
                const uint64_t m = (uint64_t)job.midpoint;
                for (int b = 0; b < 8; b++) job.out[b] = (uint8_t)(m >> (8 * b));
                const uint32_t pl = (uint32_t)packed_len;  // `as u32`, compression.rs:288
                for (int b = 0; b < 4; b++) job.out[8 + b] = (uint8_t)(pl >> (8 * b));
                job.out[12] = (uint8_t)w;
            }
        }
        if (!fits) return;
    } else if (w == WIDTH_SKIP) {
        return;
    }
    const uint64_t v0 = ta * TILE32;
    if (v0 >= job.n) return;  // (arrays of one tile count: cannot happen; kept as the guard of the loads below)
    const uint64_t left = job.n - v0;
    const uint32_t nv = left < (uint64_t)TILE32 ? (uint32_t)left : (uint32_t)TILE32;
    if (w >= 32) {
        pack_tile_raw(job, v0, nv < (uint32_t)TILE ? nv : (uint32_t)TILE, zz64);
        if (nv > (uint32_t)TILE) pack_tile_raw(job, v0 + TILE, nv - TILE, zz64);
        return;
    }
    const uint32_t mid = (uint32_t)(uint64_t)job.midpoint;  // the median of u32 values is one (or lies between two)
    constexpr int Q = TILE32 / (THREADS * 4);  // 16-byte loads per lane
    const ZBRA_GLOBAL uint32_t *src = gptr<uint32_t>(job.values) + v0;
    u32x4 v[Q];
    if (nv == (uint32_t)TILE32) {
#pragma unroll
        for (int q = 0; q < Q; ++q) v[q] = __builtin_nontemporal_load((const ZBRA_GLOBAL u32x4 *)src + q * THREADS + tid);
    } else {
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const uint32_t i = (uint32_t)q * THREADS * 4 + tid * 4;
            v[q].x = i < nv ? src[i] : mid; v[q].y = i + 1 < nv ? src[i + 1] : mid;
            v[q].z = i + 2 < nv ? src[i + 2] : mid; v[q].w = i + 3 < nv ? src[i + 3] : mid;  // mid -> zig-zag 0
        }
    }
    auto zz32 = [mid](uint32_t x) { const uint32_t d = x - mid; return (d << 1) ^ (uint32_t)((int32_t)d >> 31); };
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        uint4 z;
        z.x = zz32(v[q].x); z.y = zz32(v[q].y); z.z = zz32(v[q].z); z.w = zz32(v[q].w);
        *(uint4 *)(zz + q * THREADS * 4 + tid * 4) = z;
    }
    __syncthreads();
    if (w <= NARROW_W) emit_tile_narrow<TILE32>(job, w, v0, nv, zz);
    else emit_tile_packed<TILE32>(job, w, v0, nv, zz);
}

}  // namespace zbra
