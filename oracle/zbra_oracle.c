/*
 * zbra_oracle.c — CPU restatement of zbra-core's integer column codec.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the checker that the HIP path is
 * compared against; it is also the timed "port" CPU baseline in bench.py.
 * Nothing in the shipped library (zbra_amd/) links, loads or calls it.
 *
 * PARITY PINNING: the reference (Rust, zbra-core) cannot be built or run in
 * this environment and ships no golden bytes, so the byte layout produced
 * here is pinned by the reference's SOURCE TEXT plus the properties its own
 * tests assert (round trips, bit widths, packed sizes), not by bytes the
 * reference itself produced: "parity unpinned" against reference execution.
 * An independently written numpy restatement (zbra_oracle_np.py) and the
 * hand-derived known-answer vectors in tests/golden/ must agree with it.
 *
 * Every function follows one reference function, statement by statement,
 * including its intermediate buffers, so that its cost profile is that of the
 * reference's scalar single-threaded design
 * (zbra-core/benches/README.md:166-170).
 *
 * All arithmetic is 64-bit two's complement with wrap-around; unsigned types
 * are used wherever the reference uses wrapping_* so that no C signed
 * overflow is ever evaluated.
 */
#include <stdint.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

#define ZO_OK 0
#define ZO_ERR_SHORT_HEADER 1 /* "Invalid compressed data length"      compression.rs:300-304 */
#define ZO_ERR_SHORT_PACKED 2 /* "Insufficient data for packed array"  compression.rs:312-316 */
#define ZO_ERR_TRUNCATED 3    /* reference underflows u32 here (compression.rs:229): unspecified */
#define ZO_ERR_NOMEM 5

static int cmp_i64(const void *a, const void *b) {
    int64_t x = *(const int64_t *)a, y = *(const int64_t *)b;
    return (x > y) - (x < y);
}

static inline int64_t wrapping_sub(int64_t a, int64_t b) { return (int64_t)((uint64_t)a - (uint64_t)b); }
static inline int64_t wrapping_add(int64_t a, int64_t b) { return (int64_t)((uint64_t)a + (uint64_t)b); }

/* Median rule of frame_of_reference_encode, compression.rs:57-68.
 * `sorted` is ascending in SIGNED order.  Rust's `/` on i64 truncates toward
 * zero, as C99's does. */
static int64_t midpoint_of_sorted(const int64_t *sorted, size_t n) {
    if (n % 2 == 0) {
        int64_t a = sorted[n / 2 - 1];
        int64_t b = sorted[n / 2];
        return wrapping_add(a, wrapping_sub(b, a) / 2);
    }
    return sorted[n / 2];
}

/* frame_of_reference_encode — compression.rs:52-74.
 * Returns 0 and *midpoint = 0 for an empty input (compression.rs:53-55). */
int zo_frame_of_reference_encode(const int64_t *values, size_t n, int64_t *midpoint, int64_t *deltas) {
    if (n == 0) {
        *midpoint = 0;
        return ZO_OK;
    }
    int64_t *sorted = (int64_t *)malloc(n * sizeof(int64_t)); /* values.to_vec()  :58 */
    if (!sorted) return ZO_ERR_NOMEM;
    memcpy(sorted, values, n * sizeof(int64_t));
    qsort(sorted, n, sizeof(int64_t), cmp_i64);               /* sorted.sort()     :59 */
    int64_t mid = midpoint_of_sorted(sorted, n);
    free(sorted);
    for (size_t i = 0; i < n; i++) deltas[i] = wrapping_sub(values[i], mid); /* :71 */
    *midpoint = mid;
    return ZO_OK;
}

/* frame_of_reference_decode — compression.rs:77-82. */
void zo_frame_of_reference_decode(int64_t midpoint, const int64_t *deltas, size_t n, int64_t *out) {
    for (size_t i = 0; i < n; i++) out[i] = wrapping_add(midpoint, deltas[i]);
}

/* zig_zag_encode — compression.rs:88-98: (n as u64) << 1 ^ (n >> 63) as u64,
 * with >> arithmetic on i64. */
void zo_zig_zag_encode(const int64_t *values, size_t n, uint64_t *out) {
    for (size_t i = 0; i < n; i++) {
        uint64_t shifted = (uint64_t)values[i] << 1;
        uint64_t sign_bit = values[i] < 0 ? ~(uint64_t)0 : 0; /* n >> 63, arithmetic */
        out[i] = shifted ^ sign_bit;
    }
}

/* zig_zag_decode — compression.rs:101-111. */
void zo_zig_zag_decode(const uint64_t *values, size_t n, int64_t *out) {
    for (size_t i = 0; i < n; i++) {
        uint64_t shifted = values[i] >> 1;
        uint64_t sign_mask = (uint64_t)0 - (values[i] & 1); /* -((n & 1) as i64) */
        out[i] = (int64_t)(shifted ^ sign_mask);
    }
}

/* Bit width chosen by bp64_pack, compression.rs:123-128. */
static unsigned width_of_max(uint64_t max_value) {
    if (max_value == 0) return 1;
    return 64u - (unsigned)__builtin_clzll(max_value);
}

/* bp64_pack — compression.rs:117-178.  `out` must hold 1 + 8*n bytes.
 * Returns the number of bytes written (0 for an empty input, :118-120). */
size_t zo_bp64_pack(const uint64_t *values, size_t n, uint8_t *out) {
    if (n == 0) return 0;
    uint64_t max_value = 0;
    for (size_t i = 0; i < n; i++)
        if (values[i] > max_value) max_value = values[i];
    unsigned bit_width = width_of_max(max_value);
    size_t pos = 0;
    out[pos++] = (uint8_t)bit_width;                          /* :131 */
    if (bit_width >= 32) {                                    /* :143-149 raw fallback */
        for (size_t i = 0; i < n; i++) {
            uint64_t v = values[i];
            for (int b = 0; b < 8; b++) out[pos++] = (uint8_t)(v >> (8 * b));
        }
        return pos;
    }
    uint64_t bit_buffer = 0;                                  /* :152-170 */
    unsigned bits_in_buffer = 0;
    uint64_t mask = ((uint64_t)1 << bit_width) - 1;
    for (size_t i = 0; i < n; i++) {
        uint64_t masked = values[i] & mask;
        bit_buffer |= masked << bits_in_buffer;
        bits_in_buffer += bit_width;
        while (bits_in_buffer >= 8) {
            out[pos++] = (uint8_t)bit_buffer;
            bit_buffer >>= 8;
            bits_in_buffer -= 8;
        }
    }
    if (bits_in_buffer > 0) out[pos++] = (uint8_t)bit_buffer; /* :173-175 */
    return pos;
}

/* bp64_unpack — compression.rs:181-233.
 * *out_count is the length of the Vec the reference returns: 0 when `packed`
 * is empty (even if count > 0, :182-184), otherwise `count`.
 * A packed payload shorter than ceil(count*width/8) bytes makes the reference
 * subtract below zero on a u32 (:229) — a panic in debug builds, garbage in
 * release builds.  That case is reported as ZO_ERR_TRUNCATED instead. */
int zo_bp64_unpack(const uint8_t *packed, size_t packed_len, size_t count, uint64_t *out, size_t *out_count) {
    *out_count = 0;
    if (packed_len == 0) return ZO_OK;
    unsigned bit_width = packed[0];
    if (bit_width == 0 || count == 0) {                       /* :187-189 */
        for (size_t i = 0; i < count; i++) out[i] = 0;
        *out_count = count;
        return ZO_OK;
    }
    const uint8_t *data = packed + 1;
    size_t data_len = packed_len - 1;
    if (bit_width >= 32) {                                    /* :195-207, missing bytes -> 0 */
        for (size_t i = 0; i < count; i++) {
            size_t start = i * 8;
            if (start + 8 <= data_len) {
                uint64_t v = 0;
                for (int b = 0; b < 8; b++) v |= (uint64_t)data[start + b] << (8 * b);
                out[i] = v;
            } else {
                out[i] = 0;
            }
        }
        *out_count = count;
        return ZO_OK;
    }
    uint64_t bit_buffer = 0;                                  /* :209-230 */
    unsigned bits_in_buffer = 0;
    size_t byte_index = 0;
    uint64_t mask = ((uint64_t)1 << bit_width) - 1;
    for (size_t i = 0; i < count; i++) {
        while (bits_in_buffer < bit_width && byte_index < data_len) {
            bit_buffer |= (uint64_t)data[byte_index] << bits_in_buffer;
            bits_in_buffer += 8;
            byte_index += 1;
        }
        if (bits_in_buffer < bit_width) return ZO_ERR_TRUNCATED;
        out[i] = bit_buffer & mask;
        bit_buffer >>= bit_width;
        bits_in_buffer -= bit_width;
    }
    *out_count = count;
    return ZO_OK;
}

/* Upper bound of compress_int_array's output: 8 midpoint + 4 length + 1 width + 8 per value. */
size_t zo_compress_bound(size_t n) { return n == 0 ? 0 : 13 + 8 * n; }

/* compress_int_array — compression.rs:271-292.  Keeps the reference's three
 * intermediate vectors.  Returns bytes written (0 for an empty input). */
size_t zo_compress_int_array(const int64_t *values, size_t n, uint8_t *out) {
    if (n == 0) return 0;
    int64_t *deltas = (int64_t *)malloc(n * sizeof(int64_t));
    uint64_t *unsigned_values = (uint64_t *)malloc(n * sizeof(uint64_t));
    uint8_t *packed = (uint8_t *)malloc(1 + 8 * n);
    if (!deltas || !unsigned_values || !packed) {
        free(deltas); free(unsigned_values); free(packed);
        return (size_t)-1;
    }
    int64_t midpoint;
    zo_frame_of_reference_encode(values, n, &midpoint, deltas);
    zo_zig_zag_encode(deltas, n, unsigned_values);
    size_t packed_len = zo_bp64_pack(unsigned_values, n, packed);
    uint64_t m = (uint64_t)midpoint;
    for (int b = 0; b < 8; b++) out[b] = (uint8_t)(m >> (8 * b));          /* :287 */
    uint32_t pl = (uint32_t)packed_len;                                     /* :288, `as u32` */
    for (int b = 0; b < 4; b++) out[8 + b] = (uint8_t)(pl >> (8 * b));
    memcpy(out + 12, packed, packed_len);                                   /* :289 */
    free(deltas); free(unsigned_values); free(packed);
    return 12 + packed_len;
}

/* decompress_int_array — compression.rs:295-330. */
int zo_decompress_int_array(const uint8_t *data, size_t len, size_t count, int64_t *out, size_t *out_count) {
    *out_count = 0;
    if (len == 0) return ZO_OK;                                             /* :296-298 */
    if (len < 12) return ZO_ERR_SHORT_HEADER;                               /* :300-304 */
    uint64_t m = 0;
    for (int b = 0; b < 8; b++) m |= (uint64_t)data[b] << (8 * b);
    int64_t midpoint = (int64_t)m;
    uint32_t pl = 0;
    for (int b = 0; b < 4; b++) pl |= (uint32_t)data[8 + b] << (8 * b);
    size_t packed_len = pl;
    if (len < 12 + packed_len) return ZO_ERR_SHORT_PACKED;                  /* :312-316 */
    const uint8_t *packed = data + 12;
    uint64_t *unsigned_values = (uint64_t *)malloc((count ? count : 1) * sizeof(uint64_t));
    int64_t *deltas = (int64_t *)malloc((count ? count : 1) * sizeof(int64_t));
    if (!unsigned_values || !deltas) { free(unsigned_values); free(deltas); return ZO_ERR_NOMEM; }
    size_t got = 0;
    int rc = zo_bp64_unpack(packed, packed_len, count, unsigned_values, &got);
    if (rc == ZO_OK) {
        zo_zig_zag_decode(unsigned_values, got, deltas);
        zo_frame_of_reference_decode(midpoint, deltas, got, out);
        *out_count = got;
    }
    free(unsigned_values); free(deltas);
    return rc;
}

/* Int-array frame of binary.rs:639-654: u32 count | u32 clen | E.
 * `out` must hold 8 + zo_compress_bound(n) bytes.  Returns bytes written. */
size_t zo_write_int_array_compressed(const int64_t *values, size_t n, uint8_t *out) {
    uint32_t cnt = (uint32_t)n;                                             /* `as u32`, :640 */
    for (int b = 0; b < 4; b++) out[b] = (uint8_t)(cnt >> (8 * b));
    size_t clen = zo_compress_int_array(values, n, out + 8);
    uint32_t c32 = (uint32_t)clen;                                          /* :642 */
    for (int b = 0; b < 4; b++) out[4 + b] = (uint8_t)(c32 >> (8 * b));
    return 8 + clen;
}

/* Counter-based splitmix64 used for every synthetic column of the benchmark
 * configurations: sm(seed, i) = finalise(seed + (i+1)*0x9E3779B97F4A7C15). */
uint64_t zo_sm(uint64_t seed, uint64_t i) {
    uint64_t z = seed + (i + 1) * 0x9E3779B97F4A7C15ull;
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull;
    z ^= z >> 27; z *= 0x94D049BB133111EBull;
    z ^= z >> 31;
    return z;
}

/* Narrow-range column of the headline workload: 1.7e12 + sm(seed, first+i) mod 4000. */
void zo_gen_narrow(uint64_t seed, uint64_t first, size_t n, int64_t *out) {
    for (size_t i = 0; i < n; i++) out[i] = (int64_t)1700000000000ll + (int64_t)(zo_sm(seed, first + i) % 4000);
}
