/*
 * Drives libzbra_hip.so through its C ABI in exactly the pattern the Rust wrappers of
 * rust/zbra-hip-sys/src/lib.rs use: allocate the output at its bound (or at `count`), call,
 * truncate to what the call reports.  Plain C99, so that nothing but the ABI is between the test and
 * the library.  Every result is compared with the CPU restatement of the reference (the checker).
 * Exit status: 0 = all checks passed, 1 = a check failed, 2 = no GPU.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "zbra_hip.h"

/* the checker (oracle/zbra_oracle.c) */
size_t zo_compress_bound(size_t n);
size_t zo_compress_int_array(const int64_t *values, size_t n, uint8_t *out);
int zo_decompress_int_array(const uint8_t *data, size_t len, size_t count, int64_t *out, size_t *out_count);
uint64_t zo_sm(uint64_t seed, uint64_t i);

static int failures = 0, checks = 0;
#define CHECK(cond)                                                              \
    do {                                                                         \
        checks++;                                                                \
        if (!(cond)) { failures++; printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); } \
    } while (0)

typedef struct { uint8_t *ptr; size_t len; } bytes_t;

/* zbra_hip_sys::compress_int_array */
static int wrap_compress(const int64_t *values, size_t n, bytes_t *out) {
    size_t cap = zbra_hip_encode_bound(n), len = 0;
    uint8_t *buf = (uint8_t *)malloc(cap ? cap : 1);
    int st = zbra_hip_encode_i64(values, n, buf, cap, &len);
    if (st != ZBRA_OK) { free(buf); return st; }
    out->ptr = (uint8_t *)realloc(buf, len ? len : 1); /* Vec::truncate */
    out->len = len;
    return ZBRA_OK;
}

/* zbra_hip_sys::decompress_int_array */
static int wrap_decompress(const uint8_t *data, size_t len, size_t count, int64_t **out, size_t *out_len) {
    int64_t *buf = (int64_t *)calloc(count ? count : 1, sizeof(int64_t));
    size_t got = 0;
    int st = zbra_hip_decode_i64(data, len, count, buf, &got);
    if (st != ZBRA_OK) { free(buf); return st; }
    *out = buf;
    *out_len = got; /* Vec::truncate(got) */
    return ZBRA_OK;
}

static void column(int kind, size_t n, uint64_t seed, int64_t *v) {
    for (size_t i = 0; i < n; i++) {
        uint64_t r = zo_sm(seed, i);
        switch (kind) {
            case 0: v[i] = 1700000000000LL + (int64_t)(r % 4000); break;           /* narrow: width 12 */
            case 1: v[i] = (int64_t)r; break;                                       /* full range: raw */
            case 2: v[i] = (int64_t)(r % 2) * 5000 - 17; break;                     /* two values */
            case 3: v[i] = (int64_t)(i * 3600000ULL) + 4102444800000LL; break;      /* hourly timestamps */
            default: v[i] = -42; break;                                             /* constant */
        }
    }
}

static void single_arrays(void) {
    static const size_t sizes[] = {0, 1, 2, 5, 64, 4097, 100003};
    for (int kind = 0; kind < 5; kind++) {
        for (size_t s = 0; s < sizeof(sizes) / sizeof(sizes[0]); s++) {
            const size_t n = sizes[s];
            int64_t *v = (int64_t *)malloc((n ? n : 1) * sizeof(int64_t));
            column(kind, n, 1000 + kind, v);
            uint8_t *want = (uint8_t *)malloc(zo_compress_bound(n) + 1);
            const size_t want_len = zo_compress_int_array(v, n, want);
            bytes_t e = {0, 0};
            CHECK(wrap_compress(v, n, &e) == ZBRA_OK);
            CHECK(e.len == want_len && (want_len == 0 || memcmp(e.ptr, want, want_len) == 0));
            int64_t *back = 0;
            size_t got = 0;
            CHECK(wrap_decompress(e.ptr, e.len, n, &back, &got) == ZBRA_OK);
            CHECK(got == n && (n == 0 || memcmp(back, v, n * 8) == 0));
            /* count larger than what was encoded for empty data: empty Vec, not an error */
            if (n == 0) {
                int64_t *none = 0;
                CHECK(wrap_decompress(e.ptr, 0, 7, &none, &got) == ZBRA_OK && got == 0);
                free(none);
            }
            free(back); free(e.ptr); free(want); free(v);
        }
    }
    /* the two errors of decompress_int_array keep their messages */
    uint8_t junk[11] = {0};
    int64_t *none = 0;
    size_t got = 0;
    CHECK(wrap_decompress(junk, sizeof(junk), 1, &none, &got) == ZBRA_ERR_SHORT_HEADER);
    CHECK(strcmp(zbra_hip_strerror(ZBRA_ERR_SHORT_HEADER), "Invalid compressed data length") == 0);
    CHECK(strcmp(zbra_hip_strerror(ZBRA_ERR_SHORT_PACKED), "Insufficient data for packed array") == 0);
}

/* zbra_hip_sys::encode_frames then zbra_hip_sys::decode_batch: a Block's arrays in one call each way */
static void batched_block(void) {
    enum { NA = 7 };
    static const size_t ns[NA] = {50000, 0, 1, 4096, 70001, 333, 2048};
    int64_t *cols[NA];
    uint32_t *tags = 0;
    zbra_encode_desc ed[NA];
    size_t cap = 0;
    for (int i = 0; i < NA; i++) {
        cols[i] = (int64_t *)malloc((ns[i] ? ns[i] : 1) * sizeof(int64_t));
        column(i % 5, ns[i], 77 + i, cols[i]);
        memset(&ed[i], 0, sizeof(ed[i]));
        ed[i].values = ns[i] ? cols[i] : 0;
        ed[i].n = ns[i];
        ed[i].kind = ZBRA_ELEM_I64;
        cap += 8 + zbra_hip_encode_bound(ns[i]);
    }
    /* array 4 is an enum tag column: u32 in, u32 out */
    tags = (uint32_t *)malloc(ns[4] * sizeof(uint32_t));
    for (size_t k = 0; k < ns[4]; k++) { tags[k] = (uint32_t)(zo_sm(5, k) % 4 == 3 ? 7 : zo_sm(5, k) % 4); cols[4][k] = tags[k]; }
    ed[4].values = tags;
    ed[4].kind = ZBRA_ELEM_U32;
    uint8_t *frames = (uint8_t *)malloc(cap);
    uint64_t offsets[NA + 1];
    CHECK(zbra_hip_encode_frames(ed, NA, frames, cap, offsets, 0) == ZBRA_OK);
    const size_t total = (size_t)offsets[NA]; /* Vec::truncate */
    CHECK(offsets[0] == 0 && total <= cap);
    zbra_array_desc dd[NA];
    void *outs[NA];
    for (int i = 0; i < NA; i++) {
        const uint8_t *f = frames + offsets[i];
        uint32_t count, clen;
        memcpy(&count, f, 4);
        memcpy(&clen, f + 4, 4);
        CHECK(count == ns[i] && offsets[i + 1] - offsets[i] == 8u + clen);
        /* the frame's E equals the checker's */
        uint8_t *want = (uint8_t *)malloc(zo_compress_bound(ns[i]) + 1);
        const size_t want_len = zo_compress_int_array(cols[i], ns[i], want);
        CHECK(clen == want_len && (want_len == 0 || memcmp(f + 8, want, want_len) == 0));
        free(want);
        /* size the result with the header alone, as the wrapper does */
        size_t n = 0;
        CHECK(zbra_hip_decoded_count(f + 8, clen, count, &n) == ZBRA_OK && n == ns[i]);
        const int narrow = i == 4;
        outs[i] = calloc(n ? n : 1, narrow ? 4 : 8);
        memset(&dd[i], 0, sizeof(dd[i]));
        dd[i].data = f + 8; dd[i].len = clen; dd[i].count = n; dd[i].out = outs[i];
        dd[i].kind = narrow ? ZBRA_ELEM_U32 : ZBRA_ELEM_I64;
    }
    CHECK(zbra_hip_decode_batch(dd, NA, 0) == ZBRA_OK);
    for (int i = 0; i < NA; i++) {
        if (i == 4) CHECK(memcmp(outs[i], tags, ns[i] * 4) == 0);
        else CHECK(ns[i] == 0 || memcmp(outs[i], cols[i], ns[i] * 8) == 0);
        free(outs[i]); free(cols[i]);
    }
    free(tags); free(frames);
}

int main(void) {
    int devices = 0;
    if (zbra_hip_device_count(&devices) != ZBRA_OK || devices == 0) {
        printf("need a GPU: libzbra_hip has no CPU path\n");
        return 2;
    }
    CHECK(zbra_hip_init() == ZBRA_OK);
    single_arrays();
    batched_block();
    printf("%d checks, %d failures\n", checks, failures);
    return failures ? 1 : 0;
}
