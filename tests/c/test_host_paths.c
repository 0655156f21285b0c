/*
 * The host-memory paths that run extra threads inside the library, driven from plain C so that the
 * same program can be built with the host side of the library under ASan + UBSan (make san):
 *   - zbra_hip_encode_frames on enough values (about 190 MB) for several rounds on its two lanes,
 *     with a generous, an exact and a failing capacity: frames against the checker's, byte for byte;
 *   - zbra_hip_file_decode_multi: the frames just written serve as the "image" (an array of a file
 *     is an offset, a length and a count), two jobs per device of the box plus one for a device that
 *     does not exist.
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
uint64_t zo_sm(uint64_t seed, uint64_t i);

static int failures = 0, checks = 0;
#define CHECK(cond)                                                              \
    do {                                                                         \
        checks++;                                                                \
        if (!(cond)) { failures++; printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); } \
    } while (0)

enum { NA = 7 };
static const size_t kCounts[NA] = {4194304, 3000001, 0, 5000000, 4200000, 1, 3700003};

static void column(int kind, size_t n, uint64_t seed, int64_t *v) {
    for (size_t i = 0; i < n; i++) {
        uint64_t r = zo_sm(seed, i);
        switch (kind % 5) {
            case 0: v[i] = 1700000000000LL + (int64_t)(r % 4000); break;           /* width 12 */
            case 1: v[i] = (int64_t)(i * 1000ULL) + 1640995200000LL; break;         /* sorted timestamps: wide bracket */
            case 2: v[i] = (int64_t)(r % 2) * 5000 - 17; break;                     /* two values */
            case 3: v[i] = (int64_t)(r >> 20); break;                               /* raw */
            default: v[i] = -42; break;                                             /* constant */
        }
    }
}

int main(void) {
    int devices = 0;
    if (zbra_hip_device_count(&devices) != ZBRA_OK || devices == 0) {
        printf("need a GPU: libzbra_hip has no CPU path\n");
        return 2;
    }
    CHECK(zbra_hip_init() == ZBRA_OK);
    int64_t *cols[NA];
    uint8_t *want;
    uint64_t want_off[NA + 1];
    zbra_encode_desc ed[NA];
    size_t bound = 0, total = 0;
    for (int i = 0; i < NA; i++) {
        cols[i] = (int64_t *)malloc((kCounts[i] ? kCounts[i] : 1) * 8);
        column(i, kCounts[i], 40 + i, cols[i]);
        memset(&ed[i], 0, sizeof(ed[i]));
        ed[i].values = kCounts[i] ? cols[i] : 0;
        ed[i].n = kCounts[i];
        ed[i].kind = ZBRA_ELEM_I64;
        bound += 8 + zbra_hip_encode_bound(kCounts[i]);
    }
    want = (uint8_t *)malloc(bound);
    for (int i = 0; i < NA; i++) {  /* write_int_array_compressed, binary.rs:639-645 */
        want_off[i] = total;
        const size_t clen = zo_compress_int_array(cols[i], kCounts[i], want + total + 8);
        const uint32_t count = (uint32_t)kCounts[i], cl = (uint32_t)clen;
        memcpy(want + total, &count, 4);
        memcpy(want + total + 4, &cl, 4);
        total += 8 + clen;
    }
    want_off[NA] = total;

    uint8_t *out = (uint8_t *)malloc(bound + 64);
    uint64_t offsets[NA + 1];
    int statuses[NA];
    const size_t caps[3] = {bound, total, total + 999};
    for (int c = 0; c < 3; c++) {
        memset(out, 0xAB, bound + 64);
        CHECK(zbra_hip_encode_frames(ed, NA, out, caps[c], offsets, statuses) == ZBRA_OK);
        CHECK(memcmp(offsets, want_off, sizeof(offsets)) == 0);
        CHECK(memcmp(out, want, total) == 0);
        CHECK(out[total] == 0xAB && out[bound + 63] == 0xAB);
        for (int i = 0; i < NA; i++) CHECK(statuses[i] == ZBRA_OK);
    }
    /* frame 4 misses one byte: it and every later array fail, nothing of them is written */
    {
        const int k = 4;
        memset(out, 0xAB, bound + 64);
        CHECK(zbra_hip_encode_frames(ed, NA, out, (size_t)want_off[k + 1] - 1, offsets, statuses) == ZBRA_ERR_CAPACITY);
        for (int i = 0; i < NA; i++) CHECK(statuses[i] == (i < k ? ZBRA_OK : ZBRA_ERR_CAPACITY));
        for (int i = 0; i <= NA; i++) CHECK(offsets[i] == (i <= k ? want_off[i] : want_off[k]));
        CHECK(memcmp(out, want, (size_t)want_off[k]) == 0);
        size_t dirty = 0;
        for (size_t b = (size_t)want_off[k]; b < bound + 64; b++) dirty += out[b] != 0xAB;
        CHECK(dirty == 0);
    }

    /* the frames as an image: every device decodes all of it twice (two jobs each), all at once */
    zbra_file_array arrays[NA];
    for (int i = 0; i < NA; i++) {
        memset(&arrays[i], 0, sizeof(arrays[i]));
        uint32_t clen;
        memcpy(&clen, want + want_off[i] + 4, 4);
        arrays[i].offset = want_off[i] + 8;
        arrays[i].len = clen;
        arrays[i].count = kCounts[i];
        arrays[i].kind = ZBRA_ELEM_I64;
    }
    const int njobs = 2 * devices + 1;
    zbra_file_job *jobs = (zbra_file_job *)calloc((size_t)njobs, sizeof(zbra_file_job));
    void ***outs = (void ***)calloc((size_t)njobs, sizeof(void **));
    for (int j = 0; j < njobs; j++) {
        outs[j] = (void **)calloc(NA, sizeof(void *));
        for (int i = 0; i < NA; i++) outs[j][i] = calloc(kCounts[i] ? kCounts[i] : 1, 8);
        jobs[j].device = j < 2 * devices ? j / 2 : devices + 5;
        jobs[j].status = -1;
        jobs[j].image = want;
        jobs[j].len = total;
        jobs[j].arrays = arrays;
        jobs[j].n = NA;
        jobs[j].outs = outs[j];
        jobs[j].statuses = 0;
    }
    CHECK(zbra_hip_file_decode_multi(jobs, (size_t)njobs) == ZBRA_ERR_INVALID_ARG); /* the last job's device does not exist */
    CHECK(jobs[njobs - 1].status == ZBRA_ERR_INVALID_ARG);
    for (int j = 0; j + 1 < njobs; j++) {
        CHECK(jobs[j].status == ZBRA_OK);
        for (int i = 0; i < NA; i++) CHECK(kCounts[i] == 0 || memcmp(outs[j][i], cols[i], kCounts[i] * 8) == 0);
    }
    for (int j = 0; j < njobs; j++) {
        for (int i = 0; i < NA; i++) free(outs[j][i]);
        free(outs[j]);
    }
    free(outs); free(jobs); free(out); free(want);
    for (int i = 0; i < NA; i++) free(cols[i]);
    printf("%d devices, %d checks, %d failures\n", devices, checks, failures);
    return failures ? 1 : 0;
}
