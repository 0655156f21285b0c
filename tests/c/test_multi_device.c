/*
 * One process, every GPU: two threads per device of zbra_hip_device_count(), all running at once.
 * Thread A of device d: zbra_hip_set_device(d) -> zbra_hip_init -> encode on the device ->
 * decode on the device -> batched decode -> the no-allocation batched decode (which uses the
 * library's fixed scratch array of THAT device) -> compare.  Thread B of device d: the host-pointer
 * encode and decode on the same device at the same time.  Every result is compared with the CPU
 * restatement of the reference (the checker).  Files and Blocks are independent units
 * (zbra-core/src/binary.rs:124-127, 167-171), so an 8-GPU reader is exactly this: a thread per
 * device, no exchange.  On a one-GPU box it degenerates to one pair of threads.
 * C99 + the HIP host API (memory and streams only).
 * Exit status: 0 = all checks passed, 1 = a check failed, 2 = no GPU.
 */
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <hip/hip_runtime_api.h>

#include "zbra_hip.h"

/* the checker (oracle/zbra_oracle.c) */
size_t zo_compress_bound(size_t n);
size_t zo_compress_int_array(const int64_t *values, size_t n, uint8_t *out);
uint64_t zo_sm(uint64_t seed, uint64_t i);

static int failures = 0, checks = 0;
#define CHECK(cond)                                                                       \
    do {                                                                                  \
        __sync_fetch_and_add(&checks, 1);                                                 \
        if (!(cond)) {                                                                    \
            __sync_fetch_and_add(&failures, 1);                                           \
            printf("FAILED %s:%d (device %d): %s\n", __FILE__, __LINE__, a->device, #cond); \
        }                                                                                 \
    } while (0)

typedef struct {
    int device;
    int role; /* 0 = device-resident calls, 1 = host-pointer calls */
    pthread_barrier_t *start;
    char pci[32];
} arg_t;

static void column(int kind, size_t n, uint64_t seed, int64_t *v) {
    for (size_t i = 0; i < n; i++) {
        uint64_t r = zo_sm(seed, i);
        switch (kind) {
            case 0: v[i] = 1700000000000LL + (int64_t)(r % 4000); break;      /* width 12 */
            case 1: v[i] = (int64_t)r; break;                                  /* raw */
            case 2: v[i] = (int64_t)(i * 3600000ULL) + 4102444800000LL; break; /* hourly timestamps */
            default: v[i] = (int64_t)(r % 7) - 3; break;                       /* width 3 */
        }
    }
}

enum { NA = 4 };
static const size_t kCounts[NA] = {300001, 70001, 4097, 131072};

static void device_resident(arg_t *a) {
    hipStream_t stream;
    CHECK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking) == hipSuccess);
    int64_t *v[NA];
    uint8_t *want[NA];
    size_t want_len[NA], in_off[NA], e_off[NA], total_in = 0, total_e = 0;
    for (int i = 0; i < NA; i++) {
        const size_t n = kCounts[i];
        v[i] = (int64_t *)malloc(n * 8);
        column(i, n, 9000 + 16 * (uint64_t)a->device + i, v[i]); /* every device has its own data */
        want[i] = (uint8_t *)malloc(zo_compress_bound(n));
        want_len[i] = zo_compress_int_array(v[i], n, want[i]);
        in_off[i] = total_in; total_in += n * 8;
        e_off[i] = total_e; total_e += zbra_hip_encode_bound(n) + 3; /* odd offsets: E sits at any byte address */
    }
    char *d_in = 0, *d_e = 0, *d_out = 0, *d_out2 = 0;
    zbra_dev_result *d_res = 0;
    CHECK(hipMalloc((void **)&d_in, total_in) == hipSuccess);
    CHECK(hipMalloc((void **)&d_e, total_e + 64) == hipSuccess);
    CHECK(hipMalloc((void **)&d_out, total_in) == hipSuccess);
    CHECK(hipMalloc((void **)&d_out2, total_in) == hipSuccess);
    CHECK(hipMalloc((void **)&d_res, 3 * NA * sizeof(zbra_dev_result)) == hipSuccess);
    for (int i = 0; i < NA; i++) CHECK(hipMemcpy(d_in + in_off[i], v[i], kCounts[i] * 8, hipMemcpyHostToDevice) == hipSuccess);
    /* encode on the device, one batched call */
    zbra_encode_desc ed[NA];
    for (int i = 0; i < NA; i++) {
        memset(&ed[i], 0, sizeof(ed[i]));
        ed[i].values = d_in + in_off[i]; ed[i].n = kCounts[i]; ed[i].out = (uint8_t *)d_e + e_off[i];
        ed[i].cap = zbra_hip_encode_bound(kCounts[i]); ed[i].kind = ZBRA_ELEM_I64;
    }
    CHECK(zbra_hip_encode_batch_dev(ed, NA, d_res, stream) == ZBRA_OK);
    CHECK(hipStreamSynchronize(stream) == hipSuccess);
    zbra_dev_result res[3 * NA];
    CHECK(hipMemcpy(res, d_res, NA * sizeof(zbra_dev_result), hipMemcpyDeviceToHost) == hipSuccess);
    for (int i = 0; i < NA; i++) {
        CHECK(res[i].status == ZBRA_OK && res[i].out_count == want_len[i]);
        uint8_t *got = (uint8_t *)malloc(want_len[i]);
        CHECK(hipMemcpy(got, d_e + e_off[i], want_len[i], hipMemcpyDeviceToHost) == hipSuccess);
        CHECK(memcmp(got, want[i], want_len[i]) == 0);
        free(got);
    }
    /* decode: one array alone, the batch, and the batch through the no-allocation entry point */
    zbra_array_desc dd[NA], dn[NA];
    for (int i = 0; i < NA; i++) {
        memset(&dd[i], 0, sizeof(dd[i]));
        dd[i].data = (const uint8_t *)d_e + e_off[i]; dd[i].len = want_len[i]; dd[i].count = kCounts[i];
        dd[i].out = d_out + in_off[i]; dd[i].kind = ZBRA_ELEM_I64;
        dn[i] = dd[i];
        dn[i].out = d_out2 + in_off[i];
    }
    CHECK(hipMemsetAsync(d_out, 0xEE, total_in, stream) == hipSuccess);
    CHECK(hipMemsetAsync(d_out2, 0xEE, total_in, stream) == hipSuccess);
    CHECK(zbra_hip_decode_batch_dev(dd, NA, d_res + NA, stream) == ZBRA_OK);
    CHECK(zbra_hip_decode_batch_dev_noalloc(dn, NA, d_res + 2 * NA, stream) == ZBRA_OK);
    CHECK(hipStreamSynchronize(stream) == hipSuccess);
    CHECK(hipMemcpy(res, d_res, 3 * NA * sizeof(zbra_dev_result), hipMemcpyDeviceToHost) == hipSuccess);
    for (int i = 0; i < NA; i++) {
        int64_t *back = (int64_t *)malloc(kCounts[i] * 8);
        CHECK(res[NA + i].status == ZBRA_OK && res[NA + i].out_count == kCounts[i]);
        CHECK(memcmp(&res[NA + i], &res[2 * NA + i], sizeof(zbra_dev_result)) == 0);
        CHECK(hipMemcpy(back, d_out + in_off[i], kCounts[i] * 8, hipMemcpyDeviceToHost) == hipSuccess);
        CHECK(memcmp(back, v[i], kCounts[i] * 8) == 0);
        CHECK(hipMemcpy(back, d_out2 + in_off[i], kCounts[i] * 8, hipMemcpyDeviceToHost) == hipSuccess);
        CHECK(memcmp(back, v[i], kCounts[i] * 8) == 0);
        /* and alone, into the first buffer again */
        CHECK(hipMemsetAsync(d_out + in_off[i], 0x11, kCounts[i] * 8, stream) == hipSuccess);
        CHECK(zbra_hip_decode_dev((const uint8_t *)d_e + e_off[i], want_len[i], kCounts[i], d_out + in_off[i], ZBRA_ELEM_I64, 0, stream) == ZBRA_OK);
        CHECK(hipStreamSynchronize(stream) == hipSuccess);
        CHECK(hipMemcpy(back, d_out + in_off[i], kCounts[i] * 8, hipMemcpyDeviceToHost) == hipSuccess);
        CHECK(memcmp(back, v[i], kCounts[i] * 8) == 0);
        free(back);
    }
    /* the memory this thread got is on ITS device */
    struct hipPointerAttribute_t attr;
    CHECK(hipPointerGetAttributes(&attr, d_out) == hipSuccess && attr.device == a->device);
    hipFree(d_in); hipFree(d_e); hipFree(d_out); hipFree(d_out2); hipFree(d_res);
    hipStreamDestroy(stream);
    for (int i = 0; i < NA; i++) { free(v[i]); free(want[i]); }
}

static void host_pointers(arg_t *a) {
    for (int round = 0; round < 3; round++) {
        const size_t n = 200003 + 1000 * (size_t)round;
        int64_t *v = (int64_t *)malloc(n * 8), *back = (int64_t *)malloc(n * 8);
        column(round, n, 500 + 16 * (uint64_t)a->device + round, v);
        uint8_t *want = (uint8_t *)malloc(zo_compress_bound(n)), *got = (uint8_t *)malloc(zbra_hip_encode_bound(n));
        const size_t want_len = zo_compress_int_array(v, n, want);
        size_t len = 0, count = 0;
        CHECK(zbra_hip_encode_i64(v, n, got, zbra_hip_encode_bound(n), &len) == ZBRA_OK);
        CHECK(len == want_len && memcmp(got, want, len) == 0);
        CHECK(zbra_hip_decode_i64(want, want_len, n, back, &count) == ZBRA_OK);
        CHECK(count == n && memcmp(back, v, n * 8) == 0);
        free(v); free(back); free(want); free(got);
    }
}

static void *worker(void *p) {
    arg_t *a = (arg_t *)p;
    CHECK(zbra_hip_set_device(a->device) == ZBRA_OK);
    CHECK(zbra_hip_init() == ZBRA_OK);
    int cur = -1;
    CHECK(hipGetDevice(&cur) == hipSuccess && cur == a->device);
    if (hipDeviceGetPCIBusId(a->pci, (int)sizeof(a->pci), a->device) != hipSuccess) strcpy(a->pci, "?");
    pthread_barrier_wait(a->start); /* every device starts its work at the same moment */
    if (a->role == 0) device_resident(a);
    else host_pointers(a);
    return 0;
}

int main(void) {
    int devices = 0;
    if (zbra_hip_device_count(&devices) != ZBRA_OK || devices == 0) {
        printf("need a GPU: libzbra_hip has no CPU path\n");
        return 2;
    }
    if (devices > 16) devices = 16;
    const int nthreads = 2 * devices;
    pthread_t th[32];
    arg_t args[32];
    pthread_barrier_t start;
    pthread_barrier_init(&start, 0, (unsigned)nthreads);
    for (int t = 0; t < nthreads; t++) {
        args[t].device = t / 2; args[t].role = t % 2; args[t].start = &start; args[t].pci[0] = 0;
        pthread_create(&th[t], 0, worker, &args[t]);
    }
    for (int t = 0; t < nthreads; t++) pthread_join(th[t], 0);
    for (int d = 0; d < devices; d++) printf("device %d at %s\n", d, args[2 * d].pci);
    printf("%d devices, %d checks, %d failures\n", devices, checks, failures);
    return failures ? 1 : 0;
}
