// Per-device host-side state of libzbra_hip: workspace slots that a stream-ordered call borrows
// for its descriptor tables, and grow-only device/pinned buffers behind the host-pointer calls.
#pragma once
#include <hip/hip_runtime.h>

#include <mutex>
#include <vector>

#include "zbra_hip.h"

namespace zbra {

inline int hip_status(hipError_t e) { return e == hipSuccess ? ZBRA_OK : ZBRA_ERR_HIP_BASE + (int)e; }

#define ZBRA_HIP_TRY(expr)                                    \
    do {                                                      \
        hipError_t _e = (expr);                               \
        if (_e != hipSuccess) return ZBRA_ERR_HIP_BASE + (int)_e; \
    } while (0)

#define ZBRA_TRY(expr)                 \
    do {                               \
        int _s = (expr);               \
        if (_s != ZBRA_OK) return _s;  \
    } while (0)

}  // namespace zbra

#include "zbra_hostpipe.hpp"

namespace zbra {

// A pair of buffers (device + its pinned host twin for the part the host fills) lent to ONE enqueued call.  The host
// side may be rewritten only after everything the previous borrower enqueued has run, which the
// slot's event says.
struct Slot {
    void *host = nullptr;
    void *dev = nullptr;
    size_t cap = 0;       // of dev
    size_t host_cap = 0;  // of host
    void *scratch = nullptr;  // device-only memory of the borrower (candidate lists of an encode)
    size_t scratch_cap = 0;
    hipEvent_t done = nullptr;
    bool busy = false;  // between acquire() and release()
};

// Grow-only buffer.
struct Arena {
    void *ptr = nullptr;
    size_t cap = 0;
};

// One of the two alternating lanes of the whole-file decode: a stream and the device buffers of
// the group of arrays in flight on it.
struct FileLane {
    hipStream_t stream = nullptr;
    Arena img, res;
};

struct DeviceCtx {
    int device = -1;
    FileLane file_lanes[2];
    int ensure_file_lanes() {
        for (FileLane &l : file_lanes)
            if (!l.stream) ZBRA_HIP_TRY(hipStreamCreateWithFlags(&l.stream, hipStreamNonBlocking));
        return ZBRA_OK;
    }
    std::mutex mu;              // guards slots
    std::vector<Slot *> slots;  // pointers stay valid while the vector grows
    std::mutex host_mu;         // serialises host-pointer calls: they share the arenas and the stream
    void *noalloc_scratch = nullptr;  // this device's address of the library's static scratch array (set once, at creation)
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr;            // the way back of a piecewise host decode
    hipEvent_t piece_done[4] = {nullptr, nullptr, nullptr, nullptr};
    Arena dev_in, dev_out, dev_tmp;
    Arena pin_res;              // pinned: status words of a whole-file decode
    HostPipe pipe;              // pageable host memory <-> device, under host_mu
    HostPipe pipe_up;           // uploads issued while `pipe` is busy with the way back (piecewise decode)

    // A slot whose device buffer holds `bytes` and whose pinned host buffer holds `host_bytes`
    // (0 = as large as the device buffer: only the part the host fills needs a pinned twin);
    // scratch_bytes of further device-only memory come with it (grown while the slot is idle).
    int acquire(size_t bytes, Slot **out, size_t scratch_bytes = 0, size_t host_bytes = 0) {
        if (host_bytes == 0 || host_bytes > bytes) host_bytes = bytes;
        std::lock_guard<std::mutex> g(mu);
        Slot *small = nullptr;  // an idle slot that is too small: regrown rather than left behind
        for (Slot *s : slots) {
            if (s->busy) continue;
            if (hipEventQuery(s->done) != hipSuccess) {
                (void)hipGetLastError();  // hipErrorNotReady is not a failure
                continue;
            }
            if (s->cap < bytes || s->host_cap < host_bytes) { small = s; continue; }
            if (s->scratch_cap < scratch_bytes) {
                if (s->scratch) ZBRA_HIP_TRY(hipFree(s->scratch));
                s->scratch = nullptr; s->scratch_cap = 0;
                ZBRA_HIP_TRY(hipMalloc(&s->scratch, scratch_bytes));
                s->scratch_cap = scratch_bytes;
            }
            s->busy = true;
            *out = s;
            return ZBRA_OK;
        }
        const size_t cap = bytes < 4096 ? 4096 : bytes, hcap = host_bytes < 4096 ? 4096 : host_bytes;
        const bool fresh = small == nullptr;
        Slot *s = fresh ? new Slot() : small;
        hipError_t e = hipSuccess;
        if (!fresh) {  // everything enqueued with it has run (its event says so)
            if (s->cap < cap) { (void)hipFree(s->dev); s->dev = nullptr; s->cap = 0; }
            if (s->host_cap < hcap) { (void)hipHostFree(s->host); s->host = nullptr; s->host_cap = 0; }
            if (s->scratch_cap < scratch_bytes && s->scratch) { (void)hipFree(s->scratch); s->scratch = nullptr; s->scratch_cap = 0; }
        }
        if (!s->host) { e = hipHostMalloc(&s->host, hcap, hipHostMallocDefault); if (e == hipSuccess) s->host_cap = hcap; }
        if (e == hipSuccess && !s->dev) { e = hipMalloc(&s->dev, cap); if (e == hipSuccess) s->cap = cap; }
        if (e == hipSuccess && scratch_bytes && !s->scratch) {
            e = hipMalloc(&s->scratch, scratch_bytes);
            if (e == hipSuccess) s->scratch_cap = scratch_bytes;
        }
        if (e == hipSuccess && !s->done) e = hipEventCreateWithFlags(&s->done, hipEventDisableTiming);
        if (e != hipSuccess) {
            if (fresh) {
                if (s->host) (void)hipHostFree(s->host);
                if (s->dev) (void)hipFree(s->dev);
                if (s->scratch) (void)hipFree(s->scratch);
                delete s;
            }  // a regrown slot stays in the list with whatever buffers it still has (caps say which)
            return hip_status(e);
        }
        s->busy = true;
        if (fresh) slots.push_back(s);
        *out = s;
        return ZBRA_OK;
    }

    // Call after the last use of the slot has been enqueued on `stream`.
    int release(Slot *s, hipStream_t stream) {
        hipError_t e = hipEventRecord(s->done, stream);
        std::lock_guard<std::mutex> g(mu);
        s->busy = false;
        return hip_status(e);
    }

    static int grow_dev(Arena &a, size_t bytes) {
        if (a.cap >= bytes) return ZBRA_OK;
        if (a.ptr) ZBRA_HIP_TRY(hipFree(a.ptr));
        a.ptr = nullptr; a.cap = 0;
        size_t cap = bytes + bytes / 8 + 256;
        ZBRA_HIP_TRY(hipMalloc(&a.ptr, cap));
        a.cap = cap;
        return ZBRA_OK;
    }
    static int grow_pinned(Arena &a, size_t bytes) {
        if (a.cap >= bytes) return ZBRA_OK;
        if (a.ptr) ZBRA_HIP_TRY(hipHostFree(a.ptr));
        a.ptr = nullptr; a.cap = 0;
        ZBRA_HIP_TRY(hipHostMalloc(&a.ptr, bytes, hipHostMallocDefault));
        a.cap = bytes;
        return ZBRA_OK;
    }
};

// Context of the calling thread's current device (created on first use).
int current_ctx(DeviceCtx **out);

}  // namespace zbra
