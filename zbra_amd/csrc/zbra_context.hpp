// Per-device host-side state of libzbra_hip: workspace slots that a stream-ordered call borrows
// for its descriptor tables, and grow-only device/pinned buffers behind the host-pointer calls.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cstdlib>
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

// One of the two alternating lanes of the host write path (zbra_hip_encode_frames): a stream and
// the buffers of the round of arrays in flight on it: values in, frames out, offsets + result words
// and their pinned twin.
struct EncodeLane {
    hipStream_t stream = nullptr;
    Arena in, out, tmp, pin;
};

// Device-only memory that the kernels of ONE stream's calls pass through in stream order (the
// candidate lists of an encode): one buffer per stream serves every call on it, where a buffer per
// call in flight meant a fresh multi-GiB allocation whenever a caller ran ahead of the device.
struct StreamScratch {
    hipStream_t stream = nullptr;
    Arena mem;
    std::mutex mu;       // held while a call enqueues: kernels of two calls that share the buffer must not interleave
    uint64_t used = 0;   // tick of the last use (guarded by DeviceCtx::mu)
};

// What ONE host-pointer call (host memory at both ends) works with: streams, grow-only device and
// pinned buffers, the pinned rings.  A call holds the lane's mutex from start to end, so nothing of
// one call is still in flight on the lane when the next call takes it.  A device has a few lanes, so
// that calls from several threads overlap (one thread's upload with another's way back, or the
// fixed latencies of many small calls); a single-threaded caller only ever uses lane 0, and the
// other lanes hold no memory until a second thread arrives.
struct HostLane {
    std::mutex mu;
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr;            // the way back of a piecewise host decode
    hipEvent_t piece_done[4] = {nullptr, nullptr, nullptr, nullptr};
    Arena dev_in, dev_out, dev_tmp;
    Arena pin_res;              // pinned: status words of a whole-file decode
    HostPipe pipe;              // pageable host memory <-> device
    HostPipe pipe_up;           // uploads issued while `pipe` is busy with the way back (piecewise decode)
    FileLane file_lanes[2];
    EncodeLane enc_lanes[2];
    int ensure_stream() {
        if (!stream) ZBRA_HIP_TRY(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
        return ZBRA_OK;
    }
    int ensure_file_lanes() {
        for (FileLane &l : file_lanes)
            if (!l.stream) ZBRA_HIP_TRY(hipStreamCreateWithFlags(&l.stream, hipStreamNonBlocking));
        return ZBRA_OK;
    }
    int ensure_enc_lanes() {
        for (EncodeLane &l : enc_lanes)
            if (!l.stream) ZBRA_HIP_TRY(hipStreamCreateWithFlags(&l.stream, hipStreamNonBlocking));
        return ZBRA_OK;
    }
};

struct DeviceCtx {
    int device = -1;
    static constexpr int MAX_HOST_LANES = 4;
    HostLane host_lanes[MAX_HOST_LANES];
    // ZBRA_HIP_HOST_LANES=1..4: host-pointer calls that may run at the same time on one device
    // (default 2: each lane that is really used holds its own pinned rings and device buffers)
    const int n_host_lanes = [] {
        const char *e = std::getenv("ZBRA_HIP_HOST_LANES");
        const int v = e ? std::atoi(e) : 2;
        return v < 1 ? 1 : v > MAX_HOST_LANES ? MAX_HOST_LANES : v;
    }();
    std::atomic<unsigned> host_ticket{0};
    // The first lane nobody holds, lowest index first; with all of them taken the caller queues on
    // one of them in turn.  Returned with its mutex HELD.
    HostLane *lock_host_lane() {
        for (int i = 0; i < n_host_lanes; i++)
            if (host_lanes[i].mu.try_lock()) return &host_lanes[i];
        HostLane *l = &host_lanes[host_ticket.fetch_add(1) % (unsigned)n_host_lanes];
        l->mu.lock();
        return l;
    }
    std::mutex mu;              // guards slots
    std::vector<Slot *> slots;  // pointers stay valid while the vector grows
    void *noalloc_scratch = nullptr;  // this device's address of the library's static scratch array (set once, at creation)

    // A slot whose device buffer holds `bytes` and whose pinned host buffer holds `host_bytes`
    // (0 = as large as the device buffer: only the part the host fills needs a pinned twin).
    int acquire(size_t bytes, Slot **out, size_t host_bytes = 0) {
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
        }
        if (!s->host) { e = hipHostMalloc(&s->host, hcap, hipHostMallocDefault); if (e == hipSuccess) s->host_cap = hcap; }
        if (e == hipSuccess && !s->dev) { e = hipMalloc(&s->dev, cap); if (e == hipSuccess) s->cap = cap; }
        if (e == hipSuccess && !s->done) e = hipEventCreateWithFlags(&s->done, hipEventDisableTiming);
        if (e != hipSuccess) {
            if (fresh) {
                if (s->host) (void)hipHostFree(s->host);
                if (s->dev) (void)hipFree(s->dev);
                delete s;
            }  // a regrown slot stays in the list with whatever buffers it still has (caps say which)
            return hip_status(e);
        }
        s->busy = true;
        if (fresh) slots.push_back(s);
        *out = s;
        return ZBRA_OK;
    }

    // The scratch buffer of `stream` (created on first use), returned with its mutex HELD: unlock
    // it when the call's last kernel is enqueued.  At most MAX_STREAM_SCRATCH streams keep a buffer:
    // beyond that the least recently used one that nobody holds is handed over, once the device
    // has finished what was enqueued (streams come and go in a long-lived process).
    static constexpr size_t MAX_STREAM_SCRATCH = 16;
    std::vector<StreamScratch *> stream_scratch;
    uint64_t scratch_tick = 0;
    int lock_scratch_of(hipStream_t stream, StreamScratch **out) {
        for (;;) {
            StreamScratch *found = nullptr, *victim = nullptr;
            hipStream_t old_stream = nullptr;
            {
                std::lock_guard<std::mutex> g(mu);
                for (StreamScratch *ss : stream_scratch)
                    if (ss->stream == stream) { found = ss; break; }
                if (!found && stream_scratch.size() >= MAX_STREAM_SCRATCH) {
                    // the least recently used buffer that nobody is enqueueing with: idle entries are
                    // tried in LRU order before the list may grow past its cap
                    std::vector<StreamScratch *> by_age(stream_scratch);
                    std::sort(by_age.begin(), by_age.end(), [](const StreamScratch *a, const StreamScratch *b) { return a->used < b->used; });
                    for (StreamScratch *ss : by_age)
                        if (ss->mu.try_lock()) { victim = ss; break; }
                    if (victim) {
                        // re-labelled under `mu` (so that no other thread finds it under its old
                        // stream) and handed over with its own mutex held; the wait for the old
                        // stream's work and the free happen below, without `mu`
                        old_stream = victim->stream;
                        victim->stream = stream;
                        victim->used = ++scratch_tick;
                    }
                }
                if (!found && !victim) {  // below the cap, or every buffer is in somebody's hands
                    found = new StreamScratch();
                    found->stream = stream;
                    stream_scratch.push_back(found);
                }
                if (found) found->used = ++scratch_tick;
            }
            if (victim) {
                // kernels enqueued on the old stream may still use the buffer: only that stream is
                // waited for (a stream the caller has destroyed in the meantime reports an error,
                // and then the device is waited for instead), and nobody else's acquire()/release() waits with us
                hipError_t e = hipStreamSynchronize(old_stream);
                if (e != hipSuccess) { (void)hipGetLastError(); e = hipDeviceSynchronize(); }
                if (e != hipSuccess) { victim->mu.unlock(); return hip_status(e); }
                if (victim->mem.ptr) (void)hipFree(victim->mem.ptr);
                victim->mem = Arena();
                *out = victim;
                return ZBRA_OK;
            }
            // not under `mu`: whoever holds the buffer's mutex takes `mu` for its workspace slot
            found->mu.lock();
            if (found->stream == stream) { *out = found; return ZBRA_OK; }
            found->mu.unlock();  // handed over to another stream in between: look again
        }
    }
    // Call with ss->mu held.  Growing waits for the stream: its earlier calls read the old buffer.
    static int grow_scratch(StreamScratch *ss, size_t bytes) {
        if (ss->mem.cap >= bytes) return ZBRA_OK;
        ZBRA_HIP_TRY(hipStreamSynchronize(ss->stream));
        return grow_dev(ss->mem, bytes);
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

// A host lane of `ctx` for the length of a scope.
struct HostLaneLock {
    HostLane *lane;
    explicit HostLaneLock(DeviceCtx *ctx) : lane(ctx->lock_host_lane()) {}
    ~HostLaneLock() { lane->mu.unlock(); }
    HostLaneLock(const HostLaneLock &) = delete;
    HostLaneLock &operator=(const HostLaneLock &) = delete;
};

}  // namespace zbra
