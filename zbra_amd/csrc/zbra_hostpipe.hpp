// Copies between PAGEABLE host memory (what a caller's Vec / numpy array is) and the device at
// close to the PCIe rate.  A plain hipMemcpy on pageable memory pins or stages the pages on one
// thread and reaches about a fifth of the link; here the transfer is cut into chunks that travel
// through a ring of pinned buffers (DMA at link rate), and a few worker threads do the
// pinned <-> pageable memcpy of different chunks at the same time, which also spreads the page
// faults of a freshly allocated destination over several cores.
// Included by zbra_context.hpp (after its status macros), not on its own.
#pragma once
#include <hip/hip_runtime.h>
#include <sched.h>
#include <sys/mman.h>

#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <thread>
#include <vector>

namespace zbra {

// A stretch of a device range and where it is (or goes) in host memory.
struct PipeSegment {
    size_t dev_off;
    void *host;
    size_t len;
};

// Process-wide memcpy workers.  Never destroyed: the threads sleep on the condition variable and
// die with the process, so no static destructor can race them.
class CopyWorkers {
   public:
    // One piece of a transfer: the part [lo, hi) of a device range, whose image in the ring
    // starts at `img` for offset `img_off`, against the host segments segs[s0 ..] that overlap it
    // (a large array is one long segment, a Block of small arrays thousands of short ones: either
    // way a piece is about PIECE bytes of memcpy, never one queue entry per tiny array).
    struct Task {
        const PipeSegment *segs;
        size_t s0, nsegs;
        size_t lo, hi;
        char *img;
        size_t img_off;
        bool to_host;
        std::atomic<int> *pending;  // decremented when the piece has been copied
    };

    static CopyWorkers &instance() {
        static CopyWorkers *w = new CopyWorkers();
        return *w;
    }

    void submit(const Task &t) {
        {
            std::lock_guard<std::mutex> g(mu_);
            queue_.push_back(t);
        }
        cv_.notify_one();
    }

    // Blocks until `pending` is zero; the caller copies queued pieces itself meanwhile, so the
    // transfer makes progress even with no worker thread.
    void wait_zero(std::atomic<int> &pending) {
        while (pending.load(std::memory_order_acquire) != 0) {
            Task t;
            bool have = false;
            {
                std::lock_guard<std::mutex> g(mu_);
                if (!queue_.empty()) { t = queue_.front(); queue_.pop_front(); have = true; }
            }
            if (have) run(t);
            else sched_yield();
        }
    }

    int threads() const { return nthreads_.load(); }

    // Raises the number of worker threads to `want` (bounded by the CPUs the process may run on and by
    // 64); never lowers it.  A reader that drives several devices at once needs as many memcpy
    // workers per device as a reader of one device has, or the host side becomes the bound.
    void ensure_threads(int want) {
        if (std::getenv("ZBRA_HIP_HOST_THREADS")) return;  // the caller fixed the number
        cpu_set_t set;
        int cpus = 1;
        if (sched_getaffinity(0, sizeof(set), &set) == 0) cpus = CPU_COUNT(&set);
        if (want > cpus - 1) want = cpus - 1;
        if (want > 64) want = 64;
        std::lock_guard<std::mutex> g(grow_mu_);
        while (nthreads_.load() < want) {
            std::thread([this] { loop(); }).detach();
            nthreads_.fetch_add(1);
        }
    }

   private:
    CopyWorkers() {
        int n = 0;
        if (const char *e = std::getenv("ZBRA_HIP_HOST_THREADS")) n = std::atoi(e);
        else {
            cpu_set_t set;
            int cpus = 1;
            if (sched_getaffinity(0, sizeof(set), &set) == 0) cpus = CPU_COUNT(&set);
            n = cpus / 2;
            if (n > 8) n = 8;
        }
        if (n < 0) n = 0;
        if (n > 64) n = 64;
        nthreads_.store(n);
        for (int i = 0; i < n; i++) std::thread([this] { loop(); }).detach();
    }

    static void run(const Task &t) {
        for (size_t j = t.s0; j < t.nsegs && t.segs[j].dev_off < t.hi; j++) {
            const PipeSegment &g = t.segs[j];
            const size_t lo = g.dev_off > t.lo ? g.dev_off : t.lo;
            const size_t end = g.dev_off + g.len, hi = end < t.hi ? end : t.hi;
            if (hi <= lo) continue;
            char *h = (char *)g.host + (lo - g.dev_off), *r = t.img + (lo - t.img_off);
            if (t.to_host) std::memcpy(h, r, hi - lo);
            else std::memcpy(r, h, hi - lo);
        }
        t.pending->fetch_sub(1, std::memory_order_release);
    }

    void loop() {
        for (;;) {
            Task t;
            {
                std::unique_lock<std::mutex> g(mu_);
                cv_.wait(g, [this] { return !queue_.empty(); });
                t = queue_.front();
                queue_.pop_front();
            }
            run(t);
        }
    }

    std::mutex mu_;
    std::condition_variable cv_;
    std::deque<Task> queue_;
    std::atomic<int> nthreads_{0};
    std::mutex grow_mu_;
};

// A pinned ring of one host lane of a device context: used by the call that holds the lane.
struct HostPipe {
    static size_t env_kb(const char *name, size_t dflt) {
        const char *e = std::getenv(name);
        const long v = e ? std::atol(e) : 0;
        return v > 0 ? (size_t)v << 10 : dflt;
    }
    // One DMA moves at most CHUNK bytes.  Measured on 2 GiB coming back: 8-MiB chunks 45.6 GB/s,
    // 16 MiB 46.6, 32 MiB 48.0, 64 MiB 52.1 (8 slots: 52.9, 4 slots: 49.9).  Shorter transfers
    // are cut into 4 chunks so that DMA and memcpy still overlap; the ring grows to SLOTS chunks
    // of the largest size used, i.e. to at most twice the largest transfer and at most 512 MiB.
    const size_t CHUNK = env_kb("ZBRA_HIP_HOST_CHUNK_KB", (size_t)64 << 20);
    const size_t PIECE = env_kb("ZBRA_HIP_HOST_PIECE_KB", (size_t)1 << 20);  // one worker memcpy
    static constexpr size_t MAX_SLOTS = 16;
    const size_t SLOTS = [] {
        const char *e = std::getenv("ZBRA_HIP_HOST_SLOTS");
        const long v = e ? std::atol(e) : 0;
        return v >= 2 && v <= (long)MAX_SLOTS ? (size_t)v : (size_t)8;
    }();
    static constexpr size_t MIN_BYTES = (size_t)1 << 20;  // below this a plain hipMemcpyAsync is as good
    static constexpr size_t DIRECT_SEGS = 4;              // ... if it is not cut into many pieces

    char *ring = nullptr;
    size_t slot_bytes = 0;               // size of one ring slot
    bool have_events = false;
    hipEvent_t ev[MAX_SLOTS] = {};
    bool ev_live[MAX_SLOTS] = {};            // ev[i] was recorded and may not have been reached yet
    std::atomic<int> pending[MAX_SLOTS];     // memcpy pieces of slot i still queued or running

    HostPipe() {
        for (auto &p : pending) p.store(0);
    }

    // ZBRA_HIP_HOST_PIPE=0 falls back to plain hipMemcpyAsync on the caller's memory (for A/B runs).
    static bool enabled() {
        static const bool on = [] {
            const char *e = std::getenv("ZBRA_HIP_HOST_PIPE");
            return !(e && e[0] == '0');
        }();
        return on;
    }

    // A destination that has never been touched (a fresh Vec) is faulted in page by page while
    // the workers write it, and 4-KiB faults then cost more than the transfer.  Asking for
    // transparent huge pages on the whole 2-MiB blocks inside it cuts the faults by 512x where
    // the system allows it (THP "madvise" or "always"); elsewhere the call does nothing.  It is
    // a paging hint only.  ZBRA_HIP_HOST_HUGEPAGES=0 leaves the caller's memory alone.
    static void advise_huge(void *host, size_t bytes) {
        static const bool on = [] {
            const char *e = std::getenv("ZBRA_HIP_HOST_HUGEPAGES");
            return !(e && e[0] == '0');
        }();
        constexpr uintptr_t H = (uintptr_t)2 << 20;
        if (!on || bytes < 4 * H) return;
        const uintptr_t lo = ((uintptr_t)host + H - 1) & ~(H - 1), hi = ((uintptr_t)host + bytes) & ~(H - 1);
        if (hi > lo) (void)madvise((void *)lo, hi - lo, MADV_HUGEPAGE);
    }

    // Ring of SLOTS slots of at least `chunk` bytes.
    int ensure(size_t chunk) {
        if (!have_events) {
            for (size_t i = 0; i < SLOTS; i++) ZBRA_HIP_TRY(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming));
            have_events = true;
        }
        if (slot_bytes >= chunk) return ZBRA_OK;
        for (size_t i = 0; i < SLOTS; i++) ZBRA_TRY(slot_idle(i));  // a DMA may still read the old ring
        if (ring) ZBRA_HIP_TRY(hipHostFree(ring));
        ring = nullptr; slot_bytes = 0;
        ZBRA_HIP_TRY(hipHostMalloc((void **)&ring, chunk * SLOTS, hipHostMallocDefault));
        slot_bytes = chunk;
        return ZBRA_OK;
    }

    size_t chunk_for(size_t bytes) const {
        constexpr size_t G = (size_t)256 << 10;
        size_t c = (bytes / 4 + G - 1) / G * G;
        if (c < MIN_BYTES) c = MIN_BYTES;
        return c < CHUNK ? c : CHUNK;
    }

    int slot_idle(size_t slot) {
        if (ev_live[slot]) {
            ZBRA_HIP_TRY(hipEventSynchronize(ev[slot]));
            ev_live[slot] = false;
        }
        return ZBRA_OK;
    }

    void drain() {  // workers hold pointers into the caller's memory and the ring until they are done
        for (size_t i = 0; i < SLOTS; i++) CopyWorkers::instance().wait_zero(pending[i]);
    }

    using Segment = PipeSegment;

    // host -> device: the device range [dev, dev + bytes) is filled from the segments (sorted by
    // dev_off, not overlapping; gaps are left as they are on the device side of the chunk, i.e.
    // they receive whatever the ring held).  Returns when the host memory has been read
    // completely; the device copies are ordered on `stream` like a hipMemcpyAsync.
    int to_device_segments(const Segment *segs, size_t nsegs, void *dev, size_t bytes, hipStream_t stream) {
        // a few short segments: one plain copy each; many short segments are gathered in the ring
        // and go up in one DMA (a copy call costs microseconds, thousands of them milliseconds)
        if ((bytes < MIN_BYTES && nsegs <= DIRECT_SEGS) || !enabled()) {
            for (size_t j = 0; j < nsegs; j++)
                if (segs[j].len)
                    ZBRA_HIP_TRY(hipMemcpyAsync((char *)dev + segs[j].dev_off, segs[j].host, segs[j].len, hipMemcpyHostToDevice, stream));
            return ZBRA_OK;
        }
        ZBRA_TRY(ensure(chunk_for(bytes)));
        const int st = stage_to_device(segs, nsegs, dev, bytes, stream);
        drain();
        return st;
    }

    int stage_to_device(const Segment *segs, size_t nsegs, void *dev, size_t bytes, hipStream_t stream) {
        CopyWorkers &w = CopyWorkers::instance();
        const size_t chunk = chunk_for(bytes), nchunks = (bytes + chunk - 1) / chunk;
        size_t staged = 0, sent = 0, si = 0;
        while (sent < nchunks) {
            while (staged < nchunks && staged - sent < SLOTS) {
                const size_t slot = staged % SLOTS, off = staged * chunk;
                ZBRA_TRY(slot_idle(slot));  // the DMA that last read this slot has finished
                deliver<false>(slot, ring + slot * slot_bytes, off, bytes - off < chunk ? bytes - off : chunk, segs, nsegs, si);
                staged++;
            }
            const size_t slot = sent % SLOTS, off = sent * chunk;
            w.wait_zero(pending[slot]);
            ZBRA_HIP_TRY(hipMemcpyAsync((char *)dev + off, ring + slot * slot_bytes, bytes - off < chunk ? bytes - off : chunk,
                                        hipMemcpyHostToDevice, stream));
            ZBRA_HIP_TRY(hipEventRecord(ev[slot], stream));
            ev_live[slot] = true;
            sent++;
        }
        return ZBRA_OK;
    }

    int to_device(void *dev, const void *host, size_t bytes, hipStream_t stream) {
        const Segment whole{0, const_cast<void *>(host), bytes};
        return to_device_segments(&whole, 1, dev, bytes, stream);
    }

    // Queues the memcpys between the ring image `img` of the part [off, off + len) of the device
    // range and the segments that part overlaps: TO_HOST true = image -> segments, false =
    // segments -> image.  `si` is the first segment that can still overlap (segments are sorted by
    // dev_off and do not overlap; gaps are padding).  One task per PIECE bytes of the range.
    template <bool TO_HOST>
    void deliver(size_t slot, char *img, size_t off, size_t len, const Segment *segs, size_t nsegs, size_t &si) {
        CopyWorkers &w = CopyWorkers::instance();
        while (si < nsegs && segs[si].dev_off + segs[si].len <= off) si++;
        if (si >= nsegs || segs[si].dev_off >= off + len) return;  // only padding in this part
        const int pieces = (int)((len + PIECE - 1) / PIECE);
        pending[slot].store(pieces, std::memory_order_relaxed);
        size_t s0 = si;
        for (size_t lo = off; lo < off + len; lo += PIECE) {
            const size_t hi = lo + PIECE < off + len ? lo + PIECE : off + len;
            while (s0 < nsegs && segs[s0].dev_off + segs[s0].len <= lo) s0++;
            w.submit({segs, s0, nsegs, lo, hi, img, off, TO_HOST, &pending[slot]});
        }
    }

    // device -> host, after everything enqueued on `stream` so far: the range [dev, dev + bytes)
    // is copied back and cut up into the segments.  Returns when the host memory holds the bytes.
    // `before(last_byte)` is called before the DMA of a chunk is enqueued, with the offset one past
    // the chunk's last byte: the caller uses it to enqueue (or make `stream` wait for) the work
    // that produces those bytes, so that producing and copying back overlap.
    template <typename Before>
    int to_host_segments(const Segment *segs, size_t nsegs, const void *dev, size_t bytes, hipStream_t stream, Before &&before) {
        if ((bytes < MIN_BYTES && nsegs <= DIRECT_SEGS) || !enabled()) {
            ZBRA_TRY(before(bytes));
            for (size_t j = 0; j < nsegs; j++)
                if (segs[j].len)
                    ZBRA_HIP_TRY(hipMemcpyAsync(segs[j].host, (const char *)dev + segs[j].dev_off, segs[j].len, hipMemcpyDeviceToHost, stream));
            ZBRA_HIP_TRY(hipStreamSynchronize(stream));
            return ZBRA_OK;
        }
        ZBRA_TRY(ensure(chunk_for(bytes)));
        for (size_t j = 0; j < nsegs; j++) advise_huge(segs[j].host, segs[j].len);
        CopyWorkers &w = CopyWorkers::instance();
        const size_t chunk = chunk_for(bytes), nchunks = (bytes + chunk - 1) / chunk;
        size_t issued = 0, landed = 0, si = 0;
        int st = ZBRA_OK;
        while (landed < nchunks && st == ZBRA_OK) {
            while (issued < nchunks && issued - landed < SLOTS) {
                const size_t slot = issued % SLOTS, off = issued * chunk, len = bytes - off < chunk ? bytes - off : chunk;
                w.wait_zero(pending[slot]);  // the memcpy out of this slot has finished
                if ((st = slot_idle(slot)) != ZBRA_OK) break;
                if ((st = before(off + len)) != ZBRA_OK) break;
                hipError_t e = hipMemcpyAsync(ring + slot * slot_bytes, (const char *)dev + off, len, hipMemcpyDeviceToHost, stream);
                if (e == hipSuccess) e = hipEventRecord(ev[slot], stream);
                if (e != hipSuccess) { st = hip_status(e); break; }
                ev_live[slot] = true;
                issued++;
            }
            if (st != ZBRA_OK) break;
            const size_t slot = landed % SLOTS, off = landed * chunk;
            if ((st = slot_idle(slot)) != ZBRA_OK) break;
            deliver<true>(slot, ring + slot * slot_bytes, off, bytes - off < chunk ? bytes - off : chunk, segs, nsegs, si);
            landed++;
        }
        drain();
        return st;
    }

    template <typename Before>
    int to_host(void *host, const void *dev, size_t bytes, hipStream_t stream, Before &&before) {
        const Segment whole{0, host, bytes};
        return to_host_segments(&whole, 1, dev, bytes, stream, before);
    }
    int to_host(void *host, const void *dev, size_t bytes, hipStream_t stream) {
        return to_host(host, dev, bytes, stream, [](size_t) { return (int)ZBRA_OK; });
    }
};

}  // namespace zbra
