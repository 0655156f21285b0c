// C ABI of libzbra_hip.so (declared in include/zbra_hip.h) over the kernels of
// zbra_decode.hpp / zbra_encode.hpp / zbra_stages.hpp.  gfx950 only; there is no CPU path.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstring>
#include <map>
#include <mutex>
#include <thread>
#include <vector>

#include "zbra_context.hpp"
#include "zbra_decode.hpp"
#include "zbra_encode.hpp"
#include "zbra_stages.hpp"
#include "zbra_file.hpp"
#include "zbra_offsets.hpp"
#include "zbra_enum.hpp"
#include "zbra_hip_tuning.h"

namespace zbra {

// ---- scratch of the entry point that allocates nothing ---------------------------------------
//
// zbra_hip_decode_batch_dev_noalloc keeps what zbra_hip_decode_batch_dev borrows from a workspace
// slot (the decode plans, the tiles-per-array scan and the tile -> array table) in this array.
// Every GPU the library runs on holds its own copy, at its own address, until the process ends.
// Host code never uses the name `scratch` as an address: it is a host-side stand-in for the symbol.
// The per-device address comes from hipGetSymbolAddress once per device (current_ctx below) and
// is what the kernels get as an argument.
struct NoAllocScratch {
    DecodePlan plans[ZBRA_HIP_NOALLOC_MAX_ARRAYS];
    uint64_t tile_start[ZBRA_HIP_NOALLOC_MAX_ARRAYS + 8];  // nd + 1 used
    uint32_t tile_map[ZBRA_HIP_NOALLOC_MAX_TILES];
};
static __device__ __align__(256) unsigned char scratch[sizeof(NoAllocScratch)];

static std::mutex g_ctx_mu;
static std::map<int, DeviceCtx *> g_ctx;

int current_ctx(DeviceCtx **out) {
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return ZBRA_ERR_NO_DEVICE;
    }
    std::lock_guard<std::mutex> g(g_ctx_mu);
    auto it = g_ctx.find(dev);
    if (it != g_ctx.end()) { *out = it->second; return ZBRA_OK; }
    DeviceCtx *c = new DeviceCtx();
    c->device = dev;
    e = hipStreamCreateWithFlags(&c->host_lanes[0].stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete c; return hip_status(e); }
    // this device's copy of `scratch` (device `dev` is current: hipGetDevice just said so)
    void *sp = nullptr;
    e = hipGetSymbolAddress(&sp, HIP_SYMBOL(scratch));
    if (e != hipSuccess || !sp) {
        (void)hipStreamDestroy(c->host_lanes[0].stream);
        delete c;
        return e != hipSuccess ? hip_status(e) : ZBRA_ERR_NO_DEVICE;
    }
    c->noalloc_scratch = sp;
    g_ctx[dev] = c;
    *out = c;
    return ZBRA_OK;
}

static inline uint64_t tiles_of(uint64_t count) { return (count + TILE - 1) / TILE; }  // encode tiles
static inline unsigned grid_for(uint64_t ntiles, unsigned cap) { return (unsigned)(ntiles < cap ? (ntiles ? ntiles : 1) : cap); }
static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// Decode kernel configuration: values per tile, threads per workgroup, nontemporal accesses,
// LDS-free reads, grid cap.  The defaults are the fastest measured on MI355X;
// ZBRA_HIP_DECODE_VARIANT="tile:threads:nt:direct" and ZBRA_HIP_DECODE_GRID override them
// (tuning aids).
struct DecodeConfig {
    int tile = 2048;
    int threads = 512;
    int nt = 3;  // bit 0: nontemporal loads, bit 1: nontemporal stores
    int direct = 0;
    int xcd = 2;  // tile order: 2 = by the submission's bits per value (decode_tile_order); 0, 1 or a group size force one
    unsigned grid_cap = 0x7FFFFFFFu;  // one workgroup per tile
};

struct Tuning;
static const Tuning &tuning();

// Work-items one dispatch may have: the grid of an AQL packet is 32 bits wide in work-items, and
// a larger launch is not refused but silently runs the low 32 bits' worth of workgroups.  Launches
// with one workgroup per tile are cut (decode) or capped (pack) by this.
// zbra_tuning_set_launch_limit lowers it so that tests reach the cut with small inputs.
static uint64_t launch_limit();

template <int DT, int TH>
static void launch_decode_t(const DecodeConfig &c, hipStream_t stream, const DecodePlan *plans, const uint64_t *ts,
                            const uint32_t *tmap, uint32_t nd, uint64_t first_tile, uint64_t all_tiles, uint32_t order) {
    // c.xcd: 2 = the order the caller worked out (decode_tile_order); anything else forces that order (tuning)
    const uint32_t want = c.xcd == 2 ? order : (uint32_t)c.xcd;
    // tiles per launch: a multiple of 8 (whole rounds of the XCDs) within the dispatch limit
    uint64_t per = launch_limit() / TH & ~7ull;
    if (per == 0) per = 8;
    for (uint64_t done = 0; done < all_tiles; done += per) {
        const uint64_t base = first_tile + done;
        const uint64_t ntiles = all_tiles - done < per ? all_tiles - done : per;
        if constexpr (DT == 2048 && TH == 512) {
            if (c.direct == 2) {  // experimental shape: two tiles per workgroup, packed bytes by LDS-DMA
                const uint64_t npairs = (ntiles + 1) / 2;
                const uint32_t xp = npairs >= 64 ? want : 0u;
                if (c.nt & 1) hipLaunchKernelGGL((decode_tiles_dma_kernel<DT, TH, 3>), dim3((unsigned)npairs), dim3(TH), 0, stream, plans, ts, tmap, nd, base, ntiles, xp);
                else hipLaunchKernelGGL((decode_tiles_dma_kernel<DT, TH, 2>), dim3((unsigned)npairs), dim3(TH), 0, stream, plans, ts, tmap, nd, base, ntiles, xp);
                continue;
            }
        }
        const unsigned grid = grid_for(ntiles, c.grid_cap);
        const dim3 g(grid), b(TH);
        const uint32_t x = ((uint64_t)grid == ntiles && ntiles >= 64) ? want : 0u;
#define ZBRA_LAUNCH_DECODE(NTV, DIR) \
    hipLaunchKernelGGL((decode_tiles_kernel<DT, TH, NTV, DIR>), g, b, 0, stream, plans, ts, tmap, nd, base, ntiles, x)
        switch ((c.nt & 3) * 2 + (c.direct == 1 ? 1 : 0)) {
            case 0: ZBRA_LAUNCH_DECODE(0, false); break;
            case 1: ZBRA_LAUNCH_DECODE(0, true); break;
            case 2: ZBRA_LAUNCH_DECODE(1, false); break;
            case 3: ZBRA_LAUNCH_DECODE(1, true); break;
            case 4: ZBRA_LAUNCH_DECODE(2, false); break;
            case 5: ZBRA_LAUNCH_DECODE(2, true); break;
            case 6: ZBRA_LAUNCH_DECODE(3, false); break;
            default: ZBRA_LAUNCH_DECODE(3, true); break;
        }
#undef ZBRA_LAUNCH_DECODE
    }
}

typedef void (*DecodeLauncher)(const DecodeConfig &, hipStream_t, const DecodePlan *, const uint64_t *, const uint32_t *,
                               uint32_t, uint64_t, uint64_t, uint32_t);
struct DecodeShape { int tile, threads; DecodeLauncher fn; };
static const DecodeShape kDecodeShapes[] = {  // the default and the alternatives it was measured against
    {2048, 512, launch_decode_t<2048, 512>}, {2048, 256, launch_decode_t<2048, 256>}, {1024, 256, launch_decode_t<1024, 256>},
    {4096, 512, launch_decode_t<4096, 512>}, {4096, 1024, launch_decode_t<4096, 1024>}, {1024, 128, launch_decode_t<1024, 128>},
    {512, 64, launch_decode_t<512, 64>},
};

// Launch policy of the process: the defaults, what the environment overrides at the first call, and
// what the zbra_tuning_set_* hooks (include/zbra_hip_tuning.h) change afterwards.  A call reads the
// current snapshot through tuning(); a hook publishes a modified copy, so a call running on another
// thread sees either the old policy or the new one, never a mixture of half-written fields.
// Superseded snapshots are not freed (a call may still hold one; the hooks are test and tuning aids).
struct Tuning {
    DecodeConfig decode;
    DecodeLauncher decode_fn = launch_decode_t<2048, 512>;
    uint64_t launch_limit = 0xFFFFFFFFull;
    // the warm-up route of narrow submissions (see launch_decode_submission): packed bytes per warmed
    // chunk (0 = off), widest array that takes it, smallest submission that does
    uint64_t warm_bytes = (uint64_t)64 << 20;
    uint32_t warm_bits = 6;
    uint64_t warm_min_bytes = (uint64_t)8 << 20;
    int enum_reverse = -1;  // -1: by size; 0 / 1 force the order of the enum write pass
};
static bool apply_variant(Tuning &t, int tile, int th, int n, int d) {
    for (const DecodeShape &sh : kDecodeShapes)
        if (sh.tile == tile && sh.threads == th) {
            t.decode.tile = tile; t.decode.threads = th; t.decode.nt = n & 3; t.decode.direct = d;  // 0 LDS image, 1 LDS-free, 2 LDS-DMA pairs
            t.decode_fn = sh.fn;
            return true;
        }
    return false;
}
static std::atomic<const Tuning *> g_tuning{nullptr};
static std::mutex g_tuning_mu;  // writers: the first call's set-up and the hooks
static const Tuning *tuning_init_locked() {
    const Tuning *cur = g_tuning.load(std::memory_order_acquire);
    if (cur) return cur;
    Tuning *t = new Tuning();
    if (const char *e = getenv("ZBRA_HIP_DECODE_VARIANT")) {
        int tile = 0, th = 0, n = 0, d = 0;
        if (sscanf(e, "%d:%d:%d:%d", &tile, &th, &n, &d) == 4) apply_variant(*t, tile, th, n, d);
    }
    if (const char *e = getenv("ZBRA_HIP_DECODE_XCD")) t->decode.xcd = atoi(e);  // 0 plain, 1 chunked, 2 by bits per value, >= 4 groups of that many tiles
    if (const char *e = getenv("ZBRA_HIP_DECODE_GRID")) {
        const long v = atol(e);
        if (v > 0) t->decode.grid_cap = (unsigned)v;
    }
    if (const char *e = getenv("ZBRA_HIP_DECODE_WARM"))
        if (atoi(e) == 0) t->warm_bytes = 0;
    g_tuning.store(t, std::memory_order_release);
    return t;
}
static const Tuning &tuning() {
    const Tuning *t = g_tuning.load(std::memory_order_acquire);
    if (t) return *t;
    std::lock_guard<std::mutex> g(g_tuning_mu);
    return *tuning_init_locked();
}
// Publishes a copy of the current policy changed by `change` (which returns false to refuse).
template <typename F>
static bool retune(F &&change) {
    std::lock_guard<std::mutex> g(g_tuning_mu);
    Tuning *t = new Tuning(*tuning_init_locked());
    if (!change(*t)) { delete t; return false; }
    g_tuning.store(t, std::memory_order_release);
    return true;
}
static uint64_t launch_limit() { return tuning().launch_limit; }
static const DecodeConfig &decode_config() { return tuning().decode; }
// Tiles of one array: u32 outputs use tiles of twice the values (see decode_tiles_kernel).
static inline uint64_t decode_tiles_of(uint64_t count, uint32_t kind) {
    const uint64_t t = (uint64_t)decode_config().tile * (kind == ZBRA_ELEM_U32 ? 2 : 1);
    return (count + t - 1) / t;
}
// The tile order of a decode launch (the kernel's xcd_chunked argument): 0 = tile b for workgroup b,
// 1 = each XCD walks one contiguous eighth of the tiles, G >= 4 = inside every run of 8 G tiles
// each XCD takes G consecutive ones.  Which is fastest depends on how many packed bytes a tile
// has (interleaved A/B in one process, profiles/r02_ab_decode_tile_order.txt; 2^27 rows unless said):
//   widths 2-9     plain       0.2058 against 0.2168 ms chunked at width 2; u32 tags 0.2036 / 0.2128
//   widths 11-13   chunked     0.2122 against 0.2169 plain, 0.2265 grouped (2^28 rows, width 12:
//                              0.4168 / 0.4336 / 0.4436)
//   widths 16-31   groups of 16  width 25: 0.2365 against 0.2495 chunked, 0.2618 plain (2^28 rows:
//                              0.4661 / 0.4860 / 0.5190; 2^25: 0.0660 / 0.0698 / 0.0705)
//   raw (>= 32)    plain       0.3673 against 0.3805 chunked; groups win at 2^27 rows (0.3505) and
//                              lose at 2^28 (0.7895 against 0.7381), so they are not used
// Batches of hundreds of packed arrays are fastest chunked whatever their widths (1024 ragged
// columns: 1.94 ms against 1.98 in groups of 16 and 2.18 plain).  The host decides from the
// encoded bits per value, which it knows from the frames' lengths and counts without reading a
// byte on the device.
static inline uint32_t decode_tile_order(uint64_t encoded_bytes, uint64_t values, size_t nd) {
    const uint64_t bits = encoded_bytes * 8;
    if (bits >= values * 32) return 0;  // raw, or mostly raw
    if (nd > 1) return 1;
    return bits >= values * 16 ? 16u : bits >= values * 10 ? 1u : 0u;
}

static void launch_decode(hipStream_t stream, const DecodePlan *plans, const uint64_t *ts, const uint32_t *tmap, uint32_t nd,
                          uint64_t ntiles, uint32_t order) {
    const Tuning &t = tuning();
    t.decode_fn(t.decode, stream, plans, ts, tmap, nd, 0, ntiles, order);
}
// Narrow submissions (every array at most Tuning::warm_bits encoded bits per value, warm_min_bytes or more
// in all) are decoded in chunks of about warm_bytes packed bytes: warm_tiles_kernel reads a
// chunk's packed bytes into the memory-side cache, then the tile kernel decodes it with plain
// loads and without LDS (each lane reads the dwords that hold its own values; lanes overlap, which
// the caches now absorb).  Measured with the caches emptied before every launch (scripts/ab_decode_cold.py,
// profiles/r02_ab_decode_warm.txt): 2^27 rows of width 2: 0.214 -> 0.171 ms, 2^28 rows of width 4: 0.408 ->
// 0.354, of width 6: 0.406 -> 0.385; from width 7 up the warm-up costs what it saves, and so it does for
// u32 outputs from width 4 up (2^29 tags of width 4: 0.403 -> 0.446 ms), whose limit is half the width.
// ZBRA_HIP_DECODE_WARM=0 turns it off; zbra_tuning_set_decode_warm changes chunk and width limit.

struct WarmPlan {
    bool on = false;
    uint64_t tiles_per_chunk = 0;
    uint32_t tiles_per_wg = 1;
};
static bool warm_route_available() {
    const Tuning &t = tuning();
    return t.warm_bytes && t.decode.tile == 2048 && !t.decode.direct;
}
// Does this array belong on the warm-up route?  Bits per value as the frame's length says them: exact for
// a well-formed E (a wrong guess costs time, nothing else).  u32 outputs are half as wide: the same packed
// bytes buy half the output, so half the width pays.  Arrays too short to tell go either way (`dflt`).
static bool warm_narrow(const zbra_array_desc &d, bool dflt) {
    if (d.count < 8 || d.len <= 13) return dflt;
    return (d.len - 13) * 8 / d.count * (d.kind == ZBRA_ELEM_U32 ? 2 : 1) <= tuning().warm_bits;
}
// Chunking of `ntiles` tiles that hold `bytes` encoded bytes.
static WarmPlan warm_plan_of(uint64_t ntiles, uint64_t bytes) {
    WarmPlan wp;
    if (!ntiles) return wp;
    wp.on = true;
    const uint64_t tile_bytes = bytes / ntiles ? bytes / ntiles : 1;
    wp.tiles_per_chunk = (tuning().warm_bytes / tile_bytes) & ~7ull;
    if (wp.tiles_per_chunk < 64) wp.tiles_per_chunk = 64;
    const uint64_t per_wg = 65536 / tile_bytes;
    wp.tiles_per_wg = (uint32_t)(per_wg < 1 ? 1 : per_wg > 256 ? 256 : per_wg);
    return wp;
}
static WarmPlan warm_plan_for(const zbra_array_desc *descs, size_t nd, uint64_t ntiles, uint64_t all_bytes) {
    if (!warm_route_available() || !ntiles || all_bytes < tuning().warm_min_bytes) return WarmPlan();
    for (size_t i = 0; i < nd; i++)
        if (!warm_narrow(descs[i], true)) return WarmPlan();
    return warm_plan_of(ntiles, all_bytes);
}

// Tile kernel launches of one submission: all its tiles at once, or chunk by chunk behind the
// warm-up of each chunk.
static void launch_decode_submission(hipStream_t stream, const DecodePlan *plans, const uint64_t *ts, const uint32_t *tmap, uint32_t nd,
                                     uint64_t first, uint64_t ntiles, uint32_t order, const WarmPlan &wp) {
    const Tuning &t = tuning();
    DecodeConfig c = t.decode;
    if (!wp.on) {
        t.decode_fn(c, stream, plans, ts, tmap, nd, first, ntiles, order);
        return;
    }
    c.nt = 2;      // plain loads (they hit what the warm-up left), nontemporal stores
    c.direct = 1;
    for (uint64_t done = 0; done < ntiles; done += wp.tiles_per_chunk) {
        const uint64_t n = ntiles - done < wp.tiles_per_chunk ? ntiles - done : wp.tiles_per_chunk;
        const unsigned grid = (unsigned)((n + wp.tiles_per_wg - 1) / wp.tiles_per_wg);
        hipLaunchKernelGGL((warm_tiles_kernel<2048>), dim3(grid), dim3(256), 0, stream, plans, ts, tmap, nd, first + done, n, wp.tiles_per_wg,
                           t.warm_bits > 31 ? 31u : t.warm_bits);
        t.decode_fn(c, stream, plans, ts, tmap, nd, first + done, n, order);
    }
}

constexpr unsigned STREAM_GRID_CAP = 256 * 8;

// Arrays one device submission handles at once: its per-array launches (one workgroup of up to
// 1024 lanes per array) stay far inside the work-items of a dispatch, and its tables stay small.
constexpr size_t MAX_BATCH_ARRAYS = (size_t)1 << 20;

// ZBRA_HIP_ENCODE_BRACKET=0 turns the sampled bracket off: every array then takes the path a
// failed bracket takes (min/max pass + radix select over the whole array).  Tuning / test aid.
static bool encode_bracket_enabled() {
    static const bool on = [] {
        const char *e = getenv("ZBRA_HIP_ENCODE_BRACKET");
        return !(e && atoi(e) == 0);
    }();
    return on;
}

// Longest select source (keys) that one workgroup finishes on its own (ZBRA_HIP_SMALL_SELECT_KEYS).
static uint64_t small_select_keys() {
    static const uint64_t v = [] {
        const char *e = getenv("ZBRA_HIP_SMALL_SELECT_KEYS");
        const long long x = e ? atoll(e) : 0;
        return x > 0 ? (uint64_t)x : SMALL_SRC_MAX;
    }();
    return v;
}

static unsigned pack_grid_cap() {
    static const unsigned cap = [] {
        const char *e = getenv("ZBRA_HIP_PACK_GRID");
        const long v = e ? atol(e) : 0;
        return v > 0 ? (unsigned)v : 0x7FFFFFFFu;  // one workgroup per tile, as in decode
    }();
    return cap;
}

// ---- decode, device-resident ---------------------------------------------------------------

// Up to NOALLOC_CHUNK descriptors and their tile scan travel in the kernel arguments (about 3 KB),
// so nothing is staged in pinned memory and no copy is enqueued: the call is kernel launches only.
constexpr uint32_t NOALLOC_CHUNK = 64;
struct NoAllocChunk {
    zbra_array_desc d[NOALLOC_CHUNK];
    uint64_t ts[NOALLOC_CHUNK + 1];  // tile_start of the chunk's arrays and of the one after
};

// Workgroup b handles array first + b of the submission: header parse, its tile_start entry (the
// last workgroup also writes the entry after it) and its slice of the tile -> array table.
__global__ __launch_bounds__(256) void plan_decode_noalloc_kernel(NoAllocChunk c, uint32_t first, uint32_t n, DecodePlan *__restrict__ plans,
                                                                  uint64_t *__restrict__ tile_start, uint32_t *__restrict__ tile_map,
                                                                  zbra_dev_result *results) {
    const uint32_t b = blockIdx.x;
    if (b >= n) return;
    const uint32_t a = first + b;
    const uint64_t t0 = c.ts[b], t1 = c.ts[b + 1];
    if (threadIdx.x == 0) {
        plan_one(c.d[b], plans + a, results ? results + a : nullptr);
        tile_start[a] = t0;
        if (b == n - 1) tile_start[a + 1] = t1;
    }
    for (uint64_t t = t0 + threadIdx.x; t < t1; t += blockDim.x) tile_map[t] = a;
}


// ZBRA_HIP_SMALL_BATCH_ARGS=0: small batches stage their table like large ones (A/B aid)
static bool small_batch_args() {
    static const bool on = [] { const char *e = getenv("ZBRA_HIP_SMALL_BATCH_ARGS"); return !e || atoi(e) != 0; }();
    return on;
}

static int decode_batch_dev(DeviceCtx *ctx, const zbra_array_desc *descs, size_t nd, zbra_dev_result *d_results,
                            hipStream_t stream) {
    if (nd == 0) return ZBRA_OK;
    if (!descs || nd > 0xFFFFFFFFull) return ZBRA_ERR_INVALID_ARG;
    for (size_t i = 0; i < nd; i++) {
        const zbra_array_desc &d = descs[i];
        if (d.count > MAX_COUNT || d.kind > ZBRA_ELEM_U32) return ZBRA_ERR_INVALID_ARG;
        if ((d.len && !d.data) || (d.count && d.len && !d.out)) return ZBRA_ERR_INVALID_ARG;
    }
    if (nd > MAX_BATCH_ARRAYS) {  // arrays are independent: a longer table goes down in pieces
        for (size_t first = 0; first < nd; first += MAX_BATCH_ARRAYS) {
            const size_t n = nd - first < MAX_BATCH_ARRAYS ? nd - first : MAX_BATCH_ARRAYS;
            ZBRA_TRY(decode_batch_dev(ctx, descs + first, n, d_results ? d_results + first : nullptr, stream));
        }
        return ZBRA_OK;
    }
    if (nd == 1) {
        const uint64_t ntiles = decode_tiles_of(descs[0].count, descs[0].kind);
        Slot *s;
        ZBRA_TRY(ctx->acquire(256, &s));
        DecodePlan *plan = (DecodePlan *)s->dev;
        uint64_t *tile_start = (uint64_t *)((char *)s->dev + 64);
        hipLaunchKernelGGL(plan_decode_one_kernel, dim3(1), dim3(64), 0, stream, descs[0], ntiles, plan, tile_start,
                           d_results);
        if (ntiles)
            launch_decode_submission(stream, plan, tile_start, nullptr, 1u, 0, ntiles, decode_tile_order(descs[0].len, descs[0].count, 1),
                                     warm_plan_for(descs, 1, ntiles, descs[0].len));
        int st = hip_status(hipGetLastError());
        int st2 = ctx->release(s, stream);
        return st != ZBRA_OK ? st : st2;
    }
    // A submission that holds narrow arrays AND others is handed to the device with the narrow ones
    // first (a stable partition): their tiles then form one range, which takes the warm-up route,
    // and the other arrays' tiles the range behind it, which takes the default kernel.
    uint64_t ntiles = 0, all_bytes = 0, all_values = 0, n_tiles = 0, n_bytes = 0, n_values = 0;
    size_t n_narrow = 0;
    const bool route = warm_route_available();
    for (size_t i = 0; i < nd; i++) {
        const uint64_t t = decode_tiles_of(descs[i].count, descs[i].kind);
        ntiles += t; all_bytes += descs[i].len; all_values += descs[i].count;
        if (route && warm_narrow(descs[i], false)) { n_narrow++; n_tiles += t; n_bytes += descs[i].len; n_values += descs[i].count; }
    }
    const bool all_narrow = route && warm_plan_for(descs, nd, ntiles, all_bytes).on;
    // worth its two extra launches only when the narrow arrays are a fair share of the work (a quarter of the tiles)
    const bool mixed = !all_narrow && n_narrow && n_bytes >= tuning().warm_min_bytes && n_tiles * 4 >= ntiles;
    if (nd <= NOALLOC_CHUNK && !mixed && small_batch_args()) {
        // A Block's worth of arrays: descriptors and tile scan travel in the plan kernel's arguments
        // (as in the entry point that allocates nothing), so no table is staged in pinned memory and no
        // copy sits in front of the kernels; the workspace slot only lends device memory.
        const size_t off_ts_s = align_up(nd * sizeof(DecodePlan), 64);
        const size_t off_map_s = align_up(off_ts_s + (nd + 1) * sizeof(uint64_t), 64);
        Slot *s;
        ZBRA_TRY(ctx->acquire(off_map_s + ntiles * sizeof(uint32_t), &s, 64));
        DecodePlan *d_plans = (DecodePlan *)s->dev;
        uint64_t *d_ts = (uint64_t *)((char *)s->dev + off_ts_s);
        uint32_t *d_map = (uint32_t *)((char *)s->dev + off_map_s);
        NoAllocChunk c;
        uint64_t run = 0;
        for (uint32_t b = 0; b < (uint32_t)nd; b++) { c.d[b] = descs[b]; c.ts[b] = run; run += decode_tiles_of(descs[b].count, descs[b].kind); }
        c.ts[nd] = run;
        for (uint32_t b = (uint32_t)nd; b < NOALLOC_CHUNK; b++) { std::memset(&c.d[b], 0, sizeof(c.d[b])); c.ts[b + 1] = run; }
        hipLaunchKernelGGL(plan_decode_noalloc_kernel, dim3((unsigned)nd), dim3(256), 0, stream, c, 0u, (uint32_t)nd, d_plans, d_ts, d_map, d_results);
        if (ntiles)
            launch_decode_submission(stream, d_plans, d_ts, d_map, (uint32_t)nd, 0, ntiles, decode_tile_order(all_bytes, all_values, nd),
                                     all_narrow ? warm_plan_of(ntiles, all_bytes) : WarmPlan());
        int st = hip_status(hipGetLastError());
        int st2 = ctx->release(s, stream);
        return st != ZBRA_OK ? st : st2;
    }
    const size_t off_ts = align_up(nd * sizeof(zbra_array_desc), 64);
    const size_t off_orig = align_up(off_ts + (nd + 1) * sizeof(uint64_t), 64);       // caller's index of each array (mixed only)
    const size_t host_bytes = off_orig + (mixed ? nd * sizeof(uint32_t) : 0);
    const size_t off_plan = align_up(host_bytes, 64);
    const size_t off_map = align_up(off_plan + nd * sizeof(DecodePlan), 64);
    const size_t total = off_map + ntiles * sizeof(uint32_t);  // plans and the map live on the device only
    Slot *s;
    ZBRA_TRY(ctx->acquire(total, &s, host_bytes));
    zbra_array_desc *h_descs = (zbra_array_desc *)s->host;
    uint64_t *h_ts = (uint64_t *)((char *)s->host + off_ts);
    uint32_t *h_orig = (uint32_t *)((char *)s->host + off_orig);
    if (!mixed) {
        std::memcpy(h_descs, descs, nd * sizeof(zbra_array_desc));
    } else {
        size_t a = 0, b = n_narrow;
        for (size_t i = 0; i < nd; i++) {
            const size_t at = warm_narrow(descs[i], false) ? a++ : b++;
            h_descs[at] = descs[i];
            h_orig[at] = (uint32_t)i;
        }
    }
    uint64_t run = 0;
    for (size_t i = 0; i < nd; i++) { h_ts[i] = run; run += decode_tiles_of(h_descs[i].count, h_descs[i].kind); }
    h_ts[nd] = run;
    int st = hip_status(hipMemcpyAsync(s->dev, s->host, host_bytes, hipMemcpyHostToDevice, stream));
    if (st == ZBRA_OK) {
        const zbra_array_desc *d_descs = (const zbra_array_desc *)s->dev;
        const uint64_t *d_ts = (const uint64_t *)((char *)s->dev + off_ts);
        const uint32_t *d_orig = mixed ? (const uint32_t *)((char *)s->dev + off_orig) : nullptr;
        DecodePlan *d_plans = (DecodePlan *)((char *)s->dev + off_plan);
        uint32_t *d_map = (uint32_t *)((char *)s->dev + off_map);
        hipLaunchKernelGGL(plan_decode_batch_kernel, dim3((unsigned)nd), dim3(256), 0, stream, d_descs, (uint32_t)nd, d_ts,
                           d_plans, d_map, d_results, d_orig);
        if (ntiles && !mixed) {
            launch_decode_submission(stream, d_plans, d_ts, d_map, (uint32_t)nd, 0, ntiles, decode_tile_order(all_bytes, all_values, nd),
                                     all_narrow ? warm_plan_of(ntiles, all_bytes) : WarmPlan());
        } else if (ntiles) {
            launch_decode_submission(stream, d_plans, d_ts, d_map, (uint32_t)nd, 0, n_tiles, decode_tile_order(n_bytes, n_values, n_narrow),
                                     warm_plan_of(n_tiles, n_bytes));
            launch_decode_submission(stream, d_plans, d_ts, d_map, (uint32_t)nd, n_tiles, ntiles - n_tiles,
                                     decode_tile_order(all_bytes - n_bytes, all_values - n_values, nd - n_narrow), WarmPlan());
        }
        st = hip_status(hipGetLastError());
    }
    int st2 = ctx->release(s, stream);
    return st != ZBRA_OK ? st : st2;
}

// ---- lengths -> offsets, device-resident ------------------------------------------------------

// descs[i].out: count + 1 offsets of array i.  One submission: plan, per-tile sums, a two-level scan
// of the tile sums of all arrays as one sequence, the write pass.
static int decode_offsets_batch_dev(DeviceCtx *ctx, const zbra_array_desc *descs, size_t nd, zbra_dev_result *d_results, hipStream_t stream) {
    if (nd == 0) return ZBRA_OK;
    if (!descs) return ZBRA_ERR_INVALID_ARG;
    uint64_t ntiles = 0, all_bytes = 0;
    bool any_empty = false;
    // Widest packed width any array of the submission can have: an array whose header passes the
    // checks has ceil(count * w / 8) payload bytes inside its len - 13, so w <= (len - 13) * 8 / count.
    // It sizes the LDS images of the sums pass (narrow lengths arrays then leave room for more waves).
    uint32_t wmax = 1;
    for (size_t i = 0; i < nd; i++) {
        const zbra_array_desc &d = descs[i];
        if (d.count > MAX_COUNT || d.kind != ZBRA_ELEM_I64 || !d.out || (d.len && !d.data)) return ZBRA_ERR_INVALID_ARG;
        ntiles += (d.count + OFF_TILE - 1) / OFF_TILE;
        all_bytes += d.len;
        any_empty = any_empty || d.count == 0;
        if (d.count >= (uint64_t)OFF_TILE && d.len > 13) {  // shorter arrays have no full tile
            const uint64_t w = ((d.len - 13) * 8 + d.count - 1) / d.count;
            wmax = w >= 31 ? 31u : (uint32_t)w > wmax ? (uint32_t)w : wmax;
        }
    }
    const uint32_t image_dwords = offsets_image_dwords(wmax);
    // one workgroup per tile in the write pass: a longer table goes down in pieces (arrays are independent)
    const uint64_t most_tiles = launch_limit() / OFF_THREADS;
    if (nd > 1 && (ntiles > most_tiles || nd > MAX_BATCH_ARRAYS)) {
        size_t first = 0;
        while (first < nd) {
            size_t end = first;
            uint64_t t = 0;
            while (end < nd && end - first < MAX_BATCH_ARRAYS) {
                const uint64_t ti = (descs[end].count + OFF_TILE - 1) / OFF_TILE;
                if (end > first && t + ti > most_tiles) break;
                t += ti;
                end++;
            }
            ZBRA_TRY(decode_offsets_batch_dev(ctx, descs + first, end - first, d_results ? d_results + first : nullptr, stream));
            first = end;
        }
        return ZBRA_OK;
    }
    const uint64_t nchunks = (ntiles + SCAN_CHUNK - 1) / SCAN_CHUNK;
    const size_t off_ts = align_up(nd * sizeof(zbra_array_desc), 64);
    const size_t host_bytes = off_ts + (nd + 1) * sizeof(uint64_t);
    const size_t off_plan = align_up(host_bytes, 64);
    const size_t off_map = align_up(off_plan + nd * sizeof(DecodePlan), 64);
    const size_t off_sums = align_up(off_map + (nd > 1 ? (size_t)ntiles * sizeof(uint32_t) : 0), 64);
    const size_t total = off_sums + (size_t)(ntiles + nchunks + 1) * sizeof(uint64_t);
    Slot *s;
    ZBRA_TRY(ctx->acquire(total, &s, host_bytes));
    std::memcpy(s->host, descs, nd * sizeof(zbra_array_desc));
    uint64_t *h_ts = (uint64_t *)((char *)s->host + off_ts);
    uint64_t run = 0;
    for (size_t i = 0; i < nd; i++) { h_ts[i] = run; run += (descs[i].count + OFF_TILE - 1) / OFF_TILE; }
    h_ts[nd] = run;
    int st = hip_status(hipMemcpyAsync(s->dev, s->host, host_bytes, hipMemcpyHostToDevice, stream));
    if (st == ZBRA_OK) {
        const zbra_array_desc *d_descs = (const zbra_array_desc *)s->dev;
        const uint64_t *d_ts = (const uint64_t *)((char *)s->dev + off_ts);
        DecodePlan *d_plans = (DecodePlan *)((char *)s->dev + off_plan);
        uint32_t *d_map = nd > 1 ? (uint32_t *)((char *)s->dev + off_map) : nullptr;
        uint64_t *tile_sums = (uint64_t *)((char *)s->dev + off_sums);
        uint64_t *chunk_base = tile_sums + ntiles;
        if (nd == 1)
            hipLaunchKernelGGL(plan_decode_one_kernel, dim3(1), dim3(64), 0, stream, descs[0], ntiles, d_plans, (uint64_t *)d_ts, d_results);
        else
            hipLaunchKernelGGL(plan_decode_batch_kernel, dim3((unsigned)nd), dim3(256), 0, stream, d_descs, (uint32_t)nd, d_ts, d_plans, d_map,
                               d_results, (const uint32_t *)nullptr);
        if (any_empty) hipLaunchKernelGGL(offsets_empty_kernel, dim3((unsigned)((nd + 255) / 256)), dim3(256), 0, stream, d_descs, (uint32_t)nd);
        if (ntiles) {
            const uint64_t tpw_want = ntiles / OFF_SUM_MIN_WAVES;
            const uint32_t tpw = tpw_want < 1 ? 1u : tpw_want > OFF_SUM_TPW_MAX ? OFF_SUM_TPW_MAX : (uint32_t)tpw_want;
            const uint64_t per_wg = (uint64_t)OFF_SUM_WAVES * tpw;
#define ZBRA_LAUNCH_SUMS(WM)                                                                                                         \
    hipLaunchKernelGGL((offsets_sums_kernel<WM>), dim3((unsigned)((ntiles + per_wg - 1) / per_wg)), dim3(64 * OFF_SUM_WAVES),          \
                       (size_t)OFF_SUM_WAVES * image_dwords * sizeof(uint32_t), stream, d_plans, d_ts, d_map, (uint32_t)nd, ntiles, tile_sums, \
                       image_dwords, tpw)
            if (wmax <= 8) ZBRA_LAUNCH_SUMS(8);
            else if (wmax <= 16) ZBRA_LAUNCH_SUMS(16);
            else ZBRA_LAUNCH_SUMS(31);
#undef ZBRA_LAUNCH_SUMS
            hipLaunchKernelGGL(offsets_scan_chunks_kernel, dim3((unsigned)nchunks), dim3(SCAN_CHUNK), 0, stream, ntiles, tile_sums, chunk_base);
            hipLaunchKernelGGL(offsets_scan_totals_kernel, dim3(1), dim3(SCAN_CHUNK), 0, stream, nchunks, chunk_base);
            // the write pass finds the packed bytes in the memory-side cache (256 MB) if the sums pass left them there:
            // all of them when they are few enough, else the ones it read last, which a back-to-front pass meets first
            hipLaunchKernelGGL(offsets_write_kernel, dim3((unsigned)ntiles), dim3(OFF_THREADS), 0, stream, d_plans, d_ts, d_map, (uint32_t)nd, tile_sums,
                               chunk_base, all_bytes > ((uint64_t)208 << 20) ? 1u : 0u);
        }
        st = hip_status(hipGetLastError());
    }
    int st2 = ctx->release(s, stream);
    return st != ZBRA_OK ? st : st2;
}

// ---- Enum rows -> (variant, rank), device-resident ---------------------------------------------

static int enum_rows_batch_dev(DeviceCtx *ctx, const zbra_enum_desc *descs, size_t nd, const uint32_t *variant_tags, uint32_t n_variants,
                               zbra_dev_result *d_results, hipStream_t stream) {
    if (nd == 0) return ZBRA_OK;
    if (!descs || n_variants > (uint32_t)ENUM_MAX_VARIANTS || (n_variants && !variant_tags)) return ZBRA_ERR_INVALID_ARG;
    uint64_t ntiles = 0, all_rows = 0;
    for (size_t i = 0; i < nd; i++) {
        const zbra_enum_desc &d = descs[i];
        if (d.n > MAX_COUNT || (n_variants && !d.counts) || (d.n && (!d.tags || !d.variant || !d.rank))) return ZBRA_ERR_INVALID_ARG;
        ntiles += (d.n + ENUM_TILE - 1) / ENUM_TILE;
        all_rows += d.n;
    }
    // one workgroup per tile in the write pass: a longer table goes down in pieces (arrays are independent)
    const uint64_t most_tiles = launch_limit() / ENUM_THREADS;
    if (nd > 1 && (ntiles > most_tiles || nd > MAX_BATCH_ARRAYS)) {
        size_t first = 0;
        while (first < nd) {
            size_t end = first;
            uint64_t t = 0;
            while (end < nd && end - first < MAX_BATCH_ARRAYS) {
                const uint64_t ti = (descs[end].n + ENUM_TILE - 1) / ENUM_TILE;
                if (end > first && t + ti > most_tiles) break;
                t += ti;
                end++;
            }
            ZBRA_TRY(enum_rows_batch_dev(ctx, descs + first, end - first, variant_tags, n_variants, d_results ? d_results + first : nullptr, stream));
            first = end;
        }
        return ZBRA_OK;
    }
    const uint64_t nchunks = (ntiles + SCAN_CHUNK - 1) / SCAN_CHUNK;
    // host-filled: variant tags, descriptors, first tile of each array
    const size_t off_desc = align_up(ENUM_MAX_VARIANTS * sizeof(uint32_t), 64);
    const size_t off_ts = align_up(off_desc + nd * sizeof(zbra_enum_desc), 64);
    // ... and every array's "first row with an unknown tag" word, which starts as "none" (all ones)
    const size_t off_bad = align_up(off_ts + (nd + 1) * sizeof(uint64_t), 64);
    const size_t host_bytes = off_bad + nd * sizeof(unsigned long long);
    // device only: tile -> array, per-variant tile counts and chunk prefixes
    const size_t off_map = align_up(host_bytes, 64);
    const size_t off_counts = align_up(off_map + (nd > 1 ? (size_t)ntiles * sizeof(uint32_t) : 0), 64);
    const size_t total = off_counts + (size_t)n_variants * (ntiles + nchunks + 1) * sizeof(uint64_t);
    Slot *s;
    ZBRA_TRY(ctx->acquire(total, &s, host_bytes));
    uint32_t *h_vt = (uint32_t *)s->host;
    for (uint32_t v = 0; v < (uint32_t)ENUM_MAX_VARIANTS; v++) h_vt[v] = v < n_variants ? variant_tags[v] : 0;
    std::memcpy((char *)s->host + off_desc, descs, nd * sizeof(zbra_enum_desc));
    uint64_t *h_ts = (uint64_t *)((char *)s->host + off_ts);
    uint64_t run = 0;
    for (size_t i = 0; i < nd; i++) { h_ts[i] = run; run += (descs[i].n + ENUM_TILE - 1) / ENUM_TILE; }
    h_ts[nd] = run;
    std::memset((char *)s->host + off_bad, 0xFF, nd * sizeof(unsigned long long));
    const uint32_t *d_vt = (const uint32_t *)s->dev;
    unsigned long long *d_bad = (unsigned long long *)((char *)s->dev + off_bad);
    uint64_t *tile_counts = (uint64_t *)((char *)s->dev + off_counts);
    uint64_t *chunk_base = tile_counts + (size_t)n_variants * ntiles;
    EnumBatch batch;
    batch.arrays = (const zbra_enum_desc *)((char *)s->dev + off_desc);
    batch.tile_start = (const uint64_t *)((char *)s->dev + off_ts);
    batch.tile_map = nd > 1 ? (const uint32_t *)((char *)s->dev + off_map) : nullptr;
    batch.nd = (uint32_t)nd;
    batch.ntiles = ntiles;
    int st = hip_status(hipMemcpyAsync(s->dev, s->host, host_bytes, hipMemcpyHostToDevice, stream));
    if (st == ZBRA_OK) {
        if (nd > 1)  // the tile -> array table (a single array needs none; the unknown-tag words came down with the descriptors)
            hipLaunchKernelGGL(enum_map_kernel, dim3((unsigned)nd), dim3(256), 0, stream, batch, (uint32_t *)((char *)s->dev + off_map));
        if (ntiles) {
            if (n_variants <= (uint32_t)ENUM_FEW) {
                const uint64_t per_wg = (uint64_t)ENUM_WAVES * ENUM_COUNT_TILES;
                hipLaunchKernelGGL(enum_counts_few_kernel, dim3((unsigned)((ntiles + per_wg - 1) / per_wg)), dim3(ENUM_THREADS), 0, stream, batch, d_vt,
                                   n_variants, tile_counts, d_bad);
            } else {
                hipLaunchKernelGGL(enum_counts_kernel, dim3((unsigned)ntiles), dim3(ENUM_THREADS), 0, stream, batch, d_vt, n_variants, tile_counts, d_bad);
            }
            if (n_variants)
                hipLaunchKernelGGL(enum_scan_chunks_kernel, dim3((unsigned)nchunks, n_variants), dim3(SCAN_CHUNK), 0, stream, ntiles, nchunks,
                                   tile_counts, chunk_base);
        }
        // few arrays and few chunks: the totals' scan, the arrays' counts and the status words in one launch of one
        // workgroup (the unknown-tag words are final after the counting pass: the write pass does not touch them)
        const bool merged = n_variants && (uint64_t)nd * n_variants <= 65536 && nchunks * n_variants <= 65536;
        if (merged) {
            hipLaunchKernelGGL(enum_totals_kernel, dim3(1), dim3(SCAN_CHUNK), 0, stream, batch, n_variants, nchunks, tile_counts, chunk_base, d_bad, d_results);
        } else if (n_variants) {
            if (ntiles) hipLaunchKernelGGL(enum_scan_totals_kernel, dim3(n_variants), dim3(SCAN_CHUNK), 0, stream, nchunks, chunk_base);
            // with no tile at all every count is zero, and the kernel reads nothing of the scans
            hipLaunchKernelGGL(enum_array_counts_kernel, dim3((unsigned)((nd * n_variants + 255) / 256)), dim3(256), 0, stream, batch, n_variants, nchunks,
                               tile_counts, chunk_base);
        }
        if (ntiles) {
            const int rev = tuning().enum_reverse;
            hipLaunchKernelGGL(enum_write_kernel, dim3((unsigned)ntiles), dim3(ENUM_THREADS), 0, stream, batch, d_vt, n_variants, nchunks, tile_counts,
                               chunk_base, rev < 0 ? (all_rows * 4 > ((uint64_t)208 << 20) ? 1u : 0u) : (uint32_t)rev);
        }
        if (!merged) hipLaunchKernelGGL(enum_report_kernel, dim3((unsigned)((nd + 255) / 256)), dim3(256), 0, stream, batch, d_bad, d_results);
        st = hip_status(hipGetLastError());
    }
    int st2 = ctx->release(s, stream);
    return st != ZBRA_OK ? st : st2;
}

// ---- decode, device-resident, no allocation --------------------------------------------------

static int decode_batch_dev_noalloc(DeviceCtx *ctx, const zbra_array_desc *descs, size_t nd, zbra_dev_result *d_results,
                                    hipStream_t stream) {
    if (nd == 0) return ZBRA_OK;
    if (!descs) return ZBRA_ERR_INVALID_ARG;
    if (nd > (size_t)ZBRA_HIP_NOALLOC_MAX_ARRAYS) return ZBRA_ERR_NOALLOC_LIMIT;
    uint64_t ntiles = 0, all_bytes = 0, all_values = 0;
    for (size_t i = 0; i < nd; i++) {
        const zbra_array_desc &d = descs[i];
        if (d.count > MAX_COUNT || d.kind > ZBRA_ELEM_U32) return ZBRA_ERR_INVALID_ARG;
        if ((d.len && !d.data) || (d.count && d.len && !d.out)) return ZBRA_ERR_INVALID_ARG;
        ntiles += decode_tiles_of(d.count, d.kind);
        all_bytes += d.len;
        all_values += d.count;
    }
    if (ntiles > (uint64_t)ZBRA_HIP_NOALLOC_MAX_TILES) return ZBRA_ERR_NOALLOC_LIMIT;
    NoAllocScratch *sc = (NoAllocScratch *)ctx->noalloc_scratch;  // this device's address, never the symbol's name
    // Nothing in the scratch has to start at zero: plans[0, nd), tile_start[0, nd] and
    // tile_map[0, ntiles) are all written by this call's plan kernels before the tile kernel,
    // which reads nothing else, runs behind them on the same stream.
    uint64_t run = 0;
    for (size_t first = 0; first < nd; first += NOALLOC_CHUNK) {
        NoAllocChunk c;
        const uint32_t n = (uint32_t)(nd - first < NOALLOC_CHUNK ? nd - first : NOALLOC_CHUNK);
        for (uint32_t b = 0; b < n; b++) {
            c.d[b] = descs[first + b];
            c.ts[b] = run;
            run += decode_tiles_of(descs[first + b].count, descs[first + b].kind);
        }
        c.ts[n] = run;
        for (uint32_t b = n; b < NOALLOC_CHUNK; b++) { std::memset(&c.d[b], 0, sizeof(c.d[b])); c.ts[b + 1] = run; }
        hipLaunchKernelGGL(plan_decode_noalloc_kernel, dim3(n), dim3(256), 0, stream, c, (uint32_t)first, n, sc->plans, sc->tile_start,
                           sc->tile_map, d_results);
    }
    if (ntiles) launch_decode(stream, sc->plans, sc->tile_start, sc->tile_map, (uint32_t)nd, ntiles, decode_tile_order(all_bytes, all_values, nd));
    return hip_status(hipGetLastError());
}

// ---- encode, device-resident ---------------------------------------------------------------

struct EncodeItem {
    const void *values;
    uint64_t n;
    uint32_t kind;
    uint8_t *out;
    uint64_t cap;
};

enum EncodeMode { ENCODE_FULL, ENCODE_MEDIAN_ONLY };

// Framed output: every E is placed behind its `u32 count | u32 clen` header, back to back from
// `out` (the items' own `out` / `cap` are ignored); offsets = nd + 1 device words.
struct FrameTarget {
    uint8_t *out = nullptr;
    uint64_t cap = 0;
    uint64_t *offsets = nullptr;
};

static int encode_batch_dev(DeviceCtx *ctx, const EncodeItem *items, size_t nd, zbra_dev_result *d_results,
                            hipStream_t stream, EncodeMode mode = ENCODE_FULL, const FrameTarget *frames = nullptr) {
    if (nd == 0) return ZBRA_OK;
    if (!items || nd > 0xFFFFFFFFull) return ZBRA_ERR_INVALID_ARG;
    if (frames && (!frames->offsets || (frames->cap && !frames->out))) return ZBRA_ERR_INVALID_ARG;
    for (size_t i = 0; i < nd; i++) {
        if (items[i].n > MAX_COUNT || items[i].kind > KIND_U64_PLAIN) return ZBRA_ERR_INVALID_ARG;
        if (items[i].n && (!items[i].values || (mode == ENCODE_FULL && !frames && !items[i].out))) return ZBRA_ERR_INVALID_ARG;
    }
    if (nd > MAX_BATCH_ARRAYS) {
        // the frames of one call are laid out by one scan over its arrays; plain encodes are independent
        if (frames) return ZBRA_ERR_INVALID_ARG;
        for (size_t first = 0; first < nd; first += MAX_BATCH_ARRAYS) {
            const size_t n = nd - first < MAX_BATCH_ARRAYS ? nd - first : MAX_BATCH_ARRAYS;
            ZBRA_TRY(encode_batch_dev(ctx, items + first, n, d_results ? d_results + first : nullptr, stream, mode, nullptr));
        }
        return ZBRA_OK;
    }
    const bool bracket = encode_bracket_enabled();
    const uint64_t small_max = small_select_keys();
    // Which arrays need what.  "listed": may go through the bracket pass and the select (arrays of
    // up to SAMPLE values are sorted whole by the sample kernel and need no pass; plain u64 values
    // have no median).  "long": may have too much to read for one finishing workgroup, so the
    // many-workgroup levels are launched and these arrays get their histograms and key buffer.
    auto listed = [&](const EncodeItem &it) { return it.n && it.kind != KIND_U64_PLAIN && (!bracket || it.n > (uint64_t)SAMPLE); };
    auto is_long = [&](const EncodeItem &it) { return listed(it) && it.n > (uint64_t)CAND_DIV * small_max; };
    // candidate lists of the bracket select: CAND_SEGS segments, n/CAND_DIV values in all, per array
    // that is long enough for it
    auto seg_cap_of = [&](const EncodeItem &it) -> uint64_t {
        if (!bracket || it.kind == KIND_U64_PLAIN || it.n <= (uint64_t)SAMPLE) return 0;
        return it.n / CAND_DIV / CAND_SEGS / TILE * TILE;
    };
    uint32_t n_list = 0, n_big = 0;
    size_t scratch_bytes = 0;
    bool may_need_or = false;  // u32 values span less than 2^63: their width follows from minimum, maximum and midpoint
    for (size_t i = 0; i < nd; i++) {
        may_need_or = may_need_or || (items[i].n && items[i].kind != ZBRA_ELEM_U32);
        n_list += listed(items[i]) ? 1u : 0u;
        n_big += is_long(items[i]) ? 1u : 0u;
        scratch_bytes += seg_cap_of(items[i]) * CAND_SEGS * sizeof(int64_t);
    }
    const size_t off_small = align_up(scratch_bytes, 256);  // gathered keys of the long arrays, behind the candidate lists
    scratch_bytes = off_small + (size_t)n_big * GATHER_SEGS * GATHER_CAP * sizeof(uint64_t);
    const size_t off_ts = align_up(nd * sizeof(EncodeJob), 64);
    const size_t off_list = align_up(off_ts + (nd + 1) * sizeof(uint64_t), 64);  // the listed arrays
    const size_t host_bytes = off_list + (size_t)n_list * sizeof(uint32_t);      // what the host fills
    // device only, zeroed by every call: histograms and counters of the listed arrays, level
    // histograms of the long ones, the list of arrays left to the levels {count, arrays...}
    const size_t off_hist = align_up(host_bytes, 256);
    const size_t off_ctr = off_hist + (size_t)n_list * BINS * sizeof(uint32_t);
    const size_t off_lvl = off_ctr + (size_t)n_list * sizeof(BracketCounters);
    const size_t off_pend = off_lvl + (size_t)n_big * SEL_LEVELS * BINS * sizeof(uint32_t);
    const size_t off_map = align_up(off_pend + ((size_t)n_big + 2) * sizeof(uint32_t), 64);  // behind the list: the OR pass's flag; then tile -> array, written by the set-up kernel
    uint64_t all_tiles = 0;
    for (size_t i = 0; i < nd; i++) all_tiles += tiles_of(items[i].n);
    const size_t total = off_map + (nd > 1 ? (size_t)all_tiles * sizeof(uint32_t) : 0);
    // candidate lists and gathered keys live in the stream's scratch buffer: this call's kernels
    // follow the previous call's in stream order, and the lock keeps another thread's enqueues on
    // the same stream from interleaving with them
    StreamScratch *ss;
    ZBRA_TRY(ctx->lock_scratch_of(stream, &ss));
    std::lock_guard<std::mutex> scratch_guard(ss->mu, std::adopt_lock);
    ZBRA_TRY(DeviceCtx::grow_scratch(ss, scratch_bytes));
    char *enc_mem = (char *)ss->mem.ptr;  // (not `scratch`: that name is the no-allocation entry point's device symbol)
    Slot *s;
    ZBRA_TRY(ctx->acquire(total, &s, host_bytes));
    EncodeJob *h_jobs = (EncodeJob *)s->host;
    uint64_t *h_ts = (uint64_t *)((char *)s->host + off_ts);
    uint32_t *h_list = (uint32_t *)((char *)s->host + off_list);
    uint32_t i_list = 0, i_big = 0;
    uint64_t run = 0;
    size_t scratch_off = 0;
    for (size_t i = 0; i < nd; i++) {
        EncodeJob j;
        std::memset(&j, 0, sizeof(j));
        j.values = items[i].values; j.out = items[i].out; j.n = items[i].n; j.cap = items[i].cap; j.kind = items[i].kind;
        j.min_b = ~0ull; j.max_b = 0; j.next_above = ~0ull; j.above_key = ~0ull;
        for (int l = 0; l < SEL_LEVELS; l++) j.match_min[l] = ~0ull;  // match_max starts at 0
        j.sel_values = items[i].values; j.sel_n = items[i].n; j.sel_kind = items[i].kind;
        j.seg_cap = seg_cap_of(items[i]);
        if (j.seg_cap) {
            j.cand = (int64_t *)(enc_mem + scratch_off);
            scratch_off += j.seg_cap * CAND_SEGS * sizeof(int64_t);
        }
        if (listed(items[i])) {
            j.hist = (uint32_t *)((char *)s->dev + off_hist) + (size_t)i_list * BINS;
            j.counters = (BracketCounters *)((char *)s->dev + off_ctr) + i_list;
            h_list[i_list++] = (uint32_t)i;
            if (is_long(items[i])) {
                j.lvl_hist = (uint32_t *)((char *)s->dev + off_lvl) + (size_t)i_big * SEL_LEVELS * BINS;
                j.small = (uint64_t *)(enc_mem + off_small) + (size_t)i_big * GATHER_SEGS * GATHER_CAP;
                i_big++;
            }
        }
        h_jobs[i] = j;
        h_ts[i] = run;
        run += tiles_of(items[i].n);
    }
    h_ts[nd] = run;
    const uint64_t ntiles = run;
    EncodeJob *d_jobs = (EncodeJob *)s->dev;
    const uint32_t *d_list = (const uint32_t *)((char *)s->dev + off_list);
    uint32_t *d_pend = (uint32_t *)((char *)s->dev + off_pend);
    uint32_t *d_or_flag = d_pend + n_big + 1;
    const uint64_t *d_ts = (const uint64_t *)((char *)s->dev + off_ts);
    int st = hip_status(hipMemcpyAsync(s->dev, s->host, host_bytes, hipMemcpyHostToDevice, stream));
    if (st == ZBRA_OK) st = hip_status(hipMemsetAsync((char *)s->dev + off_hist, 0, off_map - off_hist, stream));
    uint32_t *d_map = nd > 1 ? (uint32_t *)((char *)s->dev + off_map) : nullptr;
    if (st == ZBRA_OK) {
        // Workgroups of the bracket pass, each with one contiguous run of tiles.  One 2^28-row
        // column: 512 -> 0.877 ms for the encode, 768 -> 0.828, 1024 -> 0.838, 2048 -> 0.877,
        // 4096 -> 0.895.  Batches of a thousand arrays want shorter runs (better balance, at the
        // price of more end-of-run merges): the pass over a thousand 2^20-row arrays takes 1807 us
        // with 1024 workgroups, 1754 with 2048, 1649 with 4096, 1643 with 8192, 1688 with 16384,
        // 2122 with 65536.
        static const unsigned bracket_grid = [] { const char *e = getenv("ZBRA_HIP_BRACKET_GRID"); return e ? (unsigned)atol(e) : 0u; }();
        // A run's end costs a merge of the workgroup's histogram into the array's (thousands of
        // atomics), so a batch's runs are kept at 32 tiles or more: 2^26 rows as 4 to 1024 arrays
        // took 183-208 us in this pass with 8192 workgroups of 4 tiles each, 100-114 us with 1024.
        const uint64_t batch_grid = ntiles / 32 < STREAM_GRID_CAP / 2 ? STREAM_GRID_CAP / 2 : ntiles / 32 < STREAM_GRID_CAP * 4 ? ntiles / 32 : STREAM_GRID_CAP * 4;
        // every array u32 (enum tags): the 32-bit form of the pass (ZBRA_HIP_BRACKET_U32=0: the generic one, A/B aid).
        // Its tiles are half the bytes, so a run's end weighs twice as much: runs of 128 tiles or more
        // (128 tag arrays of 2^22 rows: 0.904 ms for the encode with 8192 workgroups, 0.878 with 4096, 0.857 with 2048 or 1024).
        static const bool bracket_u32_on = [] { const char *e = getenv("ZBRA_HIP_BRACKET_U32"); return !e || atoi(e) != 0; }();
        bool bracket_u32 = bracket_u32_on;
        for (size_t i = 0; bracket_u32 && i < nd; i++) bracket_u32 = items[i].kind == ZBRA_ELEM_U32;
        const uint64_t batch_grid_u32 = ntiles / 128 < STREAM_GRID_CAP / 2 ? STREAM_GRID_CAP / 2 : ntiles / 128 < STREAM_GRID_CAP * 4 ? ntiles / 128 : STREAM_GRID_CAP * 4;
        const unsigned gs = grid_for(ntiles, bracket_grid ? bracket_grid : nd == 1 ? STREAM_GRID_CAP / 2 : (unsigned)(bracket_u32 ? batch_grid_u32 : batch_grid));
        const unsigned ga = (unsigned)((nd + 255) / 256);
        // workgroups of a many-workgroup select level, dealt to the arrays it finds pending
        static const unsigned level_grid = [] { const char *e = getenv("ZBRA_HIP_LEVEL_GRID"); return e ? (unsigned)atol(e) : 0u; }();
        static const unsigned sparse_tiles = [] { const char *e = getenv("ZBRA_HIP_SPARSE_TILES"); return e ? (unsigned)atol(e) : 0u; }();
        // 512 workgroups per array that may be pending, up to 4096: a single 2^27-row column with a
        // wide bracket encodes in 0.569 ms with 512, 0.582 with 1024 or 256, 0.596 with 4096 (most of
        // a large grid only passes through, behind the workgroups that work); the first level of
        // a thousand arrays of 2^20 rows takes 718 us with 1024 workgroups, 242 with 2048, 136 with
        // 4096, 161 with 8192, 222 with 16384 (scripts/sweep_level_grid_batches.sh)
        const unsigned gh = level_grid ? level_grid : n_big >= 8 ? 4096u : 512u * n_big;
        if (ntiles) {
            hipLaunchKernelGGL(enc_sample_kernel, dim3((unsigned)nd), dim3(SAMPLE_THREADS), 0, stream, d_jobs, (uint32_t)nd,
                               bracket ? 1u : 0u);
            // arrays outside the list (sorted whole by the sample kernel, or plain u64) need
            // neither the bracket pass nor any select launch
            if (n_list) {
                if (bracket_u32) hipLaunchKernelGGL(enc_bracket_u32_kernel, dim3(gs), dim3(THREADS), 0, stream, d_jobs, d_ts, (uint32_t)nd, ntiles);
                else hipLaunchKernelGGL(enc_bracket_kernel, dim3(gs), dim3(THREADS), 0, stream, d_jobs, d_ts, (uint32_t)nd, ntiles);
            }
        }
        hipLaunchKernelGGL(enc_setup_kernel, dim3((unsigned)nd), dim3(THREADS), 0, stream, d_jobs, (uint32_t)nd, d_pend, small_max,
                           d_ts, d_map, d_or_flag);
        if (ntiles) {
            // the levels are only launched when an array could be too long for one finishing
            // workgroup (its length says so), and end at their first load when none was
            for (uint32_t L = 1; n_big && L <= (uint32_t)SEL_LEVELS; L++)
                hipLaunchKernelGGL(enc_level_kernel, dim3(gh), dim3(THREADS), 0, stream, d_jobs, d_pend, L, sparse_tiles);
            if (n_list)
                hipLaunchKernelGGL(enc_finish_select_kernel, dim3(n_list), dim3(THREADS), 0, stream, d_jobs, d_list, n_list);
            if (mode == ENCODE_FULL && may_need_or) {
                // Few arrays of a batch need this pass (a span of 2^63 or more whose extremes do not
                // already take 64 bits), and a contiguous run per resident workgroup would leave
                // each of them to the two or three workgroups whose runs it crosses: short runs of
                // 16 tiles instead, most of which end at their first look (a thousand 2^20-row
                // arrays, two of them wide: 187 -> 47 us).
                const unsigned go = nd == 1 ? gs : grid_for(ntiles / 16 + 1, 32768);
                hipLaunchKernelGGL(enc_or_kernel, dim3(go), dim3(THREADS), 0, stream, d_jobs, d_ts, (uint32_t)nd, ntiles, d_or_flag);
            }
        }
        if (mode == ENCODE_FULL) {
            if (frames)
                hipLaunchKernelGGL(enc_frame_layout_kernel, dim3(1), dim3(nd > 1024 ? 1024 : 256), 0, stream, d_jobs, (uint32_t)nd, frames->out,
                                   frames->cap, frames->offsets);
            // a single array with values: the pack kernel's workgroups do what the finalize kernel would
            // (one dependent launch less: 6 us); ZBRA_HIP_ENCODE_FOLD=0 keeps the separate launch (A/B aid)
            static const bool fold_on = [] { const char *e = getenv("ZBRA_HIP_ENCODE_FOLD"); return !e || atoi(e) != 0; }();
            const bool fold = fold_on && nd == 1 && ntiles && !frames;
            if (!fold) hipLaunchKernelGGL(enc_finalize_kernel, dim3(ga), dim3(256), 0, stream, d_jobs, (uint32_t)nd, d_results);
            if (ntiles) {
                // One workgroup per tile.  Batches deal each XCD one contiguous eighth of the tiles
                // (a thousand 2^20-row arrays: 3.98 against 4.11 ms for the encode, the 1024 ragged
                // columns 4.70 against 5.03); a single array takes the plain order, which is as fast
                // at 2^28 rows and 1-4 % faster at 2^27 (pack at width 12: 213 against 221 us,
                // raw: 350 against 364).  ZBRA_HIP_PACK_XCD=0|1 forces one or the other.
                // past the work-items of one dispatch the workgroups take several tiles each, in the plain order
                const uint64_t pack_most = launch_limit() / THREADS ? launch_limit() / THREADS : 1;
                const unsigned gp = grid_for(ntiles, (uint64_t)pack_grid_cap() < pack_most ? pack_grid_cap() : (unsigned)pack_most);
                static const int pack_xcd_env = [] { const char *e = getenv("ZBRA_HIP_PACK_XCD"); return e ? atoi(e) : -1; }();
                const bool pack_xcd = pack_xcd_env >= 0 ? pack_xcd_env != 0 : nd > 1;
                const uint32_t chunked = (pack_xcd && (uint64_t)gp == ntiles && ntiles >= 64) ? 1u : 0u;
                // all arrays u32 and of one length (the tag columns of equal Blocks), or a single u32 array:
                // tiles of 4096 tags, numbered per array (ZBRA_HIP_PACK_U32=0: the generic kernel, A/B aid)
                static const bool pack_u32_on = [] { const char *e = getenv("ZBRA_HIP_PACK_U32"); return !e || atoi(e) != 0; }();
                bool all_u32 = pack_u32_on;
                for (size_t i = 0; all_u32 && i < nd; i++) all_u32 = items[i].kind == ZBRA_ELEM_U32 && items[i].n == items[0].n;
                const uint64_t per32 = (items[0].n + TILE32 - 1) / TILE32, ntiles32 = per32 * nd;
                if (all_u32 && per32 <= 0xFFFFFFFFull && ntiles32 <= 0xFFFFFFFFull && ntiles32 <= pack_most) {
                    const uint32_t ch32 = (pack_xcd && ntiles32 >= 64) ? 1u : 0u;
                    const uint64_t inv32 = per32 > 1 ? ~0ull / per32 + 1 : 0;  // ceil(2^64 / per32); one tile per array needs none
                    if (fold)
                        hipLaunchKernelGGL(enc_pack_u32_kernel<true>, dim3((unsigned)ntiles32), dim3(THREADS), 0, stream, d_jobs, (uint32_t)per32, inv32,
                                           ntiles32, ch32, d_results);
                    else
                        hipLaunchKernelGGL(enc_pack_u32_kernel<false>, dim3((unsigned)ntiles32), dim3(THREADS), 0, stream, d_jobs, (uint32_t)per32, inv32,
                                           ntiles32, ch32, d_results);
                } else if (fold)
                    hipLaunchKernelGGL(enc_pack_kernel<true>, dim3(gp), dim3(THREADS), 0, stream, d_jobs, d_ts, d_map, (uint32_t)nd, ntiles, chunked,
                                       d_results);
                else
                    hipLaunchKernelGGL(enc_pack_kernel<false>, dim3(gp), dim3(THREADS), 0, stream, d_jobs, d_ts, d_map, (uint32_t)nd, ntiles, chunked,
                                       d_results);
            }
        } else {
            hipLaunchKernelGGL(enc_report_median_kernel, dim3(ga), dim3(256), 0, stream, d_jobs, (uint32_t)nd, d_results);
        }
        st = hip_status(hipGetLastError());
    }
    int st2 = ctx->release(s, stream);
    return st != ZBRA_OK ? st : st2;
}

// ---- host-pointer helpers ------------------------------------------------------------------

// Host-side reading of the header, for the synchronous calls: same rules as plan_one.
static int host_check_decode(const uint8_t *data, size_t len, size_t count, size_t *out_count) {
    *out_count = 0;
    if (len == 0) return ZBRA_OK;
    if (len < 12) return ZBRA_ERR_SHORT_HEADER;
    uint32_t pl;
    std::memcpy(&pl, data + 8, 4);  // x86-64 host: little-endian
    if (len < 12 + (size_t)pl) return ZBRA_ERR_SHORT_PACKED;
    if (pl == 0 || count == 0) return ZBRA_OK;
    const uint32_t w = data[12];
    if (w >= 1 && w < 32 && (uint64_t)(pl - 1) < ((uint64_t)count * w + 7) / 8) return ZBRA_ERR_TRUNCATED;
    *out_count = count;
    return ZBRA_OK;
}

// Large arrays: the header is readable on the host, so the array is cut into pieces of
// HOST_PIECE_BYTES of output whose plans the host writes itself (a piece of a constant-width array
// is an array with the same midpoint and width that starts at a later payload byte).  Piece c+1
// is uploaded and decoded on the context's stream while piece c travels back on a second one: the
// two directions of the link work at the same time.
constexpr size_t HOST_PIECE_BYTES = (size_t)64 << 20;

static int decode_host_pieces(DeviceCtx *ctx, HostLane *hl, const uint8_t *data, size_t used, size_t count, void *out, uint32_t kind) {
    const size_t esz = kind == ZBRA_ELEM_I64 ? 8 : 4;
    const size_t per = HOST_PIECE_BYTES / esz;  // values per piece: a multiple of every tile size
    const size_t np = (count + per - 1) / per;
    int64_t mid;
    uint32_t pl;
    std::memcpy(&mid, data, 8);
    std::memcpy(&pl, data + 8, 4);
    const uint32_t w = data[12];
    const uint64_t avail = (uint64_t)pl - 1;
    // first byte of E that piece c needs (piece 0 also carries the header) and one past its last
    auto first_byte = [&](size_t c) -> size_t {
        if (c == 0) return 0;
        const uint64_t v0 = (uint64_t)c * per;
        const uint64_t o = w == 0 ? 0 : w < 32 ? v0 * w / 8 : v0 * 8;
        return (size_t)(13 + (o < avail ? o : avail));
    };
    auto end_byte = [&](size_t c) -> size_t {
        const uint64_t v1 = (uint64_t)(c + 1) * per < count ? (uint64_t)(c + 1) * per : count;
        const uint64_t o = w == 0 ? 0 : w < 32 ? (v1 * w + 7) / 8 : v1 * 8;
        return (size_t)(13 + (o < avail ? o : avail));
    };
    if (!hl->stream2) ZBRA_HIP_TRY(hipStreamCreateWithFlags(&hl->stream2, hipStreamNonBlocking));
    for (hipEvent_t &e : hl->piece_done)
        if (!e) ZBRA_HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    // plans: 128 B per piece = DecodePlan at +0, tile_start {0, ntiles} at +64
    Slot *s;
    ZBRA_TRY(ctx->acquire(np * 128, &s));
    for (size_t c = 0; c < np; c++) {
        const uint64_t v0 = (uint64_t)c * per, nv = count - v0 < per ? count - v0 : per;
        const uint64_t o = w == 0 ? 0 : w < 32 ? v0 * w / 8 : v0 * 8;
        DecodePlan p;
        // A raw piece that starts past the end of a short payload has no byte to read: its pointer
        // stays at the last byte that was uploaded (dev_in holds used + 16 bytes, no more), so the
        // aligned 16-byte chunk a lane may touch before it zeroes the value lies inside dev_in.
        const uint64_t oc = o < avail ? o : avail;
        p.payload = (const uint8_t *)hl->dev_in.ptr + 13 + oc;
        p.out = (char *)hl->dev_out.ptr + v0 * esz;
        p.count = nv; p.midpoint = mid; p.avail = avail - oc; p.width = w; p.kind = kind;
        std::memcpy((char *)s->host + c * 128, &p, sizeof(p));
        const uint64_t ts[2] = {0, decode_tiles_of(nv, kind)};
        std::memcpy((char *)s->host + c * 128 + 64, ts, sizeof(ts));
    }
    hipStream_t up = hl->stream, down = hl->stream2;
    size_t submitted = 0;  // pieces uploaded and launched so far
    auto submit = [&]() -> int {
        const size_t c = submitted;
        const size_t b0 = first_byte(c), b1 = end_byte(c);
        if (b1 > b0) ZBRA_TRY(hl->pipe_up.to_device((char *)hl->dev_in.ptr + b0, data + b0, b1 - b0, up));
        const uint64_t v0 = (uint64_t)c * per, nv = count - v0 < per ? count - v0 : per;
        launch_decode(up, (const DecodePlan *)((char *)s->dev + c * 128), (const uint64_t *)((char *)s->dev + c * 128 + 64), nullptr, 1u,
                      decode_tiles_of(nv, kind), w >= 32 ? 0u : w >= 16 ? 16u : w >= 10 ? 1u : 0u);
        ZBRA_HIP_TRY(hipGetLastError());
        ZBRA_HIP_TRY(hipEventRecord(hl->piece_done[c % 4], up));
        submitted++;
        return ZBRA_OK;
    };
    int st = hip_status(hipMemcpyAsync(s->dev, s->host, np * 128, hipMemcpyHostToDevice, up));
    size_t waited = 0;  // pieces the way back has been told to wait for
    if (st == ZBRA_OK)
        st = hl->pipe.to_host(out, hl->dev_out.ptr, count * esz, down, [&](size_t last_byte) -> int {
            // the chunk ending at last_byte needs pieces up to this one; keep one more in flight
            const size_t need = (last_byte + HOST_PIECE_BYTES - 1) / HOST_PIECE_BYTES;
            while (submitted < np && submitted < need + 1) ZBRA_TRY(submit());
            for (; waited < need; waited++) ZBRA_HIP_TRY(hipStreamWaitEvent(down, hl->piece_done[waited % 4], 0));
            return ZBRA_OK;
        });
    // nothing of this call may still run when the arenas are handed to the next one
    const int st_up = hip_status(hipStreamSynchronize(up)), st_down = hip_status(hipStreamSynchronize(down));
    const int st_rel = ctx->release(s, up);
    return st != ZBRA_OK ? st : st_up != ZBRA_OK ? st_up : st_down != ZBRA_OK ? st_down : st_rel;
}

static int decode_host(const uint8_t *data, size_t len, size_t count, void *out, uint32_t kind, size_t *out_count) {
    if (!out_count || (len && !data)) return ZBRA_ERR_INVALID_ARG;
    if (count > MAX_COUNT) return ZBRA_ERR_INVALID_ARG;
    size_t produced = 0;
    ZBRA_TRY(host_check_decode(data, len, count, &produced));
    *out_count = produced;
    if (produced == 0) return ZBRA_OK;
    if (!out) return ZBRA_ERR_INVALID_ARG;
    DeviceCtx *ctx;
    ZBRA_TRY(current_ctx(&ctx));
    HostLaneLock lane_lock(ctx);
    HostLane *hl = lane_lock.lane;
    ZBRA_TRY(hl->ensure_stream());
    const size_t esz = kind == ZBRA_ELEM_I64 ? 8 : 4;
    uint32_t packed_len;
    std::memcpy(&packed_len, data + 8, 4);               // E sits at any byte address
    const size_t used = 12 + (size_t)packed_len;         // bytes after 12 + packed_len are ignored
    ZBRA_TRY(DeviceCtx::grow_dev(hl->dev_in, used + 16));
    ZBRA_TRY(DeviceCtx::grow_dev(hl->dev_out, count * esz));
    if (count * esz >= 2 * HOST_PIECE_BYTES && HostPipe::enabled()) return decode_host_pieces(ctx, hl, data, used, count, out, kind);
    ZBRA_TRY(hl->pipe.to_device(hl->dev_in.ptr, data, used, hl->stream));
    zbra_array_desc d;
    d.data = (const uint8_t *)hl->dev_in.ptr; d.len = used; d.count = count; d.out = hl->dev_out.ptr; d.kind = kind; d.reserved = 0;
    ZBRA_TRY(decode_batch_dev(ctx, &d, 1, nullptr, hl->stream));
    ZBRA_TRY(hl->pipe.to_host(out, hl->dev_out.ptr, count * esz, hl->stream));
    return ZBRA_OK;
}

static int encode_host(const void *values, size_t n, uint32_t kind, uint8_t *out, size_t cap, size_t *out_len) {
    if (!out_len || (n && !values)) return ZBRA_ERR_INVALID_ARG;
    *out_len = 0;
    if (n == 0) return ZBRA_OK;  // compression.rs:272-274
    if (n > MAX_COUNT) return ZBRA_ERR_INVALID_ARG;
    DeviceCtx *ctx;
    ZBRA_TRY(current_ctx(&ctx));
    HostLaneLock lane_lock(ctx);
    HostLane *hl = lane_lock.lane;
    ZBRA_TRY(hl->ensure_stream());
    const size_t esz = kind == ZBRA_ELEM_I64 ? 8 : 4;
    const size_t bound = zbra_hip_encode_bound(n);
    ZBRA_TRY(DeviceCtx::grow_dev(hl->dev_in, n * esz));
    ZBRA_TRY(DeviceCtx::grow_dev(hl->dev_out, bound + 16));
    ZBRA_TRY(DeviceCtx::grow_dev(hl->dev_tmp, sizeof(zbra_dev_result)));
    ZBRA_TRY(hl->pipe.to_device(hl->dev_in.ptr, values, n * esz, hl->stream));
    EncodeItem it{hl->dev_in.ptr, n, kind, (uint8_t *)hl->dev_out.ptr, bound};
    ZBRA_TRY(encode_batch_dev(ctx, &it, 1, (zbra_dev_result *)hl->dev_tmp.ptr, hl->stream));
    zbra_dev_result r;
    ZBRA_HIP_TRY(hipMemcpyAsync(&r, hl->dev_tmp.ptr, sizeof(r), hipMemcpyDeviceToHost, hl->stream));
    ZBRA_HIP_TRY(hipStreamSynchronize(hl->stream));
    if (r.status != ZBRA_OK) return (int)r.status;
    if (r.out_count > cap || !out) return ZBRA_ERR_CAPACITY;
    ZBRA_TRY(hl->pipe.to_host(out, hl->dev_out.ptr, r.out_count, hl->stream));
    *out_len = r.out_count;
    return ZBRA_OK;
}

// Runs an elementwise stage kernel over host buffers.
template <typename Launch>
static int stage_host(const void *in, size_t in_bytes, void *out, size_t out_bytes, Launch &&launch) {
    if ((in_bytes && !in) || (out_bytes && !out)) return ZBRA_ERR_INVALID_ARG;
    if (in_bytes == 0) return ZBRA_OK;
    DeviceCtx *ctx;
    ZBRA_TRY(current_ctx(&ctx));
    HostLaneLock lane_lock(ctx);
    HostLane *hl = lane_lock.lane;
    ZBRA_TRY(hl->ensure_stream());
    ZBRA_TRY(DeviceCtx::grow_dev(hl->dev_in, in_bytes));
    ZBRA_TRY(DeviceCtx::grow_dev(hl->dev_out, out_bytes));
    ZBRA_TRY(hl->pipe.to_device(hl->dev_in.ptr, in, in_bytes, hl->stream));
    launch(hl->dev_in.ptr, hl->dev_out.ptr, hl->stream);
    ZBRA_HIP_TRY(hipGetLastError());
    ZBRA_TRY(hl->pipe.to_host(out, hl->dev_out.ptr, out_bytes, hl->stream));
    return ZBRA_OK;
}

// ---- many arrays, host memory both ends -----------------------------------------------------

// items[i]: E in HOST memory (data, len), count, kind, and the HOST buffer the values go to.  The
// decoded values of up to MAX_OUT bytes of arrays ("a round") share one device buffer that a
// single pipelined copy takes back to the callers' buffers.  Inside a round, groups of consecutive
// arrays (bounded encoded bytes) are gathered up and batch-decoded on two alternating lanes just
// ahead of that copy: upload, kernels and the way back overlap, and the copy never drains between
// groups.
static int decode_host_batch(const zbra_array_desc *items, size_t n, int *statuses) {
    for (size_t i = 0; i < n; i++) {
        if (items[i].count > MAX_COUNT || items[i].kind > ZBRA_ELEM_U32) return ZBRA_ERR_INVALID_ARG;
        if ((items[i].len && !items[i].data) || (items[i].count && items[i].len && !items[i].out)) return ZBRA_ERR_INVALID_ARG;
    }
    DeviceCtx *ctx;
    ZBRA_TRY(current_ctx(&ctx));
    HostLaneLock lane_lock(ctx);
    HostLane *hl = lane_lock.lane;
    ZBRA_TRY(hl->ensure_stream());
    ZBRA_TRY(hl->ensure_file_lanes());
    if (!hl->stream2) ZBRA_HIP_TRY(hipStreamCreateWithFlags(&hl->stream2, hipStreamNonBlocking));
    for (hipEvent_t &e : hl->piece_done)
        if (!e) ZBRA_HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    constexpr size_t MAX_IN = (size_t)16 << 20, MAX_OUT = (size_t)4 << 30, MAX_ARRAYS = 4096;
    struct Group { size_t first, end, in_bytes, out_off, out_end; };
    // status words come back into pinned memory: a copy into pageable memory would block the
    // host until the group's kernels have run, and the host has the copy back to keep fed
    ZBRA_TRY(DeviceCtx::grow_pinned(hl->pin_res, n * sizeof(zbra_dev_result)));
    zbra_dev_result *results = (zbra_dev_result *)hl->pin_res.ptr;
    std::vector<zbra_array_desc> descs;
    std::vector<HostPipe::Segment> segs, up;
    std::vector<Group> groups;
    std::vector<size_t> out_off(n);
    auto padded_out = [&](size_t k) { return (size_t)((items[k].count * (items[k].kind == ZBRA_ELEM_I64 ? 8 : 4) + 15) / 16 * 16); };
    auto padded_in = [&](size_t k) { return (size_t)((items[k].len + 15) / 16 * 16); };
    int st = ZBRA_OK;
    for (size_t r0 = 0; r0 < n && st == ZBRA_OK;) {
        // ---- one round: arrays [r0, r1) ----
        size_t r1 = r0, total = 0;
        groups.clear();
        while (r1 < n && (r1 == r0 || total + padded_out(r1) <= MAX_OUT)) {
            Group gr{r1, r1, 0, total, total};
            while (r1 < n && r1 - gr.first < MAX_ARRAYS) {
                const size_t ib = padded_in(r1), ob = padded_out(r1);
                if (r1 > gr.first && (gr.in_bytes + ib > MAX_IN || total + ob > MAX_OUT)) break;
                out_off[r1] = total;
                total += ob;
                gr.in_bytes += ib;
                r1++;
            }
            gr.end = r1;
            gr.out_end = total;
            groups.push_back(gr);
        }
        ZBRA_TRY(DeviceCtx::grow_dev(hl->dev_out, total + 16));
        // what goes back: exactly the values each array produces, known from its header, which the
        // host can read (nothing for empty / rejected arrays)
        segs.clear();
        for (size_t k = r0; k < r1; k++) {
            size_t produced = 0;
            if (host_check_decode(items[k].data, items[k].len, items[k].count, &produced) == ZBRA_OK && produced)
                segs.push_back({out_off[k], items[k].out, produced * (items[k].kind == ZBRA_ELEM_I64 ? 8 : 4)});
        }
        size_t submitted = 0, waited = 0;
        auto submit = [&]() -> int {  // upload + decode of the next group, on its lane
            const Group &gr = groups[submitted];
            FileLane &lane = hl->file_lanes[submitted & 1];
            const size_t ng = gr.end - gr.first;
            if (lane.img.cap < gr.in_bytes + 16 || lane.res.cap < ng * sizeof(zbra_dev_result)) {
                ZBRA_HIP_TRY(hipStreamSynchronize(lane.stream));  // a kernel may still read the old buffers
                ZBRA_TRY(DeviceCtx::grow_dev(lane.img, gr.in_bytes + 16));
                ZBRA_TRY(DeviceCtx::grow_dev(lane.res, ng * sizeof(zbra_dev_result)));
            }
            descs.resize(ng);
            up.clear();
            size_t o = 0;
            for (size_t k = 0; k < ng; k++) {
                const zbra_array_desc &it = items[gr.first + k];
                if (it.len) up.push_back({o, const_cast<uint8_t *>(it.data), (size_t)it.len});
                descs[k].data = (const uint8_t *)lane.img.ptr + o;
                descs[k].len = it.len; descs[k].count = it.count; descs[k].out = (char *)hl->dev_out.ptr + out_off[gr.first + k];
                descs[k].kind = it.kind; descs[k].reserved = 0;
                o += padded_in(gr.first + k);
            }
            ZBRA_TRY(hl->pipe_up.to_device_segments(up.data(), up.size(), lane.img.ptr, gr.in_bytes, lane.stream));
            ZBRA_TRY(decode_batch_dev(ctx, descs.data(), ng, (zbra_dev_result *)lane.res.ptr, lane.stream));
            ZBRA_HIP_TRY(hipMemcpyAsync(&results[gr.first], lane.res.ptr, ng * sizeof(zbra_dev_result), hipMemcpyDeviceToHost, lane.stream));
            ZBRA_HIP_TRY(hipEventRecord(hl->piece_done[submitted % 4], lane.stream));
            submitted++;
            return ZBRA_OK;
        };
        st = hl->pipe.to_host_segments(segs.data(), segs.size(), hl->dev_out.ptr, total, hl->stream2, [&](size_t last_byte) -> int {
            // groups that produce bytes below last_byte must have been enqueued; keep one more in flight
            size_t need = waited;
            while (need < groups.size() && groups[need].out_off < last_byte) need++;
            while (submitted < groups.size() && submitted < need + 1) ZBRA_TRY(submit());
            for (; waited < need; waited++) ZBRA_HIP_TRY(hipStreamWaitEvent(hl->stream2, hl->piece_done[waited % 4], 0));
            return ZBRA_OK;
        });
        // groups whose arrays produce nothing are still decoded for their status words
        while (st == ZBRA_OK && submitted < groups.size()) st = submit();
        const int st0 = hip_status(hipStreamSynchronize(hl->file_lanes[0].stream));
        const int st1 = hip_status(hipStreamSynchronize(hl->file_lanes[1].stream));
        const int st2 = hip_status(hipStreamSynchronize(hl->stream2));
        if (st == ZBRA_OK) st = st0 != ZBRA_OK ? st0 : st1 != ZBRA_OK ? st1 : st2;
        r0 = r1;
    }
    if (st != ZBRA_OK) return st;
    int first_err = ZBRA_OK;
    for (size_t k = 0; k < n; k++) {
        if (statuses) statuses[k] = (int)results[k].status;
        if (results[k].status != ZBRA_OK && first_err == ZBRA_OK) first_err = (int)results[k].status;
    }
    return first_err;
}

}  // namespace zbra

using namespace zbra;

extern "C" {

// Test and tuning hooks, declared in include/zbra_hip_tuning.h (not part of the product ABI).  Each
// publishes a new policy snapshot under the writers' mutex; calls already running keep the old one.
int zbra_tuning_set_decode_warm(uint64_t chunk_bytes, uint32_t max_bits, uint64_t min_bytes) {
    retune([&](Tuning &t) { t.warm_bytes = chunk_bytes; t.warm_bits = max_bits; t.warm_min_bytes = min_bytes; return true; });
    return 0;
}
int zbra_tuning_set_enum_reverse(int r) {
    retune([&](Tuning &t) { t.enum_reverse = r; return true; });
    return 0;
}
int zbra_tuning_set_launch_limit(uint64_t work_items) {
    retune([&](Tuning &t) { t.launch_limit = work_items && work_items < 0xFFFFFFFFull ? work_items : 0xFFFFFFFFull; return true; });
    return 0;
}
int zbra_tuning_set_decode_variant(int tile, int threads, int nt, int direct, int xcd) {
    return retune([&](Tuning &t) { t.decode.xcd = xcd; return apply_variant(t, tile, threads, nt, direct); }) ? 0 : 1;
}

const char *zbra_hip_version(void) { return "zbra-hip 0.1 (gfx950)"; }

const char *zbra_hip_strerror(int status) {
    switch (status) {
        case ZBRA_OK: return "ok";
        case ZBRA_ERR_SHORT_HEADER: return "Invalid compressed data length";       // compression.rs:302
        case ZBRA_ERR_SHORT_PACKED: return "Insufficient data for packed array";   // compression.rs:314
        case ZBRA_ERR_TRUNCATED: return "Packed payload shorter than count * width bits";
        case ZBRA_ERR_CAPACITY: return "Output buffer too small";
        case ZBRA_ERR_INVALID_ARG: return "Invalid argument";
        case ZBRA_ERR_NO_DEVICE: return "No usable HIP device";
        case ZBRA_ERR_FORMAT: return "Invalid .zbra image (magic number, tag or length)";
        case ZBRA_ERR_UNKNOWN_TAG: return "enum tag carried by no variant";                // striped.rs:352-358
        case ZBRA_ERR_NOALLOC_LIMIT: return "Submission exceeds the fixed scratch of the no-allocation entry point (arrays or tiles)";
        default:
            if (status >= ZBRA_ERR_HIP_BASE) return hipGetErrorString((hipError_t)(status - ZBRA_ERR_HIP_BASE));
            return "unknown status";
    }
}

int zbra_hip_device_count(int *count) {
    if (!count) return ZBRA_ERR_INVALID_ARG;
    *count = 0;
    hipError_t e = hipGetDeviceCount(count);
    if (e != hipSuccess) { (void)hipGetLastError(); *count = 0; return ZBRA_ERR_NO_DEVICE; }
    return ZBRA_OK;
}

int zbra_hip_set_device(int device) { return hip_status(hipSetDevice(device)); }

int zbra_hip_init(void) {
    DeviceCtx *ctx;
    return current_ctx(&ctx);
}

size_t zbra_hip_encode_bound(size_t n) { return n == 0 ? 0 : 13 + 8 * n; }

int zbra_hip_decoded_count(const uint8_t *data, size_t len, size_t count, size_t *out_count) {
    if (!out_count || (len && !data) || count > MAX_COUNT) return ZBRA_ERR_INVALID_ARG;
    return host_check_decode(data, len, count, out_count);
}

int zbra_hip_encode_i64(const int64_t *values, size_t n, uint8_t *out, size_t cap, size_t *out_len) {
    return encode_host(values, n, ZBRA_ELEM_I64, out, cap, out_len);
}
int zbra_hip_encode_u32(const uint32_t *values, size_t n, uint8_t *out, size_t cap, size_t *out_len) {
    return encode_host(values, n, ZBRA_ELEM_U32, out, cap, out_len);
}
int zbra_hip_decode_i64(const uint8_t *data, size_t len, size_t count, int64_t *out, size_t *out_count) {
    return decode_host(data, len, count, out, ZBRA_ELEM_I64, out_count);
}
int zbra_hip_decode_u32(const uint8_t *data, size_t len, size_t count, uint32_t *out, size_t *out_count) {
    return decode_host(data, len, count, out, ZBRA_ELEM_U32, out_count);
}

int zbra_hip_frame_of_reference_encode(const int64_t *values, size_t n, int64_t *midpoint, int64_t *deltas) {
    if (!midpoint || (n && (!values || !deltas))) return ZBRA_ERR_INVALID_ARG;
    *midpoint = 0;
    if (n == 0) return ZBRA_OK;  // compression.rs:53-55
    if (n > MAX_COUNT) return ZBRA_ERR_INVALID_ARG;
    DeviceCtx *ctx;
    ZBRA_TRY(current_ctx(&ctx));
    HostLaneLock lane_lock(ctx);
    HostLane *hl = lane_lock.lane;
    ZBRA_TRY(hl->ensure_stream());
    ZBRA_TRY(DeviceCtx::grow_dev(hl->dev_in, n * 8));
    ZBRA_TRY(DeviceCtx::grow_dev(hl->dev_out, n * 8));
    ZBRA_TRY(DeviceCtx::grow_dev(hl->dev_tmp, sizeof(zbra_dev_result)));
    ZBRA_TRY(hl->pipe.to_device(hl->dev_in.ptr, values, n * 8, hl->stream));
    EncodeItem it{hl->dev_in.ptr, n, ZBRA_ELEM_I64, nullptr, 0};
    ZBRA_TRY(encode_batch_dev(ctx, &it, 1, (zbra_dev_result *)hl->dev_tmp.ptr, hl->stream, ENCODE_MEDIAN_ONLY));
    const unsigned grid = grid_for((n + THREADS * 2 - 1) / (THREADS * 2), STREAM_GRID_CAP);
    hipLaunchKernelGGL(for_sub_kernel, dim3(grid), dim3(THREADS), 0, hl->stream, (const int64_t *)hl->dev_in.ptr,
                       (uint64_t)n, (const zbra_dev_result *)hl->dev_tmp.ptr, (int64_t *)hl->dev_out.ptr);
    ZBRA_HIP_TRY(hipGetLastError());
    zbra_dev_result r;
    ZBRA_HIP_TRY(hipMemcpyAsync(&r, hl->dev_tmp.ptr, sizeof(r), hipMemcpyDeviceToHost, hl->stream));
    ZBRA_TRY(hl->pipe.to_host(deltas, hl->dev_out.ptr, n * 8, hl->stream));
    *midpoint = r.midpoint;
    return ZBRA_OK;
}

int zbra_hip_frame_of_reference_decode(int64_t midpoint, const int64_t *deltas, size_t n, int64_t *out) {
    return stage_host(deltas, n * 8, out, n * 8, [&](void *din, void *dout, hipStream_t st) {
        const unsigned grid = grid_for((n + THREADS * 2 - 1) / (THREADS * 2), STREAM_GRID_CAP);
        hipLaunchKernelGGL(for_add_kernel, dim3(grid), dim3(THREADS), 0, st, (const int64_t *)din, (uint64_t)n, midpoint,
                           (int64_t *)dout);
    });
}

int zbra_hip_zig_zag_encode(const int64_t *values, size_t n, uint64_t *out) {
    return stage_host(values, n * 8, out, n * 8, [&](void *din, void *dout, hipStream_t st) {
        const unsigned grid = grid_for((n + THREADS * 2 - 1) / (THREADS * 2), STREAM_GRID_CAP);
        hipLaunchKernelGGL(zigzag_encode_kernel, dim3(grid), dim3(THREADS), 0, st, (const int64_t *)din, (uint64_t)n,
                           (uint64_t *)dout);
    });
}

int zbra_hip_zig_zag_decode(const uint64_t *values, size_t n, int64_t *out) {
    return stage_host(values, n * 8, out, n * 8, [&](void *din, void *dout, hipStream_t st) {
        const unsigned grid = grid_for((n + THREADS * 2 - 1) / (THREADS * 2), STREAM_GRID_CAP);
        hipLaunchKernelGGL(zigzag_decode_kernel, dim3(grid), dim3(THREADS), 0, st, (const uint64_t *)din, (uint64_t)n,
                           (int64_t *)dout);
    });
}

int zbra_hip_bp64_pack(const uint64_t *values, size_t n, uint8_t *out, size_t cap, size_t *out_len) {
    if (!out_len || (n && !values)) return ZBRA_ERR_INVALID_ARG;
    *out_len = 0;
    if (n == 0) return ZBRA_OK;  // compression.rs:118-120
    if (n > MAX_COUNT) return ZBRA_ERR_INVALID_ARG;
    DeviceCtx *ctx;
    ZBRA_TRY(current_ctx(&ctx));
    HostLaneLock lane_lock(ctx);
    HostLane *hl = lane_lock.lane;
    ZBRA_TRY(hl->ensure_stream());
    const size_t bound = 1 + 8 * n;
    ZBRA_TRY(DeviceCtx::grow_dev(hl->dev_in, n * 8));
    ZBRA_TRY(DeviceCtx::grow_dev(hl->dev_out, 12 + bound + 32));
    ZBRA_TRY(DeviceCtx::grow_dev(hl->dev_tmp, sizeof(zbra_dev_result)));
    ZBRA_TRY(hl->pipe.to_device(hl->dev_in.ptr, values, n * 8, hl->stream));
    // The stream is built by the encode pipeline with values taken as they are; its E layout puts
    // the width byte at +12, so `packed` is E without the 12 header bytes.
    EncodeItem it{hl->dev_in.ptr, n, KIND_U64_PLAIN, (uint8_t *)hl->dev_out.ptr, 12 + bound};
    ZBRA_TRY(encode_batch_dev(ctx, &it, 1, (zbra_dev_result *)hl->dev_tmp.ptr, hl->stream));
    zbra_dev_result r;
    ZBRA_HIP_TRY(hipMemcpyAsync(&r, hl->dev_tmp.ptr, sizeof(r), hipMemcpyDeviceToHost, hl->stream));
    ZBRA_HIP_TRY(hipStreamSynchronize(hl->stream));
    if (r.status != ZBRA_OK) return (int)r.status;
    const size_t len = (size_t)r.packed_len;
    if (len > cap || !out) return ZBRA_ERR_CAPACITY;
    ZBRA_TRY(hl->pipe.to_host(out, (uint8_t *)hl->dev_out.ptr + 12, len, hl->stream));
    *out_len = len;
    return ZBRA_OK;
}

int zbra_hip_bp64_unpack(const uint8_t *packed, size_t packed_len, size_t count, uint64_t *out, size_t *out_count) {
    if (!out_count || (packed_len && !packed)) return ZBRA_ERR_INVALID_ARG;
    *out_count = 0;
    if (packed_len == 0) return ZBRA_OK;  // compression.rs:182-184
    if (count > MAX_COUNT) return ZBRA_ERR_INVALID_ARG;
    if (count == 0) return ZBRA_OK;
    if (!out) return ZBRA_ERR_INVALID_ARG;
    const uint32_t w = packed[0];
    if (w >= 1 && w < 32 && (uint64_t)(packed_len - 1) < ((uint64_t)count * w + 7) / 8) return ZBRA_ERR_TRUNCATED;
    DeviceCtx *ctx;
    ZBRA_TRY(current_ctx(&ctx));
    HostLaneLock lane_lock(ctx);
    HostLane *hl = lane_lock.lane;
    ZBRA_TRY(hl->ensure_stream());
    ZBRA_TRY(DeviceCtx::grow_dev(hl->dev_in, packed_len + 32));
    ZBRA_TRY(DeviceCtx::grow_dev(hl->dev_out, count * 8));
    ZBRA_TRY(hl->pipe.to_device(hl->dev_in.ptr, packed, packed_len, hl->stream));
    ZBRA_TRY(DeviceCtx::grow_dev(hl->dev_tmp, 256));
    const uint64_t ntiles = decode_tiles_of(count, KIND_U64_PLAIN);
    DecodePlan *plan = (DecodePlan *)hl->dev_tmp.ptr;
    uint64_t *tile_start = (uint64_t *)((char *)hl->dev_tmp.ptr + 64);
    hipLaunchKernelGGL(plan_unpack_kernel, dim3(1), dim3(64), 0, hl->stream, (const uint8_t *)hl->dev_in.ptr,
                       (uint64_t)packed_len, (uint64_t)count, hl->dev_out.ptr, ntiles, plan, tile_start);
    launch_decode(hl->stream, plan, tile_start, nullptr, 1u, ntiles, w >= 32 ? 0u : w >= 16 ? 16u : w >= 10 ? 1u : 0u);
    ZBRA_HIP_TRY(hipGetLastError());
    ZBRA_TRY(hl->pipe.to_host(out, hl->dev_out.ptr, count * 8, hl->stream));
    *out_count = count;
    return ZBRA_OK;
}

int zbra_hip_decode_dev(const uint8_t *d_data, size_t len, size_t count, void *d_out, uint32_t kind,
                        zbra_dev_result *d_result, void *stream) {
    DeviceCtx *ctx;
    ZBRA_TRY(current_ctx(&ctx));
    zbra_array_desc d;
    d.data = d_data; d.len = len; d.count = count; d.out = d_out; d.kind = kind; d.reserved = 0;
    return decode_batch_dev(ctx, &d, 1, d_result, (hipStream_t)stream);
}

int zbra_hip_decode_offsets_dev(const uint8_t *d_data, size_t len, size_t count, uint64_t *d_offsets, zbra_dev_result *d_result,
                                void *stream_) {
    if (!d_offsets || (len && !d_data) || count > MAX_COUNT) return ZBRA_ERR_INVALID_ARG;
    zbra_array_desc d;
    d.data = d_data; d.len = len; d.count = count; d.out = d_offsets; d.kind = ZBRA_ELEM_I64; d.reserved = 0;
    DeviceCtx *ctx;
    ZBRA_TRY(current_ctx(&ctx));
    return decode_offsets_batch_dev(ctx, &d, 1, d_result, (hipStream_t)stream_);
}

int zbra_hip_decode_offsets_batch_dev(const zbra_array_desc *descs, size_t nd, zbra_dev_result *d_results, void *stream) {
    DeviceCtx *ctx;
    ZBRA_TRY(current_ctx(&ctx));
    return decode_offsets_batch_dev(ctx, descs, nd, d_results, (hipStream_t)stream);
}

int zbra_hip_enum_rows_dev(const uint32_t *d_tags, size_t n, const uint32_t *variant_tags, uint32_t n_variants,
                           uint32_t *d_variant, uint64_t *d_rank, uint64_t *d_counts, zbra_dev_result *d_result, void *stream_) {
    zbra_enum_desc d;
    d.tags = d_tags; d.n = n; d.variant = d_variant; d.rank = d_rank; d.counts = d_counts;
    DeviceCtx *ctx;
    ZBRA_TRY(current_ctx(&ctx));
    return enum_rows_batch_dev(ctx, &d, 1, variant_tags, n_variants, d_result, (hipStream_t)stream_);
}

int zbra_hip_enum_rows_batch_dev(const zbra_enum_desc *descs, size_t nd, const uint32_t *variant_tags, uint32_t n_variants,
                                 zbra_dev_result *d_results, void *stream) {
    DeviceCtx *ctx;
    ZBRA_TRY(current_ctx(&ctx));
    return enum_rows_batch_dev(ctx, descs, nd, variant_tags, n_variants, d_results, (hipStream_t)stream);
}

int zbra_hip_decode_batch_dev(const zbra_array_desc *descs, size_t nd, zbra_dev_result *d_results, void *stream) {
    DeviceCtx *ctx;
    ZBRA_TRY(current_ctx(&ctx));
    return decode_batch_dev(ctx, descs, nd, d_results, (hipStream_t)stream);
}

int zbra_hip_decode_batch_dev_noalloc(const zbra_array_desc *descs, size_t nd, zbra_dev_result *d_results, void *stream) {
    // The context (and with it this device's scratch address) exists after zbra_hip_init() or any
    // earlier call on this device; only when this is the very first call does it run the set-up.
    DeviceCtx *ctx;
    ZBRA_TRY(current_ctx(&ctx));
    return decode_batch_dev_noalloc(ctx, descs, nd, d_results, (hipStream_t)stream);
}

int zbra_hip_encode_dev(const void *d_values, size_t n, uint32_t kind, uint8_t *d_out, size_t cap,
                        zbra_dev_result *d_result, void *stream) {
    if (!d_result) return ZBRA_ERR_INVALID_ARG;
    DeviceCtx *ctx;
    ZBRA_TRY(current_ctx(&ctx));
    EncodeItem it{d_values, n, kind, d_out, cap};
    return encode_batch_dev(ctx, &it, 1, d_result, (hipStream_t)stream);
}

int zbra_hip_encode_batch_dev(const zbra_encode_desc *descs, size_t nd, zbra_dev_result *d_results, void *stream) {
    DeviceCtx *ctx;
    ZBRA_TRY(current_ctx(&ctx));
    if (nd && !descs) return ZBRA_ERR_INVALID_ARG;
    std::vector<EncodeItem> items(nd);
    for (size_t i = 0; i < nd; i++) items[i] = EncodeItem{descs[i].values, descs[i].n, descs[i].kind, descs[i].out, descs[i].cap};
    return encode_batch_dev(ctx, items.data(), nd, d_results, (hipStream_t)stream);
}

int zbra_hip_encode_frames_dev(const zbra_encode_desc *descs, size_t nd, uint8_t *d_out, size_t cap, uint64_t *d_offsets,
                               zbra_dev_result *d_results, void *stream) {
    DeviceCtx *ctx;
    ZBRA_TRY(current_ctx(&ctx));
    if (!d_offsets || (nd && !descs)) return ZBRA_ERR_INVALID_ARG;
    if (nd == 0) return hip_status(hipMemsetAsync(d_offsets, 0, sizeof(uint64_t), (hipStream_t)stream));
    std::vector<EncodeItem> items(nd);
    for (size_t i = 0; i < nd; i++) items[i] = EncodeItem{descs[i].values, descs[i].n, descs[i].kind, nullptr, 0};
    FrameTarget ft;
    ft.out = d_out; ft.cap = cap; ft.offsets = d_offsets;
    return encode_batch_dev(ctx, items.data(), nd, d_results, (hipStream_t)stream, ENCODE_FULL, &ft);
}

int zbra_hip_encode_frames(const zbra_encode_desc *descs, size_t nd, uint8_t *out, size_t cap, uint64_t *offsets,
                           int *statuses) {
    if (!offsets || (nd && !descs) || (cap && !out)) return ZBRA_ERR_INVALID_ARG;
    offsets[0] = 0;
    if (nd == 0) return ZBRA_OK;
    for (size_t i = 0; i < nd; i++) {
        if (descs[i].n > MAX_COUNT || descs[i].kind > ZBRA_ELEM_U32 || (descs[i].n && !descs[i].values)) return ZBRA_ERR_INVALID_ARG;
    }
    DeviceCtx *ctx;
    ZBRA_TRY(current_ctx(&ctx));
    HostLaneLock lane_lock(ctx);
    HostLane *hl = lane_lock.lane;
    ZBRA_TRY(hl->ensure_stream());
    ZBRA_TRY(hl->ensure_enc_lanes());
    // Rounds of consecutive arrays: bounded input bytes and array count per submission.  Rounds
    // alternate between two lanes (stream + device buffers).  This thread produces: it uploads a
    // round's values (pageable memory -> pinned ring -> device) and enqueues the batched encode with
    // device-side framing.  A second thread consumes, round by round in order: it waits for the
    // round's result words, copies the frames back behind the previous round's and fills in offsets
    // and statuses.  Round r + 1 therefore goes up while round r's frames come down: both directions
    // of the link work at once (one round after the other, 164 GB of values took 4.28 s where the
    // upload alone needs 2.9 s).  A job of one round runs both halves on this thread.
    constexpr size_t ROUND_ARRAYS = 4096;
    auto in_bytes = [&](size_t i) { return (size_t)((descs[i].n * (descs[i].kind == ZBRA_ELEM_I64 ? 8 : 4) + 15) / 16 * 16); };
    size_t all_in = 0;
    for (size_t i = 0; i < nd; i++) all_in += in_bytes(i);
    // a quarter of the job per round, so that even a job of a few hundred MiB overlaps its two directions
    const size_t ROUND_IN = all_in / 4 < ((size_t)64 << 20) ? (size_t)64 << 20 : all_in / 4 > ((size_t)512 << 20) ? (size_t)512 << 20 : all_in / 4;
    struct Round { size_t r0, r1, total_in, bound, room, off_res, tmp_bytes; };
    std::vector<Round> rounds;
    for (size_t r0 = 0; r0 < nd;) {
        Round rd{r0, r0, 0, 0, 0, 0, 0};
        while (rd.r1 < nd && rd.r1 - r0 < ROUND_ARRAYS && (rd.r1 == r0 || rd.total_in + in_bytes(rd.r1) <= ROUND_IN)) {
            rd.total_in += in_bytes(rd.r1);
            rd.bound += 8 + zbra_hip_encode_bound(descs[rd.r1].n);
            rd.r1++;
        }
        rd.off_res = align_up((rd.r1 - r0 + 1) * sizeof(uint64_t), 64);
        rd.tmp_bytes = rd.off_res + (rd.r1 - r0) * sizeof(zbra_dev_result);
        rounds.push_back(rd);
        r0 = rd.r1;
    }
    // shared between the two threads, under `mu`
    std::mutex mu;
    std::condition_variable cv;
    size_t written = 0, produced = 0, consumed = 0, outstanding_bound = 0;
    int first_err = ZBRA_OK, failure = ZBRA_OK;  // first_err: an array's status; failure: the call cannot go on
    bool full = false, stop = false;             // full: a frame did not fit

    std::vector<EncodeItem> items;
    std::vector<HostPipe::Segment> segs;
    auto produce = [&](size_t k) -> int {  // upload + encode of round k on its lane (which is free)
        Round &rd = rounds[k];
        EncodeLane &lane = hl->enc_lanes[k & 1];
        const size_t nr = rd.r1 - rd.r0;
        ZBRA_TRY(DeviceCtx::grow_dev(lane.in, rd.total_in + 16));
        ZBRA_TRY(DeviceCtx::grow_dev(lane.out, rd.room + 16));
        ZBRA_TRY(DeviceCtx::grow_dev(lane.tmp, rd.tmp_bytes));
        ZBRA_TRY(DeviceCtx::grow_pinned(lane.pin, rd.tmp_bytes));
        items.resize(nr);
        segs.clear();
        size_t o = 0;
        for (size_t i = 0; i < nr; i++) {  // the round's arrays go up as one stream, each at a 16-B aligned offset
            const zbra_encode_desc &d = descs[rd.r0 + i];
            if (d.n) segs.push_back({o, const_cast<void *>(d.values), (size_t)d.n * (d.kind == ZBRA_ELEM_I64 ? 8 : 4)});
            items[i] = EncodeItem{(char *)lane.in.ptr + o, d.n, d.kind, nullptr, 0};
            o += in_bytes(rd.r0 + i);
        }
        ZBRA_TRY(hl->pipe_up.to_device_segments(segs.data(), segs.size(), lane.in.ptr, rd.total_in, lane.stream));
        FrameTarget ft;
        ft.out = (uint8_t *)lane.out.ptr; ft.cap = rd.room; ft.offsets = (uint64_t *)lane.tmp.ptr;
        ZBRA_TRY(encode_batch_dev(ctx, items.data(), nr, (zbra_dev_result *)((char *)lane.tmp.ptr + rd.off_res), lane.stream, ENCODE_FULL, &ft));
        ZBRA_HIP_TRY(hipMemcpyAsync(lane.pin.ptr, lane.tmp.ptr, rd.tmp_bytes, hipMemcpyDeviceToHost, lane.stream));
        return ZBRA_OK;
    };
    auto consume = [&](size_t k) -> int {  // frames of round k back to the caller, behind round k - 1's
        const Round &rd = rounds[k];
        EncodeLane &lane = hl->enc_lanes[k & 1];
        const size_t nr = rd.r1 - rd.r0;
        ZBRA_HIP_TRY(hipStreamSynchronize(lane.stream));
        const uint64_t *h_off = (const uint64_t *)lane.pin.ptr;
        const zbra_dev_result *h_res = (const zbra_dev_result *)((const char *)lane.pin.ptr + rd.off_res);
        const size_t laid_out = (size_t)h_off[nr];
        size_t at;
        { std::lock_guard<std::mutex> l(mu); at = written; }
        ZBRA_TRY(hl->pipe.to_host(out + at, lane.out.ptr, laid_out, lane.stream));
        int err = ZBRA_OK;
        bool over = false;
        for (size_t i = 0; i < nr; i++) {
            offsets[rd.r0 + i] = at + h_off[i];
            if (statuses) statuses[rd.r0 + i] = (int)h_res[i].status;
            if (h_res[i].status != ZBRA_OK && err == ZBRA_OK) err = (int)h_res[i].status;
            if (h_res[i].status == ZBRA_ERR_CAPACITY) over = true;
        }
        offsets[rd.r1] = at + laid_out;
        std::lock_guard<std::mutex> l(mu);
        written = at + laid_out;
        if (first_err == ZBRA_OK) first_err = err;
        full = full || over;
        return ZBRA_OK;
    };

    const bool threaded = rounds.size() > 1;
    std::thread consumer;
    if (threaded)
        consumer = std::thread([&] {
            if (hipSetDevice(ctx->device) != hipSuccess) {
                std::lock_guard<std::mutex> l(mu);
                failure = ZBRA_ERR_NO_DEVICE; consumed = rounds.size(); cv.notify_all();
                return;
            }
            for (size_t k = 0; k < rounds.size(); k++) {
                {
                    std::unique_lock<std::mutex> l(mu);
                    cv.wait(l, [&] { return produced > k || stop; });
                    if (produced <= k) return;  // stopped before this round was produced
                }
                const int st = consume(k);
                std::lock_guard<std::mutex> l(mu);
                consumed = k + 1;
                outstanding_bound -= rounds[k].bound;
                if (st != ZBRA_OK && failure == ZBRA_OK) failure = st;
                cv.notify_all();
                if (failure != ZBRA_OK) return;
            }
        });
    size_t k = 0;
    int st = ZBRA_OK;
    for (; k < rounds.size() && st == ZBRA_OK; k++) {
        Round &rd = rounds[k];
        {
            std::unique_lock<std::mutex> l(mu);
            // the lane is free once round k - 2 has been consumed
            cv.wait(l, [&] { return failure != ZBRA_OK || k < 2 || consumed + 2 > k; });
            // Room for this round's frames.  While the caller's buffer surely holds every round in
            // flight at its bound plus this one, the round may run ahead; otherwise the rounds before
            // it are waited for, so that what is left is known exactly (and whether a frame already
            // did not fit: frames are laid out in order, nothing is written behind such a frame).
            if (failure == ZBRA_OK && cap - written - (cap - written < outstanding_bound ? cap - written : outstanding_bound) < rd.bound)
                cv.wait(l, [&] { return failure != ZBRA_OK || consumed == k; });
            if (failure != ZBRA_OK) { st = failure; break; }
            if (full) break;
            rd.room = cap - written - outstanding_bound < rd.bound ? cap - written - outstanding_bound : rd.bound;
            outstanding_bound += rd.bound;
        }
        st = produce(k);
        if (st == ZBRA_OK) {
            if (threaded) {
                std::lock_guard<std::mutex> l(mu);
                produced = k + 1;
                cv.notify_all();
            } else {
                st = consume(k);
                produced = consumed = k + 1;
                outstanding_bound -= rd.bound;
            }
        }
    }
    if (threaded) {
        {
            std::lock_guard<std::mutex> l(mu);
            stop = true;  // the consumer finishes what was produced, then leaves
            cv.notify_all();
        }
        consumer.join();
    }
    if (st == ZBRA_OK) st = failure;
    // whatever happened, nothing of this call may still run on the lanes when they are reused
    for (EncodeLane &l : hl->enc_lanes) (void)hipStreamSynchronize(l.stream);
    if (st != ZBRA_OK) return st;
    if (consumed < rounds.size()) {
        // a frame did not fit: every later array is reported as an empty range at the end
        for (size_t i = rounds[consumed].r0; i < nd; i++) {
            offsets[i + 1] = written;
            if (statuses) statuses[i] = ZBRA_ERR_CAPACITY;
        }
    }
    return first_err;
}

int zbra_hip_file_index(const uint8_t *image, size_t len, zbra_file_array *arrays, size_t cap, zbra_file_info *info,
                        size_t *err_offset) {
    if (!info || (len && !image) || (cap && !arrays)) return ZBRA_ERR_INVALID_ARG;
    std::memset(info, 0, sizeof(*info));
    std::vector<zbra_file_array> found;
    Walker w{image, len};
    if (cap) w.arrays = &found;
    if (!w.file(info)) {
        if (err_offset) *err_offset = w.fail_pos;
        return ZBRA_ERR_FORMAT;
    }
    if (cap) {
        if (found.size() > cap) return ZBRA_ERR_CAPACITY;
        std::memcpy(arrays, found.data(), found.size() * sizeof(zbra_file_array));
    }
    return ZBRA_OK;
}

int zbra_hip_file_decode(const uint8_t *image, size_t len, const zbra_file_array *arrays, size_t n, void *const *outs,
                         int *statuses) {
    if (n == 0) return ZBRA_OK;
    if (!image || !arrays || !outs) return ZBRA_ERR_INVALID_ARG;
    std::vector<zbra_array_desc> items(n);
    for (size_t i = 0; i < n; i++) {
        if (arrays[i].offset > len || arrays[i].len > len - arrays[i].offset) return ZBRA_ERR_INVALID_ARG;
        if (i && arrays[i].offset < arrays[i - 1].offset + arrays[i - 1].len) return ZBRA_ERR_INVALID_ARG;  // file order
        items[i].data = image + arrays[i].offset; items[i].len = arrays[i].len; items[i].count = arrays[i].count;
        items[i].out = outs[i]; items[i].kind = arrays[i].kind; items[i].reserved = 0;
    }
    return decode_host_batch(items.data(), n, statuses);
}

int zbra_hip_file_decode_multi(zbra_file_job *jobs, size_t n_jobs) {
    if (n_jobs == 0) return ZBRA_OK;
    if (!jobs) return ZBRA_ERR_INVALID_ARG;
    int devices = 0;
    ZBRA_TRY(zbra_hip_device_count(&devices));
    if (devices == 0) return ZBRA_ERR_NO_DEVICE;
    std::vector<int> used;  // distinct devices, in order of first appearance
    for (size_t i = 0; i < n_jobs; i++) {
        jobs[i].status = jobs[i].device < 0 || jobs[i].device >= devices ? ZBRA_ERR_INVALID_ARG : ZBRA_OK;
        if (jobs[i].status == ZBRA_OK && std::find(used.begin(), used.end(), (int)jobs[i].device) == used.end()) used.push_back(jobs[i].device);
    }
    CopyWorkers::instance().ensure_threads(8 * (int)used.size());
    // one host thread per device: the current device is per thread, and so are the device's
    // context, streams and pinned rings (the calling thread's own current device is left as it is)
    auto run_device = [&](int dev) {
        const int st_dev = zbra_hip_set_device(dev);
        for (size_t i = 0; i < n_jobs; i++) {
            if (jobs[i].device != dev || jobs[i].status != ZBRA_OK) continue;
            jobs[i].status = st_dev != ZBRA_OK ? st_dev
                                               : zbra_hip_file_decode(jobs[i].image, (size_t)jobs[i].len, jobs[i].arrays, (size_t)jobs[i].n, jobs[i].outs,
                                                                      jobs[i].statuses);
        }
    };
    std::vector<std::thread> threads;
    for (int dev : used) threads.emplace_back(run_device, dev);
    for (std::thread &t : threads) t.join();
    for (size_t i = 0; i < n_jobs; i++)
        if (jobs[i].status != ZBRA_OK) return jobs[i].status;
    return ZBRA_OK;
}

int zbra_hip_decode_batch(const zbra_array_desc *descs, size_t nd, int *statuses) {
    if (nd == 0) return ZBRA_OK;
    if (!descs) return ZBRA_ERR_INVALID_ARG;
    return decode_host_batch(descs, nd, statuses);
}

}  // extern "C"
