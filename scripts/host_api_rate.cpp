// Rate of the host-pointer (drop-in) entry points on pageable memory, straight through the C ABI:
// what zbra-core's compress_int_array / decompress_int_array would see after re-pointing.
// Build: g++ -O2 -std=c++17 -Iinclude scripts/host_api_rate.cpp -o build/host_api_rate -Lzbra_amd -lzbra_hip -Wl,-rpath,$PWD/zbra_amd
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <sys/mman.h>

#include "zbra_hip.h"

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char **argv) {
    const int max_lg = argc > 1 ? std::atoi(argv[1]) : 28;
    const int min_lg = argc > 2 ? std::atoi(argv[2]) : 0;
    if (zbra_hip_init() != ZBRA_OK) { std::fprintf(stderr, "no device\n"); return 1; }
    for (int lg : {16, 20, 24, 26, 28}) {
        if (lg > max_lg) break;
        if (lg < min_lg) continue;
        const size_t n = (size_t)1 << lg;
        std::vector<int64_t> v(n);
        uint64_t x = 0x9E3779B97F4A7C15ull * (lg + 1);
        for (size_t i = 0; i < n; i++) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; v[i] = 1700000000000ll + (int64_t)(x % 4000); }
        std::vector<uint8_t> enc(zbra_hip_encode_bound(n));
        size_t elen = 0, got = 0;
        std::vector<int64_t> back(n);
        if (zbra_hip_encode_i64(v.data(), n, enc.data(), enc.size(), &elen) != ZBRA_OK) return 2;
        if (zbra_hip_decode_i64(enc.data(), elen, n, back.data(), &got) != ZBRA_OK || got != n) return 3;
        if (std::memcmp(v.data(), back.data(), n * 8) != 0) { std::fprintf(stderr, "round trip differs\n"); return 4; }
        const int reps = lg >= 28 ? 3 : lg >= 24 ? 8 : 50;
        double t0 = now();
        for (int r = 0; r < reps; r++) zbra_hip_encode_i64(v.data(), n, enc.data(), enc.size(), &elen);
        const double te = (now() - t0) / reps;
        t0 = now();
        for (int r = 0; r < reps; r++) zbra_hip_decode_i64(enc.data(), elen, n, back.data(), &got);
        const double td = (now() - t0) / reps;
        // a destination nobody has touched yet, as a fresh Vec is: page faults included
        double tf = 0;
        for (int r = 0; r < reps; r++) {
            int64_t *fresh = (int64_t *)std::malloc(n * 8);
            t0 = now();
            zbra_hip_decode_i64(enc.data(), elen, n, fresh, &got);
            tf += now() - t0;
            if (fresh[n - 1] != v[n - 1]) return 5;
            std::free(fresh);
        }
        tf /= reps;
        // the same with the fresh destination advised to use huge pages before the call
        double th = 0;
        for (int r = 0; r < reps; r++) {
            int64_t *fresh = (int64_t *)std::malloc(n * 8);
            t0 = now();
            const uintptr_t lo = ((uintptr_t)fresh + (2u << 20) - 1) & ~(uintptr_t)((2u << 20) - 1), hi = ((uintptr_t)fresh + n * 8) & ~(uintptr_t)((2u << 20) - 1);
            if (hi > lo) madvise((void *)lo, hi - lo, MADV_HUGEPAGE);
            zbra_hip_decode_i64(enc.data(), elen, n, fresh, &got);
            th += now() - t0;
            if (fresh[n - 1] != v[n - 1]) return 5;
            std::free(fresh);
        }
        th /= reps;
        std::printf("n=2^%d  encode %9.3f ms %6.2f GB/s | decode (reused dst) %9.3f ms %6.2f GB/s | decode (fresh dst) %9.3f ms %6.2f GB/s | fresh+MADV_HUGEPAGE %9.3f ms %6.2f GB/s\n",
                    lg, te * 1e3, 8.0 * n / te / 1e9, td * 1e3, 8.0 * n / td / 1e9, tf * 1e3, 8.0 * n / tf / 1e9, th * 1e3, 8.0 * n / th / 1e9);
        std::fflush(stdout);
    }
    return 0;
}
