// TIMED CPU BASELINE ONLY (bench.py's cpu_baseline leg); test infrastructure like the rest of oracle/.
//
// compress_int_array as zbra_oracle.c restates it, except for one thing: the median's sort.  The
// reference calls Vec<i64>::sort() (zbra-core/src/compression.rs:58-59), a stable merge sort whose
// comparison is inlined; C's qsort calls a comparison function through a pointer for every pair
// and is several times slower, which would flatter the GPU.  This translation unit sorts with
// std::stable_sort (merge sort, inlined operator<) and reuses the oracle's other stages, so the
// bytes are the oracle's and the time is a fair stand-in for the Rust build's.
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

extern "C" {
void zo_zig_zag_encode(const int64_t *values, size_t n, uint64_t *out);
size_t zo_bp64_pack(const uint64_t *values, size_t n, uint8_t *out);

// Same contract as zo_compress_int_array.
size_t zb_compress_int_array_stable_sort(const int64_t *values, size_t n, uint8_t *out) {
    if (n == 0) return 0;
    // frame_of_reference_encode, compression.rs:52-74
    std::vector<int64_t> sorted(values, values + n);  // values.to_vec()
    std::stable_sort(sorted.begin(), sorted.end());   // sorted.sort()
    int64_t midpoint;
    if (n % 2 == 0) {
        const uint64_t a = (uint64_t)sorted[n / 2 - 1], b = (uint64_t)sorted[n / 2];
        midpoint = (int64_t)(a + (uint64_t)((int64_t)(b - a) / 2));  // a.wrapping_add(b.wrapping_sub(a) / 2)
    } else {
        midpoint = sorted[n / 2];
    }
    std::vector<int64_t> deltas(n);
    for (size_t i = 0; i < n; i++) deltas[i] = (int64_t)((uint64_t)values[i] - (uint64_t)midpoint);
    std::vector<uint64_t> unsigned_values(n);
    zo_zig_zag_encode(deltas.data(), n, unsigned_values.data());
    std::vector<uint8_t> packed(1 + 8 * n);
    const size_t packed_len = zo_bp64_pack(unsigned_values.data(), n, packed.data());
    const uint64_t m = (uint64_t)midpoint;
    for (int b = 0; b < 8; b++) out[b] = (uint8_t)(m >> (8 * b));
    const uint32_t pl = (uint32_t)packed_len;
    for (int b = 0; b < 4; b++) out[8 + b] = (uint8_t)(pl >> (8 * b));
    std::memcpy(out + 12, packed.data(), packed_len);
    return 12 + packed_len;
}
}
