// C++ parity tests that read like the reference's own: zbra-core/src/compression.rs:337-406
// (unit + property tests) and zbra-core/tests/compression-pipeline.rs, through the C++ host
// mirror (include/zbra/compression.hpp, binary_io.hpp) -> C ABI -> HIP kernels, with every
// result also compared bit for bit with the CPU oracle (oracle/zbra_oracle.c, linked in as the
// checker).  Needs a GPU.
#include <cinttypes>
#include <cstdio>
#include <cstdlib>
#include <random>

#include "zbra/binary_io.hpp"
#include "zbra/compression.hpp"

extern "C" {
size_t zo_compress_bound(size_t n);
size_t zo_compress_int_array(const int64_t *values, size_t n, uint8_t *out);
int zo_decompress_int_array(const uint8_t *data, size_t len, size_t count, int64_t *out, size_t *out_count);
size_t zo_bp64_pack(const uint64_t *values, size_t n, uint8_t *out);
size_t zo_write_int_array_compressed(const int64_t *values, size_t n, uint8_t *out);
}

using namespace zbra::compression;
static int checks = 0, failures = 0;
#define CHECK(cond)                                                                   \
    do {                                                                              \
        ++checks;                                                                     \
        if (!(cond)) { ++failures; std::printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #cond); } \
    } while (0)

static std::vector<uint8_t> oracle_compress(const std::vector<int64_t> &v) {
    std::vector<uint8_t> out(zo_compress_bound(v.size()) + 1);
    out.resize(zo_compress_int_array(v.data(), v.size(), out.data()));
    return out;
}

// compression.rs:337-343
static void test_frame_of_reference_roundtrip() {
    std::vector<int64_t> values = {100, 102, 98, 101, 99, 103, 97};
    auto [midpoint, deltas] = frame_of_reference_encode(values);
    CHECK(midpoint == 100);
    CHECK(frame_of_reference_decode(midpoint, deltas) == values);
}
// compression.rs:345-351
static void test_zig_zag_roundtrip() {
    std::vector<int64_t> values = {-5, -1, 0, 1, 5, -100, 100};
    auto encoded = zig_zag_encode(values);
    CHECK((encoded == std::vector<uint64_t>{9, 1, 0, 2, 10, 199, 200}));
    CHECK(zig_zag_decode(encoded) == values);
}
// compression.rs:353-359
static void test_bp64_roundtrip() {
    std::vector<uint64_t> values = {0, 1, 2, 15, 255, 1000};
    auto packed = bp64_pack(values).unwrap();
    CHECK(packed.size() == 9 && packed[0] == 10);
    CHECK(bp64_unpack(packed, values.size()).unwrap() == values);
}
// compression.rs:361-367
static void test_full_int_compression_roundtrip() {
    std::vector<int64_t> values = {100, 102, 98, 101, 99, 103, 97, -5, -1, 0};
    auto compressed = compress_int_array(values).unwrap();
    CHECK(compressed == oracle_compress(values));
    CHECK(decompress_int_array(compressed, values.size()).unwrap() == values);
}
// compression.rs:378-406 (proptest shapes) + proptest-regressions/compression.txt:7-11
static void test_properties() {
    std::mt19937_64 rng(20250718);
    std::vector<std::vector<int64_t>> cases = {
        {4611686018427387904LL},
        {0, -1152921504606846976LL},
        {5823061857575085472LL, -3400310179279690336LL, -3400310179279690336LL},
        {0, -6818581005699659960LL, -6818581005699659960LL, -2404791031155115849LL, -6818581005699659960LL, 0},
        {0, -1152921504606846975LL},
        {INT64_MIN, INT64_MAX},
        {},
    };
    for (int c = 0; c < 64; c++) {
        std::vector<int64_t> v(rng() % 100);
        for (auto &x : v) x = (int64_t)rng();
        cases.push_back(v);
    }
    for (const auto &values : cases) {
        auto [midpoint, deltas] = frame_of_reference_encode(values);
        CHECK(frame_of_reference_decode(midpoint, deltas) == values);
        CHECK(zig_zag_decode(zig_zag_encode(values)) == values);
        auto compressed = compress_int_array(values).unwrap();
        CHECK(compressed == oracle_compress(values));
        CHECK(decompress_int_array(compressed, values.size()).unwrap() == values);
        std::vector<uint64_t> u(values.size());
        for (auto &x : u) x = rng() % 1000;
        auto packed = bp64_pack(u).unwrap();
        std::vector<uint8_t> op(1 + 8 * u.size() + 1);
        op.resize(zo_bp64_pack(u.data(), u.size(), op.data()));
        CHECK(packed == op);
        CHECK(bp64_unpack(packed, u.size()).unwrap() == u);
    }
}
// tests/compression-pipeline.rs:7-90, 208-237
static void test_timestamp_compression_pipeline() {
    const int64_t base_time = 4102444800000LL;
    std::vector<int64_t> timestamps;
    for (int i = 0; i < 168; i++) timestamps.push_back(base_time + (int64_t)i * 3600000);
    auto [midpoint, deltas] = frame_of_reference_encode(timestamps);
    int64_t max_mag = 0;
    for (auto d : deltas) max_mag = std::max<int64_t>(max_mag, d < 0 ? -d : d);
    CHECK(midpoint == 4102745400000LL);
    CHECK(64 - __builtin_clzll((unsigned long long)max_mag) <= 30);
    auto unsigned_deltas = zig_zag_encode(deltas);
    uint64_t max_unsigned = 0;
    for (auto z : unsigned_deltas) max_unsigned = std::max(max_unsigned, z);
    CHECK(max_unsigned == 601200000ULL);
    auto packed = bp64_pack(unsigned_deltas).unwrap();
    CHECK(packed[0] == 30 && packed.size() == 1 + 630);
    CHECK((double)(timestamps.size() * 8) / packed.size() > 1.5);
    auto unpacked = bp64_unpack(packed, timestamps.size()).unwrap();
    CHECK(frame_of_reference_decode(midpoint, zig_zag_decode(unpacked)) == timestamps);
    auto compressed = compress_int_array(timestamps).unwrap();
    CHECK(compressed.size() == 643 && compressed == oracle_compress(timestamps));
    CHECK(decompress_int_array(compressed, timestamps.size()).unwrap() == timestamps);
}
// tests/compression-pipeline.rs:93-121
static void test_compression_efficiency_threshold() {
    std::vector<uint64_t> small_values, large_values;
    for (uint64_t i = 0; i < 64; i++) { small_values.push_back(i * 1000); large_values.push_back((1ULL << 40) + i); }
    auto small_packed = bp64_pack(small_values).unwrap();
    auto large_packed = bp64_pack(large_values).unwrap();
    CHECK(small_packed.size() == 129 && small_packed[0] == 16);
    CHECK(large_packed.size() == 513 && large_packed[0] == 41);
    CHECK(bp64_unpack(large_packed, 64).unwrap() == large_values);
}
// tests/compression-pipeline.rs:131-205
static void test_frame_of_reference_effectiveness() {
    std::vector<int64_t> scattered = {1577836800000LL, 4102444800000LL, 7258118400000LL};
    auto [mid, deltas] = frame_of_reference_encode(scattered);
    CHECK(mid == 4102444800000LL);
    int64_t max_mag = 0;
    for (auto d : deltas) max_mag = std::max<int64_t>(max_mag, d < 0 ? -d : d);
    CHECK(64 - __builtin_clzll((unsigned long long)max_mag) > 32);
    auto e = compress_int_array(scattered).unwrap();
    CHECK(e.size() == 37 && e[12] == 43);
}
// error and empty conventions: compression.rs:272-274, 296-304, 312-316, 182-189
static void test_error_conventions() {
    CHECK(compress_int_array({}).unwrap().empty());
    CHECK(decompress_int_array({}, 5).unwrap().empty());
    CHECK(bp64_unpack({}, 5).unwrap().empty());
    auto r = decompress_int_array(std::vector<uint8_t>(11, 0), 1);
    CHECK(r.is_err() && r.error().kind == zbra::BinaryError::DecompressionError &&
          r.error().message == "Invalid compressed data length");
    auto e = compress_int_array({1, 2, 3, 4, 5}).unwrap();
    CHECK((e == std::vector<uint8_t>{3, 0, 0, 0, 0, 0, 0, 0, 3, 0, 0, 0, 3, 0x0b, 0x44}));
    auto cut = e; cut.pop_back();
    auto r2 = decompress_int_array(cut, 5);
    CHECK(r2.is_err() && r2.error().message == "Insufficient data for packed array");
}
// binary.rs:639-674 framing through the mirror helpers, bytes against the oracle's frame
static void test_framing() {
    std::vector<int64_t> values;
    for (int i = 0; i < 5000; i++) values.push_back(1700000000000LL + (i * 7919) % 4000);
    std::vector<uint8_t> w;
    CHECK(zbra::binary::write_int_array_compressed(w, values).is_ok());
    std::vector<uint8_t> ow(8 + zo_compress_bound(values.size()));
    ow.resize(zo_write_int_array_compressed(values.data(), values.size(), ow.data()));
    CHECK(w == ow);
    std::vector<uint32_t> tags;
    for (int i = 0; i < 3001; i++) tags.push_back((uint32_t)((i * 31) % 4 == 3 ? 7 : (i * 31) % 4));
    CHECK(zbra::binary::write_u32_array_compressed(w, tags).is_ok());
    std::vector<size_t> lengths = {0, 3, 17, 4096, 1};
    CHECK(zbra::binary::write_int_array_usize_compressed(w, lengths).is_ok());
    zbra::binary::Reader r(w);
    CHECK(zbra::binary::read_int_array_compressed(r).unwrap() == values);
    CHECK(zbra::binary::read_u32_array_compressed(r).unwrap() == tags);
    CHECK(zbra::binary::read_int_array_usize_compressed(r).unwrap() == lengths);
    CHECK(r.pos == w.size());
    CHECK(zbra::binary::read_int_array_compressed(r).is_err());  // EOF
}

// A whole image through the batched writer and the indexed reader: Blocks of an Enum column (tags
// + one Int column per variant), written with the container grammar of binary.rs:103-130, 184-356.
static void test_file_write_and_read_batched() {
    using namespace zbra::binary;
    const int nblocks = 3;
    std::vector<std::vector<uint32_t>> tags(nblocks);
    std::vector<std::vector<int64_t>> va(nblocks), vb(nblocks);
    for (int b = 0; b < nblocks; b++) {
        const int rows = 1000 + 777 * b;
        for (int i = 0; i < rows; i++) {
            const uint32_t t = (uint32_t)((i * 7 + b) % 3 == 0);
            tags[b].push_back(t);
            if (t == 0) va[b].push_back(1700000000000LL + (i * 7919) % 4000);
            else vb[b].push_back((int64_t)i * 1000003 - 5000000000LL);
        }
    }
    // first walk: everything but the int arrays, remembering where each frame goes
    std::vector<uint8_t> walk;
    std::vector<size_t> at;
    std::vector<IntArrayRef> refs;
    auto bytes = [&](const char *s) { walk.insert(walk.end(), s, s + std::strlen(s)); };
    auto sized = [&](const char *s) { write_u32(walk, (uint32_t)std::strlen(s)); write_u32(walk, (uint32_t)std::strlen(s)); bytes(s); };
    auto str = [&](const char *s) { write_u32(walk, (uint32_t)std::strlen(s)); bytes(s); };
    auto int_column = [&](const std::vector<int64_t> &v) {
        walk.push_back(1); walk.push_back(0); walk.push_back(0); walk.push_back(0);  // Int, Allow, int encoding, Int
        at.push_back(walk.size()); refs.emplace_back(v);
    };
    bytes("||_ZBRA||00001||");
    sized("{}");
    sized("{\"binary_data\":\"None\",\"strings\":\"None\"}");
    write_u32(walk, nblocks);
    for (int b = 0; b < nblocks; b++) {
        write_u32(walk, (uint32_t)tags[b].size());
        walk.push_back(1); walk.push_back(0);   // Table::Array, Allow
        walk.push_back(6); walk.push_back(0);   // Column::Enum, Allow
        at.push_back(walk.size()); refs.emplace_back(tags[b]);
        write_u32(walk, 2);
        str("a"); write_u32(walk, 0); int_column(va[b]);
        str("b"); write_u32(walk, 1); int_column(vb[b]);
    }
    // one batched encode, frames spliced in
    std::vector<uint8_t> frames;
    auto offs = write_int_arrays_compressed(frames, refs).unwrap();
    std::vector<uint8_t> image;
    size_t prev = 0;
    for (size_t i = 0; i < refs.size(); i++) {
        image.insert(image.end(), walk.begin() + prev, walk.begin() + at[i]);
        image.insert(image.end(), frames.begin() + offs[i], frames.begin() + offs[i + 1]);
        prev = at[i];
    }
    image.insert(image.end(), walk.begin() + prev, walk.end());
    // each frame is what the per-array helper (and the oracle) writes
    for (int b = 0; b < nblocks; b++) {
        std::vector<uint8_t> ow(8 + zo_compress_bound(va[b].size()));
        ow.resize(zo_write_int_array_compressed(va[b].data(), va[b].size(), ow.data()));
        const size_t i = 3 * b + 1;
        CHECK(std::vector<uint8_t>(frames.begin() + offs[i], frames.begin() + offs[i + 1]) == ow);
        std::vector<uint8_t> tw;
        CHECK(write_u32_array_compressed(tw, tags[b]).is_ok());
        CHECK(std::vector<uint8_t>(frames.begin() + offs[3 * b], frames.begin() + offs[3 * b + 1]) == tw);
    }
    auto f = read_file_int_arrays(image.data(), image.size()).unwrap();
    CHECK(f.info.block_count == (uint32_t)nblocks && f.info.n_arrays == (uint64_t)3 * nblocks && f.info.end_offset == image.size());
    for (int b = 0; b < nblocks; b++) {
        CHECK(f.arrays[3 * b].where.role == ZBRA_ROLE_ENUM_TAGS && f.arrays[3 * b].u32 == tags[b]);
        CHECK(f.arrays[3 * b + 1].where.role == ZBRA_ROLE_INT_VALUES && f.arrays[3 * b + 1].i64 == va[b]);
        CHECK(f.arrays[3 * b + 2].i64 == vb[b] && f.arrays[3 * b + 2].where.block == (uint32_t)b);
    }
    image[3] ^= 0x40;  // magic number
    CHECK(read_file_int_arrays(image.data(), image.size()).is_err());
}

int main() {
    int ndev = 0;
    if (zbra_hip_device_count(&ndev) != ZBRA_OK || ndev == 0) { std::printf("no device: these tests need a GPU\n"); return 2; }
    test_frame_of_reference_roundtrip();
    test_zig_zag_roundtrip();
    test_bp64_roundtrip();
    test_full_int_compression_roundtrip();
    test_properties();
    test_timestamp_compression_pipeline();
    test_compression_efficiency_threshold();
    test_frame_of_reference_effectiveness();
    test_error_conventions();
    test_framing();
    test_file_write_and_read_batched();
    std::printf("%d checks, %d failures\n", checks, failures);
    return failures ? 1 : 0;
}
