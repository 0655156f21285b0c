// The C++ tree writer / reader of include/zbra/binary_file.hpp (Table::write_to, Column::write_to,
// their read_from counterparts: zbra-core/src/binary.rs:204-237, 275-356, 240-270, 359-441) on a
// table with every column kind:
//   - the image of the batched writer equals the image of a reference-style walk that encodes each
//     array on the spot with the CPU oracle (write_int_array_compressed, binary.rs:639-645);
//   - from_bytes gives the tree back;
//   - `--dump PATH` leaves the image for tests/test_cpp_binary_file.py, which builds the same table in
//     Python (same counter-based generator) and compares the bytes.
// Exit status: 0 = all checks passed, 1 = a check failed, 2 = no GPU.
#include <cstdio>
#include <cstring>
#include <fstream>

#include "zbra/binary_file.hpp"

extern "C" {
size_t zo_compress_bound(size_t n);
size_t zo_write_int_array_compressed(const int64_t *values, size_t n, uint8_t *out);
uint64_t zo_sm(uint64_t seed, uint64_t i);
}

using namespace zbra;
using namespace zbra::striped;

static int failures = 0, checks = 0;
#define CHECK(cond)                                                                          \
    do {                                                                                     \
        checks++;                                                                            \
        if (!(cond)) { failures++; std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); } \
    } while (0)

// The table of tests/test_cpp_binary_file.py::rich_table, value for value.
static TablePtr rich_table(size_t rows, uint64_t s) {
    std::vector<int64_t> ints(rows), wide(rows);
    std::vector<double> dbl(rows);
    std::vector<size_t> strlens(rows), arrlens(rows), ones(rows, 1);
    std::vector<uint32_t> tags(rows);
    static const uint32_t TAGS[4] = {0, 1, 2, 7};
    size_t counts[4] = {0, 0, 0, 0}, strtotal = 0, arrtotal = 0;
    for (size_t i = 0; i < rows; i++) {
        ints[i] = 1700000000000LL + (int64_t)(zo_sm(s, i) % 4000);
        wide[i] = (int64_t)zo_sm(s + 1, i);
        const uint64_t bits = zo_sm(s + 2, i);
        std::memcpy(&dbl[i], &bits, 8);
        strlens[i] = zo_sm(s + 3, i) % 4; strtotal += strlens[i];
        arrlens[i] = zo_sm(s + 4, i) % 5; arrtotal += arrlens[i];
        const uint32_t v = (uint32_t)(zo_sm(s + 6, i) % 4);
        tags[i] = TAGS[v];
        counts[v]++;
    }
    std::vector<uint8_t> strdata(strtotal);
    for (size_t k = 0; k < strtotal; k++) strdata[k] = (uint8_t)('a' + k % 26);
    std::vector<int64_t> elems(arrtotal);
    for (size_t k = 0; k < arrtotal; k++) elems[k] = (int64_t)(zo_sm(s + 5, k) % 2000) - 1000;
    std::vector<int64_t> bx(counts[1]), by(counts[1]), cz(counts[2]), keys(2 * counts[3]);
    for (size_t k = 0; k < counts[1]; k++) { bx[k] = 5000000 + (int64_t)(zo_sm(s + 7, k) % 1000); by[k] = 1640000000000LL + (int64_t)k * 1000; }
    for (size_t k = 0; k < counts[2]; k++) cz[k] = (int64_t)(zo_sm(s + 8, k) % (1ull << 30)) - (1ll << 29);
    std::vector<double> mapvals(2 * counts[3]);
    for (size_t k = 0; k < 2 * counts[3]; k++) { keys[k] = (int64_t)k; const uint64_t bits = k; std::memcpy(&mapvals[k], &bits, 8); }
    std::vector<uint8_t> blob(2 * rows);
    for (size_t k = 0; k < rows; k++) { blob[2 * k] = 0; blob[2 * k + 1] = 1; }
    return Table::array(Column::struct_({
        {"ints", Column::int_(ints)},
        {"wide", Column::int_(wide)},
        {"dbl", Column::double_(dbl)},
        {"str", Column::binary(strlens, strdata, 1 /* Utf8 */)},
        {"arr", Column::array(arrlens, Column::int_(elems, 1 /* Date */))},
        {"sum", Column::enum_(tags, {{"a", 0, Column::unit(counts[0])},
                                     {"b", 1, Column::struct_({{"x", Column::int_(bx)}, {"y", Column::int_(by, 3 /* TimeMilliseconds */)}})},
                                     {"c", 2, Column::reversed(Column::int_(cz))},
                                     {"d", 7, Column::nested(std::vector<size_t>(counts[3], 2), Table::map(Column::int_(keys), Column::double_(mapvals)))}})},
        {"blob", Column::nested(ones, Table::binary(blob))},
    }));
}

// The reference's way: one walk, every int array encoded where it is met (here by the CPU oracle).
static std::vector<uint8_t> oracle_image(const binary::BinaryFile &f) {
    binary::TreeWriter w(true, true);
    w.bytes(binary::MAGIC_NUMBER, 16);
    w.sized(f.schema_json.data(), f.schema_json.size());
    w.sized(f.compression_json.data(), f.compression_json.size());
    w.u32((uint32_t)f.blocks.size());
    for (const auto &b : f.blocks) { w.u32(b.row_count); w.table(*b.table); }
    std::vector<uint8_t> image;
    size_t prev = 0;
    for (size_t i = 0; i < w.arrays.size(); i++) {
        image.insert(image.end(), w.out.begin() + prev, w.out.begin() + w.cuts[i]);
        const binary::IntArrayRef &a = w.arrays[i];
        std::vector<int64_t> widened;  // write_u32_array_compressed widens `as i64` first (binary.rs:666-669)
        const int64_t *v = (const int64_t *)a.values;
        if (a.is_u32) { widened.assign((const uint32_t *)a.values, (const uint32_t *)a.values + a.n); v = widened.data(); }
        std::vector<uint8_t> frame(8 + zo_compress_bound(a.n) + 1);
        frame.resize(zo_write_int_array_compressed(v, a.n, frame.data()));
        image.insert(image.end(), frame.begin(), frame.end());
        prev = w.cuts[i];
    }
    image.insert(image.end(), w.out.begin() + prev, w.out.end());
    return image;
}

int main(int argc, char **argv) {
    int devices = 0;
    if (zbra_hip_device_count(&devices) != ZBRA_OK || devices == 0) {
        std::printf("need a GPU: libzbra_hip has no CPU path\n");
        return 2;
    }
    binary::BinaryFile f;
    f.schema_json = "{}";
    const size_t rows[] = {0, 1, 257, 5000, 70001};
    for (size_t i = 0; i < sizeof(rows) / sizeof(rows[0]); i++) {
        binary::Block b;
        b.row_count = (uint32_t)rows[i];
        b.table = rich_table(rows[i], 100 + 10 * i);
        f.blocks.push_back(b);
    }
    auto image = f.to_bytes();
    CHECK(image.is_ok());
    if (image.is_ok()) {
        CHECK(image.value() == oracle_image(f));
        auto back = binary::BinaryFile::from_bytes(image.value());
        CHECK(back.is_ok());
        if (back.is_ok()) {
            CHECK(back.value().schema_json == f.schema_json && back.value().compression_json == f.compression_json);
            CHECK(back.value().blocks.size() == f.blocks.size());
            for (size_t i = 0; i < f.blocks.size() && i < back.value().blocks.size(); i++) {
                CHECK(back.value().blocks[i].row_count == f.blocks[i].row_count);
                CHECK(striped::equal(back.value().blocks[i].table, f.blocks[i].table));
            }
        }
        for (int a = 1; a + 1 < argc; a++)
            if (std::strcmp(argv[a], "--dump") == 0) {
                std::ofstream o(argv[a + 1], std::ios::binary);
                o.write((const char *)image.value().data(), (std::streamsize)image.value().size());
            }
    }
    // a byte blob under a zstd config is refused (zstd is another path); a tree without blobs is not
    binary::BinaryFile z;
    z.schema_json = "{}";
    z.compression_json = "{\"binary_data\":{\"Zstd\":{\"level\":3}},\"strings\":{\"Zstd\":{\"level\":3}}}";
    binary::Block zb;
    zb.row_count = 3;
    zb.table = Table::array(Column::int_({1, 2, 3}));
    z.blocks.push_back(zb);
    auto zi = z.to_bytes();
    CHECK(zi.is_ok());
    if (zi.is_ok()) {  // ... and reads back under that config: it holds no blob
        auto zr = binary::BinaryFile::from_bytes(zi.value());
        CHECK(zr.is_ok() && zr.value().compression_json == z.compression_json && striped::equal(zr.value().blocks[0].table, zb.table));
    }
    z.blocks[0].table = Table::array(Column::binary({2}, {'h', 'i'}));
    CHECK(z.to_bytes().is_err());
    // The read side of the same rule: an image whose config says Zstd for its blobs, as the reference
    // writes by default (compression.rs:34-41), holds compressed bytes where the blob is
    // (binary.rs:690-699 -> decompress_binary).  They must not come back as the column's data.
    // The image is the one the None config writes, with the config document swapped.
    binary::BinaryFile plain = z;
    plain.compression_json = binary::COMPRESSION_NONE;
    auto pi = plain.to_bytes();
    CHECK(pi.is_ok());
    if (pi.is_ok()) {
        auto pr = binary::BinaryFile::from_bytes(pi.value());
        CHECK(pr.is_ok() && striped::equal(pr.value().blocks[0].table, plain.blocks[0].table));
        zbra_file_info info;
        CHECK(zbra_hip_file_index(pi.value().data(), pi.value().size(), nullptr, 0, &info, nullptr) == ZBRA_OK);
        const std::vector<uint8_t> &src = pi.value();
        std::vector<uint8_t> swapped(src.begin(), src.begin() + (long)info.compression_offset - 8);
        const uint32_t zl = (uint32_t)z.compression_json.size();
        for (int rep = 0; rep < 2; rep++)
            for (int k = 0; k < 4; k++) swapped.push_back((uint8_t)(zl >> (8 * k)));
        swapped.insert(swapped.end(), z.compression_json.begin(), z.compression_json.end());
        swapped.insert(swapped.end(), src.begin() + (long)(info.compression_offset + info.compression_len), src.end());
        auto sr = binary::BinaryFile::from_bytes(swapped);
        CHECK(sr.is_err());
        if (sr.is_err()) CHECK(sr.error().kind == BinaryError::DecompressionError);
        // strings only: binary_data may stay Zstd when the tree holds no Table::Binary
        binary::BinaryFile half = plain;
        half.compression_json = "{ \"binary_data\": {\"Zstd\":{\"level\":3}}, \"strings\": \"None\" }";
        auto hi = half.to_bytes();
        CHECK(hi.is_ok());
        if (hi.is_ok()) CHECK(binary::BinaryFile::from_bytes(hi.value()).is_ok());
    }
    std::printf("%d checks, %d failures\n", checks, failures);
    return failures ? 1 : 0;
}
