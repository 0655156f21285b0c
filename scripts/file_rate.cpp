// The full-file configuration from C++: a .zbra image of B Blocks of 2^R rows x 4 Int columns
// (attribute id: < 200, entity id: clustered runs, time: milliseconds, monotone, value: < 4000)
// built by the C++ tree writer (zbra::binary::BinaryFile::to_bytes: the walk of
// zbra-core/src/binary.rs:103-130, 184-356 with ONE batched framed encode), then read back
// host memory -> host memory (zbra_hip_file_index + zbra_hip_file_decode behind
// read_file_int_arrays) and compared with what was written.  Same rows as scripts/bench_file.py.
//
//   g++ -O2 -std=c++17 -Iinclude scripts/file_rate.cpp -o /tmp/file_rate -Lzbra_amd -lzbra_hip -Wl,-rpath,$PWD/zbra_amd
//   /tmp/file_rate [blocks=24] [rows_log2=22] [reps=3]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "zbra/binary_file.hpp"

using namespace zbra;
using namespace zbra::striped;

static uint64_t sm(uint64_t seed, uint64_t i) {
    uint64_t z = seed + (i + 1) * 0x9E3779B97F4A7C15ull;
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull;
    z ^= z >> 27; z *= 0x94D049BB133111EBull;
    z ^= z >> 31;
    return z;
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char **argv) {
    const size_t blocks = argc > 1 ? (size_t)atol(argv[1]) : 24;
    const int rows_log2 = argc > 2 ? atoi(argv[2]) : 22;
    const int reps = argc > 3 ? atoi(argv[3]) : 3;
    const size_t n = (size_t)1 << rows_log2;
    if (zbra_hip_init() != ZBRA_OK) { std::printf("{\"error\": \"no device\"}\n"); return 2; }

    binary::BinaryFile file;
    file.schema_json = "{}";
    for (size_t b = 0; b < blocks; b++) {
        std::vector<int64_t> attribute(n), entity(n), t_ms(n), value(n);
        for (size_t k = 0; k < n; k++) {
            const uint64_t i = b * n + k;
            attribute[k] = (int64_t)(sm(0x5EED0005, i) % 200);
            entity[k] = (int64_t)(1000 + (i / 64) % 50000);
            t_ms[k] = 1700000000000LL + (int64_t)i * 1000;
            value[k] = (int64_t)(sm(0x5EED1005, i) % 4000);
        }
        std::vector<FieldColumn> fields;
        fields.push_back(FieldColumn{"attribute_id", Column::int_(std::move(attribute))});
        fields.push_back(FieldColumn{"entity_id", Column::int_(std::move(entity))});
        fields.push_back(FieldColumn{"time", Column::int_(std::move(t_ms), 3 /* TimeMilliseconds */)});
        fields.push_back(FieldColumn{"value", Column::int_(std::move(value))});
        binary::Block blk;
        blk.row_count = (uint32_t)n;
        blk.table = Table::array(Column::struct_(std::move(fields)));
        file.blocks.push_back(std::move(blk));
    }
    const double values_bytes = 8.0 * 4 * (double)n * (double)blocks;

    double t_write = 1e30;
    std::vector<uint8_t> image;
    for (int r = 0; r < reps; r++) {
        const double t0 = now();
        auto img = file.to_bytes();
        const double dt = now() - t0;
        if (img.is_err()) { std::printf("{\"error\": \"%s\"}\n", img.error().message.c_str()); return 1; }
        if (dt < t_write) t_write = dt;
        image = std::move(img).unwrap();
    }

    double t_read = 1e30;
    bool same = true;
    for (int r = 0; r < reps; r++) {
        const double t0 = now();
        auto dec = binary::read_file_int_arrays(image.data(), image.size());
        const double dt = now() - t0;
        if (dec.is_err()) { std::printf("{\"error\": \"%s\"}\n", dec.error().message.c_str()); return 1; }
        if (dt < t_read) t_read = dt;
        if (r == 0) {  // what was read is what was written, array by array in file order
            const binary::DecodedFile &df = dec.value();
            same = df.arrays.size() == 4 * blocks;
            for (size_t b = 0; b < blocks && same; b++)
                for (size_t f = 0; f < 4 && same; f++) {
                    const std::vector<int64_t> &want = file.blocks[b].table->column->fields[f].column->values;
                    const std::vector<int64_t> &got = df.arrays[4 * b + f].i64;
                    same = got.size() == want.size() && std::memcmp(got.data(), want.data(), 8 * want.size()) == 0;
                }
        }
    }
    std::printf("{\"what\": \"full-file configuration from the C++ writer: %zu Blocks of 2^%d rows x 4 Int columns\", \"image_bytes\": %zu, "
                "\"values_bytes\": %.0f, \"to_bytes_s\": %.4f, \"to_bytes_values_GBps\": %.2f, \"read_host_to_host_s\": %.4f, "
                "\"read_decoded_GBps\": %.2f, \"round_trip_equal\": %s}\n",
                blocks, rows_log2, image.size(), values_bytes, t_write, values_bytes / t_write / 1e9, t_read,
                values_bytes / t_read / 1e9, same ? "true" : "false");
    return same ? 0 : 1;
}
