// Fuzz driver for the .zbra walker (zbra_amd/csrc/zbra_file.hpp, pure host code), meant to be
// built with -fsanitize=address,undefined: random trees written with the container grammar of
// binary.rs:103-130, 184-356, then truncated, extended and corrupted.  Every image, valid or not,
// must be walked without touching a byte outside it; valid images must index exactly the arrays
// that were written.  Usage: fuzz_walker [iterations] [seed]
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../zbra_amd/csrc/zbra_file.hpp"

static uint64_t rng_state = 1;
static uint64_t rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return rng_state; }
static uint32_t below(uint32_t n) { return (uint32_t)(rnd() % n); }

struct Written { size_t offset, len, count; uint32_t kind, role; };
static std::vector<uint8_t> img;
static std::vector<Written> written;

static void u8(uint8_t v) { img.push_back(v); }
static void u32(uint32_t v) { for (int b = 0; b < 4; b++) img.push_back((uint8_t)(v >> (8 * b))); }
static void str() { uint32_t n = below(6); u32(n); for (uint32_t i = 0; i < n; i++) u8((uint8_t)('a' + below(26))); }
static void blob() { uint32_t n = below(40); u32(n); u32(n); for (uint32_t i = 0; i < n; i++) u8((uint8_t)rnd()); }
static void int_array(uint32_t kind, uint32_t role) {
    const uint32_t count = below(1000), clen = below(3) ? below(64) : 0;
    u32(count); u32(clen);
    written.push_back({img.size(), clen, count, kind, role});
    for (uint32_t i = 0; i < clen; i++) u8((uint8_t)rnd());
}
static void table(int depth);
static void column(int depth) {
    uint32_t tag = below(9);
    if (depth > 4 && tag >= 4) tag = below(4);
    u8((uint8_t)tag);
    switch (tag) {
        case 0: u32(below(100)); break;
        case 1: u8((uint8_t)below(2)); u8(0); u8((uint8_t)below(5)); int_array(ZBRA_ELEM_I64, ZBRA_ROLE_INT_VALUES); break;
        case 2: u8((uint8_t)below(2)); int_array(ZBRA_ELEM_I64, ZBRA_ROLE_DOUBLE_BITS); break;
        case 3: u8((uint8_t)below(2)); u8(1); u8((uint8_t)below(2)); int_array(ZBRA_ELEM_I64, ZBRA_ROLE_BINARY_LENGTHS); blob(); break;
        case 4: u8((uint8_t)below(2)); int_array(ZBRA_ELEM_I64, ZBRA_ROLE_ARRAY_LENGTHS); column(depth + 1); break;
        case 5: { u8((uint8_t)below(2)); uint32_t n = 1 + below(4); u32(n); for (uint32_t i = 0; i < n; i++) { str(); column(depth + 1); } break; }
        case 6: { u8((uint8_t)below(2)); int_array(ZBRA_ELEM_U32, ZBRA_ROLE_ENUM_TAGS); uint32_t n = 1 + below(3); u32(n);
                  for (uint32_t i = 0; i < n; i++) { str(); u32(below(10)); column(depth + 1); } break; }
        case 7: int_array(ZBRA_ELEM_I64, ZBRA_ROLE_NESTED_LENGTHS); table(depth + 1); break;
        default: column(depth + 1); break;
    }
}
static void table(int depth) {
    const uint32_t tag = depth > 4 ? 0 : below(3);
    u8((uint8_t)tag); u8((uint8_t)below(2));
    if (tag == 0) { u8(1); u8((uint8_t)below(2)); blob(); }
    else if (tag == 1) column(depth + 1);
    else { column(depth + 1); column(depth + 1); }
}
static void make_image() {
    img.clear(); written.clear();
    const char *magic = "||_ZBRA||00001||";
    img.insert(img.end(), magic, magic + 16);
    blob(); blob();
    const uint32_t nb = below(4);
    u32(nb);
    for (uint32_t b = 0; b < nb; b++) { u32(below(100000)); table(0); }
}

// Walks a heap copy of exactly `n` bytes, so that the sanitizer sees any access past the image.
static int walk(const uint8_t *data, size_t n, std::vector<zbra_file_array> *out, zbra_file_info *info) {
    uint8_t *copy = (uint8_t *)std::malloc(n ? n : 1);
    if (n) std::memcpy(copy, data, n);
    std::memset(info, 0, sizeof(*info));
    zbra::Walker w{copy, n};
    w.arrays = out;
    const bool ok = w.file(info);
    if (!ok && w.fail_pos > n) { std::printf("fail_pos %zu beyond image %zu\n", w.fail_pos, n); std::exit(1); }
    std::free(copy);
    return ok;
}

int main(int argc, char **argv) {
    const long iters = argc > 1 ? std::atol(argv[1]) : 20000;
    rng_state = argc > 2 ? std::strtoull(argv[2], nullptr, 0) : 0x9E3779B97F4A7C15ull;
    long valid_ok = 0, rejected = 0, accepted_mutants = 0;
    for (long it = 0; it < iters; it++) {
        make_image();
        std::vector<zbra_file_array> found;
        zbra_file_info info;
        if (!walk(img.data(), img.size(), &found, &info)) { std::printf("valid image rejected at iteration %ld\n", it); return 1; }
        if (found.size() != written.size() || info.n_arrays != written.size() || info.end_offset != img.size()) {
            std::printf("index mismatch at iteration %ld: %zu found, %zu written\n", it, found.size(), written.size());
            return 1;
        }
        for (size_t i = 0; i < found.size(); i++)
            if (found[i].offset != written[i].offset || found[i].len != written[i].len || found[i].count != written[i].count ||
                found[i].kind != written[i].kind || found[i].role != written[i].role) {
                std::printf("array %zu differs at iteration %ld\n", i, it);
                return 1;
            }
        valid_ok++;
        // every truncation of a short image, a sample of longer ones
        const size_t step = img.size() > 400 ? 1 + img.size() / 200 : 1;
        for (size_t cut = 0; cut < img.size(); cut += step) {
            found.clear();
            if (walk(img.data(), cut, &found, &info)) { std::printf("truncated image accepted (cut %zu of %zu, iteration %ld)\n", cut, img.size(), it); return 1; }
            rejected++;
        }
        // corrupted bytes: may be accepted or rejected, must stay inside the image
        for (int m = 0; m < 8; m++) {
            std::vector<uint8_t> bad = img;
            const int flips = 1 + (int)below(4);
            for (int f = 0; f < flips; f++) bad[16 + below((uint32_t)(bad.size() - 16))] = (uint8_t)rnd();
            found.clear();
            if (walk(bad.data(), bad.size(), &found, &info)) {
                accepted_mutants++;
                for (const zbra_file_array &a : found)
                    if (a.offset > bad.size() || a.len > bad.size() - a.offset) { std::printf("indexed array outside the image\n"); return 1; }
            } else rejected++;
        }
    }
    std::printf("%ld valid images indexed exactly, %ld rejections, %ld corrupted images accepted with in-bounds arrays\n", valid_ok, rejected,
                accepted_mutants);
    return 0;
}
