// C++ mirror of the int-array framing helpers of zbra-core/src/binary.rs:639-674:
//   u32 count LE | u32 compressed_len LE | compressed bytes
// over a byte vector (Write) and a cursor (Read).  The u32 variant narrows / widens on the device
// (no intermediate Vec<i64> as in binary.rs:666-674); the usize variant is the i64 path.
#pragma once
#include <cstring>

#include "compression.hpp"

namespace zbra {
namespace binary {

// std::io::Cursor<&[u8]> stand-in.
struct Reader {
    const uint8_t *data;
    size_t len;
    size_t pos = 0;
    Reader(const uint8_t *d, size_t n) : data(d), len(n) {}
    explicit Reader(const std::vector<uint8_t> &v) : data(v.data()), len(v.size()) {}
    bool read_exact(void *dst, size_t n) {  // false = BinaryError::IoError(UnexpectedEof)
        if (len - pos < n) return false;
        std::memcpy(dst, data + pos, n);
        pos += n;
        return true;
    }
};

inline void write_u32(std::vector<uint8_t> &w, uint32_t v) {
    for (int b = 0; b < 4; b++) w.push_back((uint8_t)(v >> (8 * b)));
}
inline bool read_u32(Reader &r, uint32_t *v) {
    uint8_t b[4];
    if (!r.read_exact(b, 4)) return false;
    *v = (uint32_t)b[0] | (uint32_t)b[1] << 8 | (uint32_t)b[2] << 16 | (uint32_t)b[3] << 24;
    return true;
}

inline BinaryError eof_error() { return BinaryError{BinaryError::DecompressionError, "failed to fill whole buffer", -1}; }

// write_int_array_compressed                                             binary.rs:639-645
inline Result<bool> write_int_array_compressed(std::vector<uint8_t> &w, const std::vector<int64_t> &values) {
    write_u32(w, (uint32_t)values.size());                       // `values.len() as u32`
    auto compressed = compression::compress_int_array(values);
    if (compressed.is_err()) return compressed.error();
    write_u32(w, (uint32_t)compressed.value().size());           // `compressed.len() as u32`
    w.insert(w.end(), compressed.value().begin(), compressed.value().end());
    return true;
}

// read_int_array_compressed                                              binary.rs:648-654
inline Result<std::vector<int64_t>> read_int_array_compressed(Reader &r) {
    uint32_t len = 0, compressed_size = 0;
    if (!read_u32(r, &len) || !read_u32(r, &compressed_size)) return eof_error();
    std::vector<uint8_t> compressed(compressed_size);
    if (!r.read_exact(compressed.data(), compressed_size)) return eof_error();
    return compression::decompress_int_array(compressed, len);
}

// write_int_array_usize_compressed / read_…                              binary.rs:656-664
inline Result<bool> write_int_array_usize_compressed(std::vector<uint8_t> &w, const std::vector<size_t> &values) {
    std::vector<int64_t> v(values.begin(), values.end());       // `v as i64`
    return write_int_array_compressed(w, v);
}
inline Result<std::vector<size_t>> read_int_array_usize_compressed(Reader &r) {
    auto v = read_int_array_compressed(r);
    if (v.is_err()) return v.error();
    return std::vector<size_t>(v.value().begin(), v.value().end());  // `v as usize`
}

// write_u32_array_compressed                                             binary.rs:666-669
inline Result<bool> write_u32_array_compressed(std::vector<uint8_t> &w, const std::vector<uint32_t> &values) {
    write_u32(w, (uint32_t)values.size());
    std::vector<uint8_t> out(zbra_hip_encode_bound(values.size()));
    size_t n = 0;
    const int st = zbra_hip_encode_u32(values.data(), values.size(), out.data(), out.size(), &n);
    if (st != ZBRA_OK) return detail::make_error(st, BinaryError::CompressionError);
    write_u32(w, (uint32_t)n);
    w.insert(w.end(), out.begin(), out.begin() + n);
    return true;
}

// read_u32_array_compressed                                              binary.rs:671-674
inline Result<std::vector<uint32_t>> read_u32_array_compressed(Reader &r) {
    uint32_t len = 0, compressed_size = 0;
    if (!read_u32(r, &len) || !read_u32(r, &compressed_size)) return eof_error();
    if (r.len - r.pos < compressed_size) return eof_error();
    std::vector<uint32_t> out(len);
    size_t got = 0;
    const int st = zbra_hip_decode_u32(r.data + r.pos, compressed_size, len, out.data(), &got);
    r.pos += compressed_size;
    if (st != ZBRA_OK) return detail::make_error(st, BinaryError::DecompressionError);
    out.resize(got);
    return out;
}

}  // namespace binary
}  // namespace zbra
