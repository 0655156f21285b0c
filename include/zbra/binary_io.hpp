// C++ mirror of the int-array framing helpers of zbra-core/src/binary.rs:639-674:
//   u32 count LE | u32 compressed_len LE | compressed bytes
// over a byte vector (Write) and a cursor (Read).  The u32 variant narrows / widens on the device
// (no intermediate Vec<i64> as in binary.rs:666-674); the usize variant is the i64 path.
#pragma once
#include <algorithm>
#include <cstring>
#include <memory>
#include <string>
#include <thread>

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

// ---- batched forms: what a writer / reader of a whole file uses ----------------------------------

// One int array of a tree walk: Int values, Double bits, lengths (all 8-byte elements) or enum
// tags (4-byte elements).
struct IntArrayRef {
    const void *values;
    size_t n;
    bool is_u32;
    IntArrayRef(const void *p, size_t count, bool narrow) : values(p), n(count), is_u32(narrow) {}
    IntArrayRef(const std::vector<int64_t> &v) : values(v.data()), n(v.size()), is_u32(false) {}
    IntArrayRef(const std::vector<uint32_t> &v) : values(v.data()), n(v.size()), is_u32(true) {}
};

// write_int_array_compressed / write_u32_array_compressed (binary.rs:639-645, 666-669) for every
// array of `arrays` in ONE submission: their frames are appended to `w` back to back;
// frame i = w[offsets[i] .. offsets[i+1]) relative to w's size at the call.
// The frames land in an UNINITIALISED staging buffer of the worst-case size (13 + 8 n bytes per
// array: value-initialising it, as a std::vector would, touches several times the bytes that are
// ever written) and are returned with their offsets: frame i = buf[offsets[i] .. offsets[i+1]).
struct StagedFrames {
    std::unique_ptr<uint8_t[]> buf;
    std::vector<uint64_t> offsets;  // arrays.size() + 1
};
inline Result<StagedFrames> encode_frames_staged(const std::vector<IntArrayRef> &arrays) {
    std::vector<zbra_encode_desc> descs(arrays.size());
    size_t bound = 0;
    for (size_t i = 0; i < arrays.size(); i++) {
        descs[i] = zbra_encode_desc{arrays[i].values, arrays[i].n, nullptr, 0,
                                    (uint32_t)(arrays[i].is_u32 ? ZBRA_ELEM_U32 : ZBRA_ELEM_I64), 0};
        bound += 8 + zbra_hip_encode_bound(arrays[i].n);
    }
    StagedFrames f;
    f.buf.reset(new uint8_t[bound ? bound : 1]);
    f.offsets.assign(arrays.size() + 1, 0);
    const int st = zbra_hip_encode_frames(descs.data(), descs.size(), f.buf.get(), bound, f.offsets.data(), nullptr);
    if (st != ZBRA_OK) return detail::make_error(st, BinaryError::CompressionError);
    return f;
}
inline Result<std::vector<size_t>> write_int_arrays_compressed(std::vector<uint8_t> &w, const std::vector<IntArrayRef> &arrays) {
    auto staged = encode_frames_staged(arrays);
    if (staged.is_err()) return staged.error();
    const StagedFrames &f = staged.value();
    w.insert(w.end(), f.buf.get(), f.buf.get() + f.offsets.back());
    return std::vector<size_t>(f.offsets.begin(), f.offsets.end());
}

// Every int array of a .zbra image, found without decoding (zbra_hip_file_index) and then decoded
// in one pipelined pass (zbra_hip_file_decode): BinaryFile::read_from's int arrays
// (binary.rs:139-174, 359-441) in file order.  Enum tags land in `u32`, everything else in `i64`.
struct DecodedArray {
    zbra_file_array where;
    std::vector<int64_t> i64;
    std::vector<uint32_t> u32;
};
struct DecodedFile {
    zbra_file_info info;
    std::vector<DecodedArray> arrays;
};
inline Result<DecodedFile> read_file_int_arrays(const uint8_t *image, size_t len) {
    DecodedFile f;
    size_t err_at = 0;
    int st = zbra_hip_file_index(image, len, nullptr, 0, &f.info, &err_at);
    if (st != ZBRA_OK) return BinaryError{BinaryError::DecompressionError, std::string(zbra_hip_strerror(st)) + " at offset " + std::to_string(err_at), st};
    std::vector<zbra_file_array> idx(f.info.n_arrays);
    if (!idx.empty()) {
        st = zbra_hip_file_index(image, len, idx.data(), idx.size(), &f.info, &err_at);
        if (st != ZBRA_OK) return detail::make_error(st, BinaryError::DecompressionError);
    }
    f.arrays.resize(idx.size());
    std::vector<void *> outs(idx.size());
    for (size_t i = 0; i < idx.size(); i++) {
        f.arrays[i].where = idx[i];
        // the length of the Vec decompress_int_array returns: 0 for empty data and for an empty
        // packed slice whatever the frame's count says (compression.rs:296-298, 182-184)
        size_t n = 0;
        st = zbra_hip_decoded_count(image + idx[i].offset, idx[i].len, idx[i].count, &n);
        if (st != ZBRA_OK) return detail::make_error(st, BinaryError::DecompressionError);
        idx[i].count = n;  // what is asked of the decode: an array that yields nothing has no output buffer
    }
    // A std::vector comes value-initialised: gigabytes of zeros written (and pages faulted in) by
    // one thread took ten times as long as the decode itself.  Several threads share the arrays.
    {
        const unsigned hw = std::thread::hardware_concurrency();
        const size_t nthreads = idx.size() < 2 ? 1 : std::min<size_t>(std::min<size_t>(8, hw ? hw / 2 + 1 : 2), idx.size());
        auto allocate = [&](size_t first) {
            for (size_t i = first; i < idx.size(); i += nthreads) {
                if (idx[i].kind == ZBRA_ELEM_U32) { f.arrays[i].u32.resize(idx[i].count); outs[i] = f.arrays[i].u32.data(); }
                else { f.arrays[i].i64.resize(idx[i].count); outs[i] = f.arrays[i].i64.data(); }
            }
        };
        std::vector<std::thread> pool;
        for (size_t t = 1; t < nthreads; t++) pool.emplace_back(allocate, t);
        allocate(0);
        for (std::thread &t : pool) t.join();
    }
    st = zbra_hip_file_decode(image, len, idx.data(), idx.size(), outs.data(), nullptr);
    if (st != ZBRA_OK) return detail::make_error(st, BinaryError::DecompressionError);
    return f;
}

}  // namespace binary
}  // namespace zbra
