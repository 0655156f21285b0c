// C++ host-side mirror of `zbra_core::compression` (zbra-core/src/compression.rs:43-330) over the
// C ABI of libzbra_hip.so.  Same eight public names, argument meaning, results and error
// behaviour as the Rust module: slices become std::vector, Result<T, BinaryError> becomes
// zbra::Result<T>.  Every function runs HIP kernels; there is no CPU path (a missing device is a
// DeviceError).
#pragma once
#include <cstdint>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "zbra_hip.h"

namespace zbra {

// zbra_core::error::BinaryError, the variants this path can produce (zbra-core/src/error.rs).
struct BinaryError {
    enum Kind { DecompressionError, CompressionError, DeviceError } kind;
    std::string message;
    int status;  // zbra_status behind it
};

// Rust's Result<T, BinaryError>.
template <typename T>
class Result {
public:
    Result(T v) : ok_(true), value_(std::move(v)) {}
    Result(BinaryError e) : ok_(false), error_(std::move(e)) {}
    bool is_ok() const { return ok_; }
    bool is_err() const { return !ok_; }
    const T &value() const { if (!ok_) throw std::runtime_error("Result::value on Err: " + error_.message); return value_; }
    T unwrap() && { if (!ok_) throw std::runtime_error("called unwrap() on an Err value: " + error_.message); return std::move(value_); }
    T unwrap() const & { if (!ok_) throw std::runtime_error("called unwrap() on an Err value: " + error_.message); return value_; }
    const BinaryError &error() const { return error_; }
private:
    bool ok_;
    T value_{};
    BinaryError error_{BinaryError::DeviceError, "", 0};
};

namespace detail {
inline BinaryError make_error(int st, BinaryError::Kind fallback) {
    BinaryError::Kind k = fallback;
    if (st == ZBRA_ERR_SHORT_HEADER || st == ZBRA_ERR_SHORT_PACKED || st == ZBRA_ERR_TRUNCATED) k = BinaryError::DecompressionError;
    else if (st == ZBRA_ERR_NO_DEVICE || st >= ZBRA_ERR_HIP_BASE) k = BinaryError::DeviceError;
    return BinaryError{k, zbra_hip_strerror(st), st};
}
// The stage functions are infallible in the reference; here only the device can fail.
inline void check_infallible(int st) {
    if (st != ZBRA_OK) throw std::runtime_error(std::string("libzbra_hip: ") + zbra_hip_strerror(st));
}
}  // namespace detail

namespace compression {

// frame_of_reference_encode(values: &[i64]) -> (i64, Vec<i64>)          compression.rs:52-74
inline std::pair<int64_t, std::vector<int64_t>> frame_of_reference_encode(const std::vector<int64_t> &values) {
    std::vector<int64_t> deltas(values.size());
    int64_t mid = 0;
    detail::check_infallible(zbra_hip_frame_of_reference_encode(values.data(), values.size(), &mid, deltas.data()));
    return {mid, std::move(deltas)};
}

// frame_of_reference_decode(midpoint: i64, deltas: &[i64]) -> Vec<i64>  compression.rs:77-82
inline std::vector<int64_t> frame_of_reference_decode(int64_t midpoint, const std::vector<int64_t> &deltas) {
    std::vector<int64_t> out(deltas.size());
    detail::check_infallible(zbra_hip_frame_of_reference_decode(midpoint, deltas.data(), deltas.size(), out.data()));
    return out;
}

// zig_zag_encode(values: &[i64]) -> Vec<u64>                            compression.rs:88-98
inline std::vector<uint64_t> zig_zag_encode(const std::vector<int64_t> &values) {
    std::vector<uint64_t> out(values.size());
    detail::check_infallible(zbra_hip_zig_zag_encode(values.data(), values.size(), out.data()));
    return out;
}

// zig_zag_decode(values: &[u64]) -> Vec<i64>                            compression.rs:101-111
inline std::vector<int64_t> zig_zag_decode(const std::vector<uint64_t> &values) {
    std::vector<int64_t> out(values.size());
    detail::check_infallible(zbra_hip_zig_zag_decode(values.data(), values.size(), out.data()));
    return out;
}

// bp64_pack(values: &[u64]) -> Result<Vec<u8>>                          compression.rs:117-178
inline Result<std::vector<uint8_t>> bp64_pack(const std::vector<uint64_t> &values) {
    std::vector<uint8_t> out(values.empty() ? 0 : 1 + 8 * values.size());
    size_t len = 0;
    const int st = zbra_hip_bp64_pack(values.data(), values.size(), out.data(), out.size(), &len);
    if (st != ZBRA_OK) return detail::make_error(st, BinaryError::CompressionError);
    out.resize(len);
    return out;
}

// bp64_unpack(packed: &[u8], count: usize) -> Result<Vec<u64>>          compression.rs:181-233
inline Result<std::vector<uint64_t>> bp64_unpack(const std::vector<uint8_t> &packed, size_t count) {
    std::vector<uint64_t> out(count);
    size_t got = 0;
    const int st = zbra_hip_bp64_unpack(packed.data(), packed.size(), count, out.data(), &got);
    if (st != ZBRA_OK) return detail::make_error(st, BinaryError::DecompressionError);
    out.resize(got);
    return out;
}

// compress_int_array(values: &[i64]) -> Result<Vec<u8>>                 compression.rs:271-292
inline Result<std::vector<uint8_t>> compress_int_array(const std::vector<int64_t> &values) {
    std::vector<uint8_t> out(zbra_hip_encode_bound(values.size()));
    size_t len = 0;
    const int st = zbra_hip_encode_i64(values.data(), values.size(), out.data(), out.size(), &len);
    if (st != ZBRA_OK) return detail::make_error(st, BinaryError::CompressionError);
    out.resize(len);
    return out;
}

// decompress_int_array(data: &[u8], count: usize) -> Result<Vec<i64>>   compression.rs:295-330
inline Result<std::vector<int64_t>> decompress_int_array(const std::vector<uint8_t> &data, size_t count) {
    std::vector<int64_t> out(count);
    size_t got = 0;
    const int st = zbra_hip_decode_i64(data.data(), data.size(), count, out.data(), &got);
    if (st != ZBRA_OK) return detail::make_error(st, BinaryError::DecompressionError);
    out.resize(got);
    return out;
}

}  // namespace compression
}  // namespace zbra
