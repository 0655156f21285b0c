//! `zbra-hip-sys`: the C ABI of `libzbra_hip.so` (header `include/zbra_hip.h`) for Rust, and safe
//! wrappers with the signatures zbra-core's `compression` module has today.
//!
//! The `ffi` module declares every entry point of the header, one for one.  The functions at the
//! crate root take slices and return `Vec`s the way `zbra_core::compression` does
//! (`zbra-core/src/compression.rs:52-330`), so that the bodies there can become one-line calls, and
//! add the batched forms a whole `Block` uses (`zbra-core/src/binary.rs:184-199`).
//!
//! Not compiled in the environment this crate was written in (no Rust toolchain there): the
//! declarations are checked against the header by `tests/test_rust_ffi.py`, which parses both, and
//! the call pattern of the wrappers (allocate, call, truncate) is exercised on a GPU through the same
//! C ABI by `tests/c/test_wrapper_pattern.c`.

#![allow(non_camel_case_types)]
#![allow(clippy::missing_safety_doc)]

use std::fmt;
use std::os::raw::{c_char, c_int, c_void};

/// Raw declarations: `include/zbra_hip.h`, same order, same names.
pub mod ffi {
    use super::*;

    pub const ZBRA_OK: c_int = 0;
    pub const ZBRA_ERR_SHORT_HEADER: c_int = 1;
    pub const ZBRA_ERR_SHORT_PACKED: c_int = 2;
    pub const ZBRA_ERR_TRUNCATED: c_int = 3;
    pub const ZBRA_ERR_CAPACITY: c_int = 4;
    pub const ZBRA_ERR_INVALID_ARG: c_int = 5;
    pub const ZBRA_ERR_NO_DEVICE: c_int = 6;
    pub const ZBRA_ERR_FORMAT: c_int = 7;
    pub const ZBRA_ERR_UNKNOWN_TAG: c_int = 8;
    pub const ZBRA_ERR_NOALLOC_LIMIT: c_int = 9;
    pub const ZBRA_ERR_HIP_BASE: c_int = 100;

    pub const ZBRA_ELEM_I64: u32 = 0;
    pub const ZBRA_ELEM_U32: u32 = 1;

    pub const ZBRA_ROLE_INT_VALUES: u32 = 0;
    pub const ZBRA_ROLE_DOUBLE_BITS: u32 = 1;
    pub const ZBRA_ROLE_BINARY_LENGTHS: u32 = 2;
    pub const ZBRA_ROLE_ARRAY_LENGTHS: u32 = 3;
    pub const ZBRA_ROLE_ENUM_TAGS: u32 = 4;
    pub const ZBRA_ROLE_NESTED_LENGTHS: u32 = 5;

    pub const ZBRA_HIP_NOALLOC_MAX_ARRAYS: usize = 4096;
    pub const ZBRA_HIP_NOALLOC_MAX_TILES: usize = 1 << 20;

    #[repr(C)]
    #[derive(Debug, Clone, Copy, Default)]
    pub struct zbra_dev_result {
        pub status: u32,
        pub width: u32,
        pub out_count: u64,
        pub midpoint: i64,
        pub packed_len: u64,
    }

    #[repr(C)]
    #[derive(Debug, Clone, Copy)]
    pub struct zbra_array_desc {
        pub data: *const u8,
        pub len: u64,
        pub count: u64,
        pub out: *mut c_void,
        pub kind: u32,
        pub reserved: u32,
    }

    #[repr(C)]
    #[derive(Debug, Clone, Copy)]
    pub struct zbra_encode_desc {
        pub values: *const c_void,
        pub n: u64,
        pub out: *mut u8,
        pub cap: u64,
        pub kind: u32,
        pub reserved: u32,
    }

    #[repr(C)]
    #[derive(Debug, Clone, Copy)]
    pub struct zbra_enum_desc {
        pub tags: *const u32,
        pub n: u64,
        pub variant: *mut u32,
        pub rank: *mut u64,
        pub counts: *mut u64,
    }

    #[repr(C)]
    #[derive(Debug, Clone, Copy, Default)]
    pub struct zbra_file_array {
        pub offset: u64,
        pub len: u64,
        pub count: u64,
        pub kind: u32,
        pub role: u32,
        pub block: u32,
        pub reserved: u32,
    }

    #[repr(C)]
    #[derive(Debug, Clone, Copy)]
    pub struct zbra_file_job {
        pub device: i32,
        pub status: i32,
        pub image: *const u8,
        pub len: u64,
        pub arrays: *const zbra_file_array,
        pub n: u64,
        pub outs: *const *mut c_void,
        pub statuses: *mut c_int,
    }

    #[repr(C)]
    #[derive(Debug, Clone, Copy, Default)]
    pub struct zbra_file_info {
        pub schema_offset: u64,
        pub schema_len: u64,
        pub compression_offset: u64,
        pub compression_len: u64,
        pub blocks_offset: u64,
        pub block_count: u32,
        pub reserved: u32,
        pub n_arrays: u64,
        pub total_values: u64,
        pub end_offset: u64,
    }

    extern "C" {
        pub fn zbra_hip_version() -> *const c_char;
        pub fn zbra_hip_strerror(status: c_int) -> *const c_char;
        pub fn zbra_hip_device_count(count: *mut c_int) -> c_int;
        pub fn zbra_hip_set_device(device: c_int) -> c_int;
        pub fn zbra_hip_init() -> c_int;
        pub fn zbra_hip_encode_bound(n: usize) -> usize;
        pub fn zbra_hip_decoded_count(data: *const u8, len: usize, count: usize, out_count: *mut usize) -> c_int;
        pub fn zbra_hip_encode_i64(values: *const i64, n: usize, out: *mut u8, cap: usize, out_len: *mut usize) -> c_int;
        pub fn zbra_hip_encode_u32(values: *const u32, n: usize, out: *mut u8, cap: usize, out_len: *mut usize) -> c_int;
        pub fn zbra_hip_decode_i64(data: *const u8, len: usize, count: usize, out: *mut i64, out_count: *mut usize) -> c_int;
        pub fn zbra_hip_decode_u32(data: *const u8, len: usize, count: usize, out: *mut u32, out_count: *mut usize) -> c_int;
        pub fn zbra_hip_frame_of_reference_encode(values: *const i64, n: usize, midpoint: *mut i64, deltas: *mut i64) -> c_int;
        pub fn zbra_hip_frame_of_reference_decode(midpoint: i64, deltas: *const i64, n: usize, out: *mut i64) -> c_int;
        pub fn zbra_hip_zig_zag_encode(values: *const i64, n: usize, out: *mut u64) -> c_int;
        pub fn zbra_hip_zig_zag_decode(values: *const u64, n: usize, out: *mut i64) -> c_int;
        pub fn zbra_hip_bp64_pack(values: *const u64, n: usize, out: *mut u8, cap: usize, out_len: *mut usize) -> c_int;
        pub fn zbra_hip_bp64_unpack(packed: *const u8, packed_len: usize, count: usize, out: *mut u64, out_count: *mut usize) -> c_int;
        pub fn zbra_hip_decode_dev(d_data: *const u8, len: usize, count: usize, d_out: *mut c_void, kind: u32,
                                   d_result: *mut zbra_dev_result, stream: *mut c_void) -> c_int;
        pub fn zbra_hip_decode_offsets_dev(d_data: *const u8, len: usize, count: usize, d_offsets: *mut u64,
                                           d_result: *mut zbra_dev_result, stream: *mut c_void) -> c_int;
        pub fn zbra_hip_enum_rows_dev(d_tags: *const u32, n: usize, variant_tags: *const u32, n_variants: u32, d_variant: *mut u32,
                                      d_rank: *mut u64, d_counts: *mut u64, d_result: *mut zbra_dev_result, stream: *mut c_void) -> c_int;
        pub fn zbra_hip_decode_offsets_batch_dev(descs: *const zbra_array_desc, nd: usize, d_results: *mut zbra_dev_result,
                                                 stream: *mut c_void) -> c_int;
        pub fn zbra_hip_enum_rows_batch_dev(descs: *const zbra_enum_desc, nd: usize, variant_tags: *const u32, n_variants: u32,
                                            d_results: *mut zbra_dev_result, stream: *mut c_void) -> c_int;
        pub fn zbra_hip_decode_batch_dev(descs: *const zbra_array_desc, nd: usize, d_results: *mut zbra_dev_result,
                                         stream: *mut c_void) -> c_int;
        pub fn zbra_hip_decode_batch_dev_noalloc(descs: *const zbra_array_desc, nd: usize, d_results: *mut zbra_dev_result,
                                                 stream: *mut c_void) -> c_int;
        pub fn zbra_hip_encode_dev(d_values: *const c_void, n: usize, kind: u32, d_out: *mut u8, cap: usize,
                                   d_result: *mut zbra_dev_result, stream: *mut c_void) -> c_int;
        pub fn zbra_hip_encode_batch_dev(descs: *const zbra_encode_desc, nd: usize, d_results: *mut zbra_dev_result,
                                         stream: *mut c_void) -> c_int;
        pub fn zbra_hip_encode_frames_dev(descs: *const zbra_encode_desc, nd: usize, d_out: *mut u8, cap: usize, d_offsets: *mut u64,
                                          d_results: *mut zbra_dev_result, stream: *mut c_void) -> c_int;
        pub fn zbra_hip_file_index(image: *const u8, len: usize, arrays: *mut zbra_file_array, cap: usize, info: *mut zbra_file_info,
                                   err_offset: *mut usize) -> c_int;
        pub fn zbra_hip_file_decode(image: *const u8, len: usize, arrays: *const zbra_file_array, n: usize, outs: *const *mut c_void,
                                    statuses: *mut c_int) -> c_int;
        pub fn zbra_hip_file_decode_multi(jobs: *mut zbra_file_job, n_jobs: usize) -> c_int;
        pub fn zbra_hip_decode_batch(descs: *const zbra_array_desc, nd: usize, statuses: *mut c_int) -> c_int;
        pub fn zbra_hip_encode_frames(descs: *const zbra_encode_desc, nd: usize, out: *mut u8, cap: usize, offsets: *mut u64,
                                      statuses: *mut c_int) -> c_int;
    }
}

/// What a call can fail with.  `ShortHeader` and `ShortPacked` display the exact messages of
/// `BinaryError::DecompressionError` at `zbra-core/src/compression.rs:301-303, 313-315`.
#[derive(Debug, Clone, PartialEq, Eq)]
pub enum Error {
    ShortHeader,
    ShortPacked,
    /// Packed payload shorter than `count * width` bits: the reference underflows a `u32` there
    /// (`compression.rs:229`); this library reports it.
    Truncated,
    Capacity,
    InvalidArg,
    NoDevice,
    /// A `.zbra` image that `BinaryFile::read_from` would reject, with the offset where the walk stopped.
    Format { offset: usize },
    UnknownTag,
    NoAllocLimit,
    Hip(i32),
}

impl Error {
    pub fn from_status(status: c_int) -> Error {
        match status {
            ffi::ZBRA_ERR_SHORT_HEADER => Error::ShortHeader,
            ffi::ZBRA_ERR_SHORT_PACKED => Error::ShortPacked,
            ffi::ZBRA_ERR_TRUNCATED => Error::Truncated,
            ffi::ZBRA_ERR_CAPACITY => Error::Capacity,
            ffi::ZBRA_ERR_INVALID_ARG => Error::InvalidArg,
            ffi::ZBRA_ERR_NO_DEVICE => Error::NoDevice,
            ffi::ZBRA_ERR_FORMAT => Error::Format { offset: 0 },
            ffi::ZBRA_ERR_UNKNOWN_TAG => Error::UnknownTag,
            ffi::ZBRA_ERR_NOALLOC_LIMIT => Error::NoAllocLimit,
            s => Error::Hip(s - ffi::ZBRA_ERR_HIP_BASE),
        }
    }

    /// True for the two errors the reference's `decompress_int_array` raises itself.
    pub fn is_reference_error(&self) -> bool {
        matches!(self, Error::ShortHeader | Error::ShortPacked)
    }
}

impl fmt::Display for Error {
    fn fmt(&self, f: &mut fmt::Formatter<'_>) -> fmt::Result {
        match self {
            Error::ShortHeader => f.write_str("Invalid compressed data length"),
            Error::ShortPacked => f.write_str("Insufficient data for packed array"),
            Error::Truncated => f.write_str("Packed payload shorter than count * width bits"),
            Error::Capacity => f.write_str("Output buffer too small"),
            Error::InvalidArg => f.write_str("Invalid argument"),
            Error::NoDevice => f.write_str("No usable HIP device"),
            Error::Format { offset } => write!(f, "Invalid .zbra image (magic number, tag or length) at offset {}", offset),
            Error::UnknownTag => f.write_str("enum tag carried by no variant"),
            Error::NoAllocLimit => f.write_str("Submission exceeds the fixed scratch of the no-allocation entry point"),
            Error::Hip(e) => write!(f, "HIP error {}", e),
        }
    }
}

impl std::error::Error for Error {}

pub type Result<T> = std::result::Result<T, Error>;

fn check(status: c_int) -> Result<()> {
    if status == ffi::ZBRA_OK { Ok(()) } else { Err(Error::from_status(status)) }
}

/// Creates the per-device context ahead of the first call (optional).
pub fn init() -> Result<()> {
    check(unsafe { ffi::zbra_hip_init() })
}

pub fn set_device(device: i32) -> Result<()> {
    check(unsafe { ffi::zbra_hip_set_device(device) })
}

pub fn device_count() -> Result<i32> {
    let mut n: c_int = 0;
    check(unsafe { ffi::zbra_hip_device_count(&mut n) })?;
    Ok(n)
}

// ---- zbra_core::compression, same signatures ------------------------------------------------

/// `compress_int_array(values: &[i64]) -> Result<Vec<u8>>` — `compression.rs:271-292`.
pub fn compress_int_array(values: &[i64]) -> Result<Vec<u8>> {
    let mut out = vec![0u8; unsafe { ffi::zbra_hip_encode_bound(values.len()) }];
    let mut len = 0usize;
    check(unsafe { ffi::zbra_hip_encode_i64(values.as_ptr(), values.len(), out.as_mut_ptr(), out.len(), &mut len) })?;
    out.truncate(len);
    Ok(out)
}

/// `write_u32_array_compressed`'s widen + compress — `binary.rs:666-669`, without the `Vec<i64>`.
pub fn compress_u32_array(values: &[u32]) -> Result<Vec<u8>> {
    let mut out = vec![0u8; unsafe { ffi::zbra_hip_encode_bound(values.len()) }];
    let mut len = 0usize;
    check(unsafe { ffi::zbra_hip_encode_u32(values.as_ptr(), values.len(), out.as_mut_ptr(), out.len(), &mut len) })?;
    out.truncate(len);
    Ok(out)
}

/// `decompress_int_array(data: &[u8], count: usize) -> Result<Vec<i64>>` — `compression.rs:295-330`.
/// The result has 0 elements for empty data and for an empty packed slice, whatever `count` says
/// (`compression.rs:296-298, 182-184`).
pub fn decompress_int_array(data: &[u8], count: usize) -> Result<Vec<i64>> {
    let mut out = vec![0i64; count];
    let mut got = 0usize;
    check(unsafe { ffi::zbra_hip_decode_i64(data.as_ptr(), data.len(), count, out.as_mut_ptr(), &mut got) })?;
    out.truncate(got);
    Ok(out)
}

/// `read_u32_array_compressed`'s decompress + `as u32` — `binary.rs:671-674`, without the `Vec<i64>`.
pub fn decompress_u32_array(data: &[u8], count: usize) -> Result<Vec<u32>> {
    let mut out = vec![0u32; count];
    let mut got = 0usize;
    check(unsafe { ffi::zbra_hip_decode_u32(data.as_ptr(), data.len(), count, out.as_mut_ptr(), &mut got) })?;
    out.truncate(got);
    Ok(out)
}

/// Length of the `Vec` `decompress_int_array(data, count)` returns, from the header alone (host only).
pub fn decoded_count(data: &[u8], count: usize) -> Result<usize> {
    let mut got = 0usize;
    check(unsafe { ffi::zbra_hip_decoded_count(data.as_ptr(), data.len(), count, &mut got) })?;
    Ok(got)
}

/// `frame_of_reference_encode(values) -> (i64, Vec<i64>)` — `compression.rs:52-74`.
pub fn frame_of_reference_encode(values: &[i64]) -> Result<(i64, Vec<i64>)> {
    let mut deltas = vec![0i64; values.len()];
    let mut midpoint = 0i64;
    check(unsafe { ffi::zbra_hip_frame_of_reference_encode(values.as_ptr(), values.len(), &mut midpoint, deltas.as_mut_ptr()) })?;
    Ok((midpoint, deltas))
}

/// `frame_of_reference_decode(midpoint, deltas) -> Vec<i64>` — `compression.rs:77-82`.
pub fn frame_of_reference_decode(midpoint: i64, deltas: &[i64]) -> Result<Vec<i64>> {
    let mut out = vec![0i64; deltas.len()];
    check(unsafe { ffi::zbra_hip_frame_of_reference_decode(midpoint, deltas.as_ptr(), deltas.len(), out.as_mut_ptr()) })?;
    Ok(out)
}

/// `zig_zag_encode(values) -> Vec<u64>` — `compression.rs:88-98`.
pub fn zig_zag_encode(values: &[i64]) -> Result<Vec<u64>> {
    let mut out = vec![0u64; values.len()];
    check(unsafe { ffi::zbra_hip_zig_zag_encode(values.as_ptr(), values.len(), out.as_mut_ptr()) })?;
    Ok(out)
}

/// `zig_zag_decode(values) -> Vec<i64>` — `compression.rs:101-111`.
pub fn zig_zag_decode(values: &[u64]) -> Result<Vec<i64>> {
    let mut out = vec![0i64; values.len()];
    check(unsafe { ffi::zbra_hip_zig_zag_decode(values.as_ptr(), values.len(), out.as_mut_ptr()) })?;
    Ok(out)
}

/// `bp64_pack(values) -> Result<Vec<u8>>` — `compression.rs:117-178`.
pub fn bp64_pack(values: &[u64]) -> Result<Vec<u8>> {
    let mut out = vec![0u8; if values.is_empty() { 0 } else { 1 + 8 * values.len() }];
    let mut len = 0usize;
    check(unsafe { ffi::zbra_hip_bp64_pack(values.as_ptr(), values.len(), out.as_mut_ptr(), out.len(), &mut len) })?;
    out.truncate(len);
    Ok(out)
}

/// `bp64_unpack(packed, count) -> Result<Vec<u64>>` — `compression.rs:181-233`.
pub fn bp64_unpack(packed: &[u8], count: usize) -> Result<Vec<u64>> {
    let mut out = vec![0u64; count];
    let mut got = 0usize;
    check(unsafe { ffi::zbra_hip_bp64_unpack(packed.as_ptr(), packed.len(), count, out.as_mut_ptr(), &mut got) })?;
    out.truncate(got);
    Ok(out)
}

// ---- batched forms: every int array of a Block in one call ----------------------------------

/// One int array handed to [`encode_frames`]: what `write_int_array_compressed` /
/// `write_int_array_usize_compressed` (I64: `usize` and `f64` bit patterns have the same layout on
/// 64-bit targets) or `write_u32_array_compressed` (U32) is called with — `binary.rs:639-669`.
#[derive(Debug, Clone, Copy)]
pub enum IntArrayRef<'a> {
    I64(&'a [i64]),
    U32(&'a [u32]),
}

/// Frames `u32 count | u32 clen | E` of `arrays`, back to back, and the `arrays.len() + 1` offsets of
/// the frames: what `Column::write_to` emits for each of them (`binary.rs:293-347`), in one
/// submission.
pub fn encode_frames(arrays: &[IntArrayRef<'_>]) -> Result<(Vec<u8>, Vec<u64>)> {
    let mut descs = Vec::with_capacity(arrays.len());
    let mut cap = 0usize;
    for a in arrays {
        let (values, n, kind) = match a {
            IntArrayRef::I64(v) => (v.as_ptr() as *const c_void, v.len(), ffi::ZBRA_ELEM_I64),
            IntArrayRef::U32(v) => (v.as_ptr() as *const c_void, v.len(), ffi::ZBRA_ELEM_U32),
        };
        cap += 8 + unsafe { ffi::zbra_hip_encode_bound(n) };
        descs.push(ffi::zbra_encode_desc { values, n: n as u64, out: std::ptr::null_mut(), cap: 0, kind, reserved: 0 });
    }
    let mut out = vec![0u8; cap];
    let mut offsets = vec![0u64; arrays.len() + 1];
    check(unsafe { ffi::zbra_hip_encode_frames(descs.as_ptr(), descs.len(), out.as_mut_ptr(), out.len(), offsets.as_mut_ptr(), std::ptr::null_mut()) })?;
    out.truncate(offsets[arrays.len()] as usize);
    Ok((out, offsets))
}

/// One compressed array handed to [`decode_batch`]: the `compressed` bytes `read_int_array_compressed`
/// has just `read_exact`ed, the `len` of its frame, and whether the caller narrows to `u32`
/// (`binary.rs:648-654, 671-674`).
#[derive(Debug, Clone, Copy)]
pub struct CompressedArray<'a> {
    pub data: &'a [u8],
    pub count: usize,
    pub u32_out: bool,
}

/// A decoded array of [`decode_batch`].
#[derive(Debug, Clone, PartialEq, Eq)]
pub enum IntArray {
    I64(Vec<i64>),
    U32(Vec<u32>),
}

/// `decompress_int_array` / `read_u32_array_compressed` for many arrays in one submission: uploads,
/// the batched decode and the copies back overlap.  Fails with the first array's error, like the
/// sequential reads of `Column::read_from` (`binary.rs:359-441`).
pub fn decode_batch(arrays: &[CompressedArray<'_>]) -> Result<Vec<IntArray>> {
    let mut outs: Vec<IntArray> = Vec::with_capacity(arrays.len());
    let mut descs = Vec::with_capacity(arrays.len());
    for a in arrays {
        // the library writes nothing to an array that yields no values: size the result up front
        let n = decoded_count(a.data, a.count)?;
        let (out, ptr, kind) = if a.u32_out {
            let mut v = vec![0u32; n];
            let p = v.as_mut_ptr() as *mut c_void;
            (IntArray::U32(v), p, ffi::ZBRA_ELEM_U32)
        } else {
            let mut v = vec![0i64; n];
            let p = v.as_mut_ptr() as *mut c_void;
            (IntArray::I64(v), p, ffi::ZBRA_ELEM_I64)
        };
        outs.push(out);  // moving the Vec does not move its heap buffer
        descs.push(ffi::zbra_array_desc { data: a.data.as_ptr(), len: a.data.len() as u64, count: n as u64, out: ptr, kind, reserved: 0 });
    }
    check(unsafe { ffi::zbra_hip_decode_batch(descs.as_ptr(), descs.len(), std::ptr::null_mut()) })?;
    Ok(outs)
}

/// Every int array of a `.zbra` image, found without decoding (`BinaryFile::read_from`'s control
/// flow, `binary.rs:139-174, 195-199, 240-270, 359-441`).  Host only.
pub fn file_index(image: &[u8]) -> Result<(ffi::zbra_file_info, Vec<ffi::zbra_file_array>)> {
    let mut info = ffi::zbra_file_info::default();
    let mut err = 0usize;
    let st = unsafe { ffi::zbra_hip_file_index(image.as_ptr(), image.len(), std::ptr::null_mut(), 0, &mut info, &mut err) };
    if st == ffi::ZBRA_ERR_FORMAT {
        return Err(Error::Format { offset: err });
    }
    check(st)?;
    let mut arrays = vec![ffi::zbra_file_array::default(); info.n_arrays as usize];
    check(unsafe { ffi::zbra_hip_file_index(image.as_ptr(), image.len(), arrays.as_mut_ptr(), arrays.len(), &mut info, &mut err) })?;
    Ok((info, arrays))
}

/// Decodes the indexed arrays of an image (the whole-file read path): one pipelined pass.
pub fn file_decode(image: &[u8], arrays: &[ffi::zbra_file_array]) -> Result<Vec<IntArray>> {
    let mut outs: Vec<IntArray> = Vec::with_capacity(arrays.len());
    let mut ptrs: Vec<*mut c_void> = Vec::with_capacity(arrays.len());
    let mut sized = arrays.to_vec();
    for a in sized.iter_mut() {
        let e = &image[a.offset as usize..(a.offset + a.len) as usize];
        let n = decoded_count(e, a.count as usize)?;
        a.count = n as u64;
        if a.kind == ffi::ZBRA_ELEM_U32 {
            let mut v = vec![0u32; n];
            ptrs.push(v.as_mut_ptr() as *mut c_void);
            outs.push(IntArray::U32(v));
        } else {
            let mut v = vec![0i64; n];
            ptrs.push(v.as_mut_ptr() as *mut c_void);
            outs.push(IntArray::I64(v));
        }
    }
    check(unsafe { ffi::zbra_hip_file_decode(image.as_ptr(), image.len(), sized.as_ptr(), sized.len(), ptrs.as_ptr(), std::ptr::null_mut()) })?;
    Ok(outs)
}
