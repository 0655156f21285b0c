/*
 * zbra_hip.h — C ABI of libzbra_hip.so: Zbra's integer column codec
 * (frame-of-reference by median, zig-zag, fixed-width LSB-first bit-packing
 * with a raw 8-byte fallback at width >= 32) as HIP kernels for gfx950.
 *
 * The reference has no FFI seam for this path: the seam is the eight public
 * functions of zbra-core/src/compression.rs and the six private framing
 * helpers of zbra-core/src/binary.rs:639-674 that call them.  Every entry
 * point below names the reference function it replaces.
 *
 * Conventions
 *   - plain pointers and sizes only; caller allocates every output;
 *   - return value: 0 = ZBRA_OK, otherwise a zbra_status; values >= 100 carry
 *     a hipError_t (status - 100);
 *   - the reference's "empty" conventions are results, not errors:
 *       compress_int_array([])            -> 0 bytes          compression.rs:272-274
 *       decompress_int_array(b"", count)  -> 0 values         compression.rs:296-298
 *       bp64_unpack(b"", count)           -> 0 values         compression.rs:182-184
 *   - functions without a _dev suffix take HOST pointers, run on the current
 *     HIP device and return after the result is in host memory;
 *   - functions with a _dev suffix take DEVICE pointers, enqueue work on the
 *     given stream (a hipStream_t passed as void*, NULL = the null stream)
 *     and return without synchronising; results that are only known on the
 *     device (status, lengths) land in a zbra_dev_result in device memory;
 *   - all entry points use the calling thread's current HIP device; calls on
 *     different streams or devices may run concurrently from different
 *     threads.  Host-pointer calls from several threads on ONE device run
 *     two at a time (each on its own streams, pinned buffers and device
 *     buffers; ZBRA_HIP_HOST_LANES=1..4 changes the number) and queue
 *     beyond that.  There is no CPU fallback: without a usable device every
 *     compute entry point fails with ZBRA_ERR_NO_DEVICE.
 */
#ifndef ZBRA_HIP_H
#define ZBRA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum zbra_status {
    ZBRA_OK = 0,
    /* BinaryError::DecompressionError("Invalid compressed data length")     compression.rs:300-304 */
    ZBRA_ERR_SHORT_HEADER = 1,
    /* BinaryError::DecompressionError("Insufficient data for packed array") compression.rs:312-316 */
    ZBRA_ERR_SHORT_PACKED = 2,
    /* Packed payload shorter than ceil(count*width/8) bytes.  The reference underflows a u32
     * there (compression.rs:229: panic in debug, garbage in release); this library reports it. */
    ZBRA_ERR_TRUNCATED = 3,
    ZBRA_ERR_CAPACITY = 4,    /* caller's output buffer is too small */
    ZBRA_ERR_INVALID_ARG = 5, /* NULL pointer with a non-zero size, count above 2^32-1, ... */
    ZBRA_ERR_NO_DEVICE = 6,   /* no HIP device / HIP runtime unusable */
    /* .zbra image: BinaryError::InvalidMagicNumber / Invalid{Table,Column,Default,Encoding,
     * IntEncoding,BinaryEncoding}Tag / unexpected end of data   binary.rs:143-145, 268, 439, 500,
     * 526, 551, 570 */
    ZBRA_ERR_FORMAT = 7,
    /* a row's enum tag is carried by no variant: SchemaError::UnsupportedType("enum tag N"),
     * zbra-core/src/striped.rs:352-358 */
    ZBRA_ERR_UNKNOWN_TAG = 8,
    /* zbra_hip_decode_batch_dev_noalloc: more arrays or more tiles than its fixed scratch holds */
    ZBRA_ERR_NOALLOC_LIMIT = 9,
    ZBRA_ERR_HIP_BASE = 100   /* status - 100 = hipError_t */
} zbra_status;

/* What kind of values an int array carries (binary.rs:639-674). */
typedef enum zbra_elem_kind {
    ZBRA_ELEM_I64 = 0, /* Vec<i64> / Vec<usize> / f64 bit patterns  binary.rs:639-664, 299-300 */
    ZBRA_ELEM_U32 = 1  /* Vec<u32> enum tags: widened `as i64` on write, narrowed `as u32` on read
                          binary.rs:666-674 */
} zbra_elem_kind;

/* Result words a _dev call leaves in DEVICE memory (32 bytes). */
typedef struct zbra_dev_result {
    uint32_t status;    /* zbra_status of this array */
    uint32_t width;     /* bit width byte of the encoded array (0 if none) */
    uint64_t out_count; /* decode: values produced; encode: bytes produced */
    int64_t midpoint;   /* frame-of-reference midpoint */
    uint64_t packed_len;/* value of the u32 length field (1 + payload bytes) */
} zbra_dev_result;

/* One encoded int array of a batch: where its `E` bytes are (the `clen` bytes that follow the
 * `u32 count | u32 clen` frame of binary.rs:639-645), and where its values go. */
typedef struct zbra_array_desc {
    const uint8_t *data; /* first byte of E (the midpoint) */
    uint64_t len;        /* clen */
    uint64_t count;      /* count from the frame */
    void *out;           /* count elements of `kind` */
    uint32_t kind;       /* zbra_elem_kind */
    uint32_t reserved;
} zbra_array_desc;

/* One int array of an encode batch: values in, E out. */
typedef struct zbra_encode_desc {
    const void *values; /* n elements of `kind` */
    uint64_t n;
    uint8_t *out;       /* receives E */
    uint64_t cap;       /* bytes available at `out`; zbra_hip_encode_bound(n) always suffices */
    uint32_t kind;      /* zbra_elem_kind */
    uint32_t reserved;
} zbra_encode_desc;

/* The tag array of one Enum column of a batch (zbra_hip_enum_rows_batch_dev): all pointers are
 * DEVICE pointers. */
typedef struct zbra_enum_desc {
    const uint32_t *tags; /* n tags: Column::Enum.tags            striped.rs:47-52 */
    uint64_t n;
    uint32_t *variant;    /* receives n variant indices */
    uint64_t *rank;       /* receives n positions in the variants' columns */
    uint64_t *counts;     /* receives n_variants row counts */
} zbra_enum_desc;

/* Which striped vector an int array of a .zbra image carries (the six call sites of
 * binary.rs:366-434). */
typedef enum zbra_array_role {
    ZBRA_ROLE_INT_VALUES = 0,     /* Column::Int.values                 binary.rs:369 */
    ZBRA_ROLE_DOUBLE_BITS = 1,    /* Column::Double.values as i64 bits  binary.rs:378-382 */
    ZBRA_ROLE_BINARY_LENGTHS = 2, /* Column::Binary.lengths             binary.rs:388 */
    ZBRA_ROLE_ARRAY_LENGTHS = 3,  /* Column::Array.lengths              binary.rs:399 */
    ZBRA_ROLE_ENUM_TAGS = 4,      /* Column::Enum.tags (u32)            binary.rs:418 */
    ZBRA_ROLE_NESTED_LENGTHS = 5  /* Column::Nested.lengths             binary.rs:431 */
} zbra_array_role;

/* One int array found in a .zbra image by zbra_hip_file_index, in file order. */
typedef struct zbra_file_array {
    uint64_t offset; /* of E (the byte after `u32 count | u32 clen`) in the image */
    uint64_t len;    /* clen */
    uint64_t count;  /* count from the frame */
    uint32_t kind;   /* zbra_elem_kind the reader narrows to: U32 for enum tags, else I64 */
    uint32_t role;   /* zbra_array_role */
    uint32_t block;  /* index of the Block it belongs to */
    uint32_t reserved;
} zbra_file_array;

/* Summary of a .zbra image. */
typedef struct zbra_file_info {
    uint64_t schema_offset, schema_len;           /* JSON TableSchema, stored as is  binary.rs:108-113 */
    uint64_t compression_offset, compression_len; /* JSON CompressionConfig          binary.rs:116-121 */
    uint64_t blocks_offset;                       /* of the u32 block_count          binary.rs:124 */
    uint32_t block_count;
    uint32_t reserved;
    uint64_t n_arrays;                            /* int arrays in all Blocks */
    uint64_t total_values;                        /* sum of their counts */
    uint64_t end_offset;                          /* first byte after the last Block */
} zbra_file_info;

/* ---- library / device ------------------------------------------------------------------- */

const char *zbra_hip_version(void);
const char *zbra_hip_strerror(int status);
int zbra_hip_device_count(int *count);
int zbra_hip_set_device(int device);
/* Creates the per-device context (workspaces) ahead of the first call. Optional. */
int zbra_hip_init(void);

/* 13 + 8*n for n > 0, else 0: capacity that compress_int_array's result never exceeds. */
size_t zbra_hip_encode_bound(size_t n);

/* Length of the Vec that decompress_int_array(data, count) returns, and its status, from the header
 * alone (compression.rs:296-321, bp64_unpack :182-189): 0 values for empty data and for
 * packed_len == 0 whatever `count` says, else `count`; the same status a decode call gives.  Pure
 * host code on HOST bytes, no device needed.  The batched calls below write nothing to an output
 * whose array yields 0 values: callers size the result with this. */
int zbra_hip_decoded_count(const uint8_t *data, size_t len, size_t count, size_t *out_count);

/* ---- full pipeline, host pointers --------------------------------------------------------- */

/* compress_int_array(values: &[i64]) -> Vec<u8>                      compression.rs:271-292 */
int zbra_hip_encode_i64(const int64_t *values, size_t n, uint8_t *out, size_t cap, size_t *out_len);
/* write_u32_array_compressed's widen + compress_int_array            binary.rs:666-669 */
int zbra_hip_encode_u32(const uint32_t *values, size_t n, uint8_t *out, size_t cap, size_t *out_len);
/* decompress_int_array(data: &[u8], count) -> Vec<i64>               compression.rs:295-330
 * `out` holds `count` elements; *out_count = length of the Vec the reference returns. */
int zbra_hip_decode_i64(const uint8_t *data, size_t len, size_t count, int64_t *out, size_t *out_count);
/* read_u32_array_compressed's decompress + narrow `as u32`            binary.rs:671-674 */
int zbra_hip_decode_u32(const uint8_t *data, size_t len, size_t count, uint32_t *out, size_t *out_count);

/* ---- stage functions, host pointers ------------------------------------------------------- */

/* frame_of_reference_encode(values) -> (midpoint, deltas)            compression.rs:52-74 */
int zbra_hip_frame_of_reference_encode(const int64_t *values, size_t n, int64_t *midpoint, int64_t *deltas);
/* frame_of_reference_decode(midpoint, deltas) -> values              compression.rs:77-82 */
int zbra_hip_frame_of_reference_decode(int64_t midpoint, const int64_t *deltas, size_t n, int64_t *out);
/* zig_zag_encode(values) -> Vec<u64>                                 compression.rs:88-98 */
int zbra_hip_zig_zag_encode(const int64_t *values, size_t n, uint64_t *out);
/* zig_zag_decode(values) -> Vec<i64>                                 compression.rs:101-111 */
int zbra_hip_zig_zag_decode(const uint64_t *values, size_t n, int64_t *out);
/* bp64_pack(values) -> Vec<u8>; cap >= 1 + 8*n always suffices       compression.rs:117-178 */
int zbra_hip_bp64_pack(const uint64_t *values, size_t n, uint8_t *out, size_t cap, size_t *out_len);
/* bp64_unpack(packed, count) -> Vec<u64>                             compression.rs:181-233 */
int zbra_hip_bp64_unpack(const uint8_t *packed, size_t packed_len, size_t count, uint64_t *out, size_t *out_count);

/* ---- device-resident, stream-ordered (the measured path) --------------------------------- */

/* decompress_int_array on bytes already in HBM.  d_data: `len` bytes of E; d_out: `count`
 * elements of `kind`; d_result (may be NULL): status, width, out_count, midpoint.
 * `len` and `count` are host values because the frame that carries them (binary.rs:648-650) is
 * read by the host; everything inside E is parsed on the device.     compression.rs:295-330 */
int zbra_hip_decode_dev(const uint8_t *d_data, size_t len, size_t count, void *d_out, uint32_t kind,
                        zbra_dev_result *d_result, void *stream);

/* read_int_array_usize_compressed (binary.rs:661-664) fused with the running sum every consumer of
 * a lengths array computes next (`offset += length`, striped.rs:523-583): d_offsets receives
 * out_count + 1 words, d_offsets[0] = 0 and d_offsets[i + 1] = d_offsets[i] + lengths[i]
 * (wrapping, as usize adds), where out_count is what decompress_int_array would return (0 for
 * empty or rejected data: then only d_offsets[0] is written).  The lengths themselves are never
 * stored: two reads of the packed bytes, one write of 8 bytes per value. */
int zbra_hip_decode_offsets_dev(const uint8_t *d_data, size_t len, size_t count, uint64_t *d_offsets, zbra_dev_result *d_result,
                                void *stream);

/* The row loop of Column::from_values for an Enum column (zbra-core/src/striped.rs:340-387), from
 * the tag column alone: for every row i, d_variant[i] = index of the FIRST variant whose tag equals
 * d_tags[i] (striped.rs:349) and d_rank[i] = the row's position in that variant's column, i.e. the
 * number of earlier rows of the same variant (variant_data[v].push, striped.rs:351); d_counts[v] =
 * rows of variant v.  With these a payload column in HBM is scattered to / gathered from the
 * variant columns by index.  variant_tags: n_variants (<= 32) tags in HOST memory (the schema).
 * A tag that no variant carries: d_result->status = ZBRA_ERR_UNKNOWN_TAG, out_count = the first
 * such row, midpoint = its tag (striped.rs:352-358), outputs undefined; else ZBRA_OK, out_count = n.
 * Column::to_values (striped.rs:617-657) needs no counterpart: it reads variant column v at the
 * row index itself (striped.rs:631-632). */
int zbra_hip_enum_rows_dev(const uint32_t *d_tags, size_t n, const uint32_t *variant_tags, uint32_t n_variants,
                           uint32_t *d_variant, uint64_t *d_rank, uint64_t *d_counts, zbra_dev_result *d_result, void *stream);

/* zbra_hip_decode_offsets_dev for `nd` lengths arrays in one submission: the Binary, Array and
 * Nested lengths of one Block or of many (binary.rs:388, 399, 431 read them one after another).
 * `descs` is a HOST array; descs[i].data / .len / .count as in zbra_hip_decode_batch_dev,
 * descs[i].out = DEVICE pointer to count + 1 words for that array's offsets (never NULL: an empty or
 * rejected array still gets its offsets[0] = 0), descs[i].kind = ZBRA_ELEM_I64.  d_results (may be
 * NULL) receives nd entries.  The arrays are independent: each running sum starts at 0. */
int zbra_hip_decode_offsets_batch_dev(const zbra_array_desc *descs, size_t nd, zbra_dev_result *d_results, void *stream);

/* zbra_hip_enum_rows_dev for the tag arrays of `nd` Enum columns that share one variant list (the
 * same column in many Blocks: Column::from_values runs once per Block, striped.rs:340-387), in one
 * submission.  `descs` is a HOST array of DEVICE pointers; array i's variant / rank / counts are as
 * in zbra_hip_enum_rows_dev, each array ranked on its own.  d_results (may be NULL) receives nd
 * entries: an array with a tag that no variant carries gets ZBRA_ERR_UNKNOWN_TAG (out_count = its
 * first such row, midpoint = the tag) and undefined outputs, the other arrays are unaffected. */
int zbra_hip_enum_rows_batch_dev(const zbra_enum_desc *descs, size_t nd, const uint32_t *variant_tags, uint32_t n_variants,
                                 zbra_dev_result *d_results, void *stream);

/* The same for `nd` independent arrays in one submission (all arrays of a Block, or of many
 * Blocks: binary.rs:359-441 decodes them one after another).  `descs` is a HOST array whose
 * data/out members are DEVICE pointers; d_results (may be NULL) receives nd entries. */
int zbra_hip_decode_batch_dev(const zbra_array_desc *descs, size_t nd, zbra_dev_result *d_results, void *stream);

/* zbra_hip_decode_batch_dev for callers that cannot allocate: code under HIP graph capture, tight
 * serving loops, harnesses that forbid hipMalloc.  It has no counterpart in the reference.  Same
 * arguments and same results as zbra_hip_decode_batch_dev; always the default tile kernel (never the
 * warm-up route that zbra_hip_decode_batch_dev picks for narrow submissions).  The decode plans, the
 * tiles-per-array scan and the tile -> array table live in a fixed array of device memory inside
 * the library (about 4.2 MiB on every GPU the library runs on, held until the process ends), and
 * the descriptors travel in kernel arguments: the call is kernel launches only, no hipMalloc, no
 * hipFree, no copy, no pinned staging.
 * Limits: nd <= ZBRA_HIP_NOALLOC_MAX_ARRAYS, and the submission's tiles, sum over the arrays of
 * ceil(count / 2048) for i64 outputs and ceil(count / 4096) for u32 outputs, at most
 * ZBRA_HIP_NOALLOC_MAX_TILES (2^31 int64 values in full tiles); beyond either the call does nothing
 * and returns ZBRA_ERR_NOALLOC_LIMIT.
 * Every call on a GPU uses that GPU's one array: calls whose work may overlap on one GPU (different
 * streams, different threads) race; calls on one stream are ordered and safe.  The per-device
 * set-up (context and the array's address on that device) runs in zbra_hip_init() or in any earlier
 * call on the device; when this call is the first one on a device it runs the set-up itself, which
 * creates a stream: call zbra_hip_init() beforehand where even that is not allowed (under graph
 * capture it is not: the captured call must not be the device's first). */
#define ZBRA_HIP_NOALLOC_MAX_ARRAYS 4096
#define ZBRA_HIP_NOALLOC_MAX_TILES (1u << 20)
int zbra_hip_decode_batch_dev_noalloc(const zbra_array_desc *descs, size_t nd, zbra_dev_result *d_results, void *stream);

/* compress_int_array on values already in HBM.  d_out must hold zbra_hip_encode_bound(n) bytes;
 * the encoded length is only known on the device: d_result->out_count (d_result must not be
 * NULL).                                                             compression.rs:271-292 */
int zbra_hip_encode_dev(const void *d_values, size_t n, uint32_t kind, uint8_t *d_out, size_t cap,
                        zbra_dev_result *d_result, void *stream);

/* The same for `nd` independent arrays in one submission (Column::write_to encodes the arrays of
 * a Block one after another, binary.rs:275-356).  `descs` is a HOST array of DEVICE pointers;
 * d_results receives nd entries (status, width, midpoint, out_count = bytes of E). */
int zbra_hip_encode_batch_dev(const zbra_encode_desc *descs, size_t nd, zbra_dev_result *d_results, void *stream);

/* write_int_array_compressed / write_u32_array_compressed for `nd` arrays (binary.rs:639-645,
 * 666-669): array i becomes the frame `u32 count | u32 clen | E`, and the frames are laid out back
 * to back in d_out in the order of `descs`, as they sit in a file when nothing else is written
 * between them.  The layout happens on the device once the widths are known: d_offsets receives
 * nd + 1 words, frame i = [d_offsets[i], d_offsets[i+1]).  descs[i].out and .cap are ignored.
 * Frames that would pass `cap` are not written and get ZBRA_ERR_CAPACITY in d_results (may be
 * NULL); cap = sum of 8 + zbra_hip_encode_bound(n_i) always suffices.  An empty array is the
 * 8-byte frame count = 0, clen = 0.  One call lays out at most 2^20 frames (ZBRA_ERR_INVALID_ARG
 * beyond; the other batched calls take tables of any length and work through them in pieces). */
int zbra_hip_encode_frames_dev(const zbra_encode_desc *descs, size_t nd, uint8_t *d_out, size_t cap, uint64_t *d_offsets,
                               zbra_dev_result *d_results, void *stream);

/* ---- .zbra images ------------------------------------------------------------------------ */

/* Walks a .zbra image the way BinaryFile::read_from does (binary.rs:139-174, 195-199, 240-270,
 * 359-441) without decoding anything: every int array is length-prefixed and byte blobs are
 * skipped by their compressed size (binary.rs:694-697).  Pure host code, no device needed.
 * Writes up to `cap` entries to `arrays` (may be NULL with cap 0 to count only); info->n_arrays
 * is the number found.  ZBRA_ERR_FORMAT: bad magic, invalid tag or truncated image (*err_offset,
 * if not NULL, gets the offset where the walk stopped). */
int zbra_hip_file_index(const uint8_t *image, size_t len, zbra_file_array *arrays, size_t cap, zbra_file_info *info,
                        size_t *err_offset);

/* Decodes int arrays of a .zbra image that is in HOST memory into HOST buffers: the whole-file
 * read path.  arrays[i] (from zbra_hip_file_index) goes to outs[i], which holds arrays[i].count
 * elements of arrays[i].kind.  Copies (pinned hipMemcpyAsync) and kernels of neighbouring groups
 * of arrays overlap.  statuses (may be NULL) receives one zbra_status per array; the return value
 * is the first non-OK status. */
int zbra_hip_file_decode(const uint8_t *image, size_t len, const zbra_file_array *arrays, size_t n, void *const *outs,
                         int *statuses);

/* One .zbra image to decode on one device: the arguments of zbra_hip_file_decode plus the device. */
typedef struct zbra_file_job {
    int32_t device;                /* HIP device index the job runs on */
    int32_t status;                /* out: what zbra_hip_file_decode returned for this job */
    const uint8_t *image;          /* HOST memory */
    uint64_t len;
    const zbra_file_array *arrays; /* from zbra_hip_file_index */
    uint64_t n;
    void *const *outs;             /* HOST buffers, one per array */
    int *statuses;                 /* may be NULL */
} zbra_file_job;

/* Whole files on several GPUs from ONE process: files (and Blocks) never exchange data
 * (binary.rs:124-127, 167-171), so file f goes to device jobs[f].device and nothing is reduced.
 * The call starts one host thread per distinct device; each sets its device and runs that device's
 * jobs in the order given through the pipeline of zbra_hip_file_decode (pinned hipMemcpyAsync both
 * ways, batched decode in between), all devices at once.  Returns when every job is done; the
 * return value is the first non-OK jobs[i].status in job order.  A device index that does not
 * exist gives that job ZBRA_ERR_INVALID_ARG without disturbing the others.  The pinned-memory copy
 * workers of the process are raised to eight per device in use (never above the CPUs the process
 * may run on). */
int zbra_hip_file_decode_multi(zbra_file_job *jobs, size_t n_jobs);

/* decompress_int_array / read_u32_array_compressed for `nd` arrays whose encoded bytes sit in
 * separate HOST buffers, as they do after the reference's read_exact into a fresh Vec per array
 * (binary.rs:648-654, 671-674): descs[i].data / .out are HOST pointers here, out holding
 * descs[i].count elements of descs[i].kind.  Same pipeline as zbra_hip_file_decode (which is this
 * call on the arrays of one image).  statuses (may be NULL) receives one zbra_status per array;
 * the return value is the first non-OK status. */
int zbra_hip_decode_batch(const zbra_array_desc *descs, size_t nd, int *statuses);

/* The write path's counterpart: int arrays in HOST memory (descs[i].values, .n, .kind; .out and
 * .cap ignored) become frames `u32 count | u32 clen | E` back to back in the HOST buffer `out`,
 * which is what Column::write_to emits for each of them (binary.rs:293-347 through :639-669).
 * offsets receives nd + 1 entries; statuses (may be NULL) one zbra_status per array; the return
 * value is the first non-OK status.  The arrays go through in rounds on two alternating streams:
 * while one round's values are uploaded and encoded, the previous round's frames are copied back
 * (by a second host thread that lives for the call), so both directions of the link are busy. */
int zbra_hip_encode_frames(const zbra_encode_desc *descs, size_t nd, uint8_t *out, size_t cap, uint64_t *offsets,
                           int *statuses);

#ifdef __cplusplus
}
#endif
#endif /* ZBRA_HIP_H */
