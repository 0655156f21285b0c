"""Host-side mirror of `zbra_core::binary` and the striped types it serialises
(zbra-core/src/binary.rs:17-200, 202-442, 486-636; zbra-core/src/striped.rs:9-83).

Writing walks the column tree exactly as `Table::write_to` / `Column::write_to` do and sends every
int array (Int values, Double bit patterns, Binary/Array/Nested lengths, Enum tags) through the
integer codec.  Reading is two-phase: `zbra_hip_file_index` finds every int array in the image
without decoding, `zbra_hip_file_decode` decodes them all (batched, copies overlapped with
kernels), and a second walk hangs the decoded vectors into the tree.

Byte blobs (`Table::Binary.data`, `Column::Binary.data`) use the file's CompressionConfig; only
`"None"` is supported here — the zstd byte codec is a different path and is not part of this
library.  The JSON schema is carried as an opaque string.
"""
import ctypes as C
import json
import struct
from dataclasses import dataclass, field
from typing import List, Tuple, Union

import numpy as np

from . import _lib
from . import compression as _gpu
from ._lib import FileArray, FileInfo, lib

MAGIC_NUMBER = b"||_ZBRA||00001||"                      # binary.rs:17
ALLOW, DENY = 0, 1                                       # data::Default, binary.rs:486-503
INT_ENCODINGS = ("Int", "Date", "TimeSeconds", "TimeMilliseconds", "TimeMicroseconds")  # :531-554
BINARY_ENCODINGS = ("Binary", "Utf8")                    # :556-573
COMPRESSION_NONE = '{"binary_data":"None","strings":"None"}'  # serde JSON of CompressionConfig{None, None}


class BinaryFormatError(_gpu.BinaryError):
    """BinaryError::InvalidMagicNumber / Invalid*Tag / IoError(UnexpectedEof)."""


# ---- striped types (striped.rs:9-83) -----------------------------------------------------------

@dataclass
class Unit:
    count: int


@dataclass
class Int:
    values: np.ndarray                       # int64
    encoding: str = "Int"                    # IntEncoding
    default: int = ALLOW


@dataclass
class Double:
    values: np.ndarray                       # float64
    default: int = ALLOW


@dataclass
class Binary:
    lengths: np.ndarray                      # uint64 (usize)
    data: bytes
    encoding: str = "Utf8"                   # BinaryEncoding
    default: int = ALLOW


@dataclass
class Array:
    lengths: np.ndarray
    element: "Column"
    default: int = ALLOW


@dataclass
class Struct:
    fields: List[Tuple[str, "Column"]]       # FieldColumn{name, column}
    default: int = ALLOW


@dataclass
class Enum:
    tags: np.ndarray                         # uint32
    variants: List[Tuple[str, int, "Column"]]  # VariantColumn{name, tag, column}
    default: int = ALLOW


@dataclass
class Nested:
    lengths: np.ndarray
    table: "Table"


@dataclass
class Reversed:
    inner: "Column"


Column = Union[Unit, Int, Double, Binary, Array, Struct, Enum, Nested, Reversed]


@dataclass
class BinaryTable:
    data: bytes
    encoding: str = "Binary"
    default: int = ALLOW


@dataclass
class ArrayTable:
    column: Column
    default: int = ALLOW


@dataclass
class MapTable:
    key_column: Column
    value_column: Column
    default: int = ALLOW


Table = Union[BinaryTable, ArrayTable, MapTable]


@dataclass
class Block:                                 # binary.rs:63-66
    row_count: int
    table: Table


# ---- writer (binary.rs:103-130, 184-356) ----------------------------------------------------

class _Writer:
    """The tree walk of write_to.  With a codec that has `write_int_arrays_compressed` (the GPU
    one) the int arrays are only collected during the walk and encoded in ONE batched submission
    at the end (`finish`), their frames then spliced in where the walk met them; with a
    per-array codec (a test's stand-in) each array is encoded on the spot, as the reference does."""

    def __init__(self, codec, blobs_stored=(True, True)):
        self.out = bytearray()
        self.codec = codec
        # CompressionAlgorithm::None for (binary_data, strings): only then can a byte blob be written here
        self.blobs_stored = blobs_stored
        self.batched = hasattr(codec, "write_int_arrays_compressed")
        self.pending = []                                  # (position in self.out, array)

    def finish(self) -> bytes:
        if not self.pending:
            return bytes(self.out)
        frames, offs = self.codec.write_int_arrays_compressed([a for _, a in self.pending], as_bytes=False)
        walk = memoryview(self.out)
        parts, prev = [], 0
        for i, (pos, _) in enumerate(self.pending):
            parts.append(walk[prev:pos])
            parts.append(memoryview(frames[offs[i]:offs[i + 1]]))
            prev = pos
        parts.append(walk[prev:])
        return b"".join(parts)                             # the one copy of the image

    def u8(self, v):
        self.out.append(v)

    def u32(self, v):
        self.out += struct.pack("<I", v & 0xFFFFFFFF)

    def string(self, s):                                   # write_string :599-604
        b = s.encode()
        self.u32(len(b))
        self.out += b

    def sized(self, b):                                    # write_sized_byte_array :621-626
        self.u32(len(b))
        self.u32(len(b))
        self.out += b

    def zblob(self, data, which):                          # write_sized_byte_array_compressed :677-687
        if not self.blobs_stored[which]:
            raise NotImplementedError("byte blobs are written with CompressionAlgorithm::None only (zstd is another path)")
        self.sized(bytes(data))                            # compress_binary(None) = the bytes as they are

    def encoding(self, kind, name):                        # :507-519
        if kind == "int":
            self.u8(0)
            self.u8(INT_ENCODINGS.index(name))
        else:
            self.u8(1)
            self.u8(BINARY_ENCODINGS.index(name))

    def int_array(self, values):                           # write_int_array_compressed :639-645
        v = np.ascontiguousarray(values, dtype=np.int64)
        if self.batched:
            self.pending.append((len(self.out), v))
            return
        self.u32(v.size)
        e = self.codec.compress_int_array(v)
        self.u32(len(e))
        self.out += e

    def u32_array(self, values):                           # write_u32_array_compressed :666-669
        v = np.ascontiguousarray(values, dtype=np.uint32)
        if self.batched:
            self.pending.append((len(self.out), v))
            return
        self.u32(v.size)
        e = self.codec.compress_u32_array(v) if hasattr(self.codec, "compress_u32_array") \
            else self.codec.compress_int_array(v.astype(np.int64))
        self.u32(len(e))
        self.out += e

    def column(self, c):                                   # Column::write_to :275-356
        if isinstance(c, Unit):
            self.u8(0); self.u32(c.count)
        elif isinstance(c, Int):
            self.u8(1); self.u8(c.default); self.encoding("int", c.encoding); self.int_array(c.values)
        elif isinstance(c, Double):
            self.u8(2); self.u8(c.default)
            self.int_array(np.ascontiguousarray(c.values, dtype=np.float64).view(np.int64))  # f64::to_bits as i64
        elif isinstance(c, Binary):
            self.u8(3); self.u8(c.default); self.encoding("binary", c.encoding)
            self.int_array(np.asarray(c.lengths, dtype=np.uint64).view(np.int64))
            self.zblob(c.data, 1)                          # compression.strings
        elif isinstance(c, Array):
            self.u8(4); self.u8(c.default)
            self.int_array(np.asarray(c.lengths, dtype=np.uint64).view(np.int64))
            self.column(c.element)
        elif isinstance(c, Struct):
            self.u8(5); self.u8(c.default); self.u32(len(c.fields))
            for name, col in c.fields:
                self.string(name); self.column(col)
        elif isinstance(c, Enum):
            self.u8(6); self.u8(c.default); self.u32_array(c.tags); self.u32(len(c.variants))
            for name, tag, col in c.variants:
                self.string(name); self.u32(tag); self.column(col)
        elif isinstance(c, Nested):
            self.u8(7); self.int_array(np.asarray(c.lengths, dtype=np.uint64).view(np.int64)); self.table(c.table)
        elif isinstance(c, Reversed):
            self.u8(8); self.column(c.inner)
        else:
            raise TypeError(type(c))

    def table(self, t):                                    # Table::write_to :204-237
        if isinstance(t, BinaryTable):
            self.u8(0); self.u8(t.default); self.encoding("binary", t.encoding); self.zblob(t.data, 0)  # compression.binary_data
        elif isinstance(t, ArrayTable):
            self.u8(1); self.u8(t.default); self.column(t.column)
        elif isinstance(t, MapTable):
            self.u8(2); self.u8(t.default); self.column(t.key_column); self.column(t.value_column)
        else:
            raise TypeError(type(t))


# ---- reader: second walk over an indexed image --------------------------------------------------

class _Reader:
    def __init__(self, image: bytes, decoded):
        self.b = image
        self.pos = 0
        self.decoded = iter(decoded)

    def take(self, n):
        if len(self.b) - self.pos < n:
            raise BinaryFormatError("failed to fill whole buffer")
        v = self.b[self.pos:self.pos + n]
        self.pos += n
        return v

    def u8(self):
        return self.take(1)[0]

    def u32(self):
        return struct.unpack("<I", self.take(4))[0]

    def string(self):
        return bytes(self.take(self.u32())).decode()

    def sized(self):                                       # read_sized_byte_array(_compressed) :629-636, 690-699
        self.u32()
        return bytes(self.take(self.u32()))

    def default(self):
        t = self.u8()
        if t > 1:
            raise BinaryFormatError(f"InvalidDefaultTag({t})")
        return t

    def encoding(self):
        t = self.u8()
        s = self.u8()
        if t == 0 and s < len(INT_ENCODINGS):
            return INT_ENCODINGS[s]
        if t == 1 and s < len(BINARY_ENCODINGS):
            return BINARY_ENCODINGS[s]
        raise BinaryFormatError(f"InvalidEncodingTag({t},{s})")

    def int_array(self):                                   # the vector was decoded by zbra_hip_file_decode
        self.u32()
        self.take(self.u32())
        return next(self.decoded)

    def column(self):                                      # Column::read_from :359-441
        tag = self.u8()
        if tag == 0:
            return Unit(self.u32())
        if tag == 1:
            d = self.default(); e = self.encoding()
            return Int(self.int_array(), e, d)
        if tag == 2:
            d = self.default()
            return Double(self.int_array().view(np.float64), d)   # f64::from_bits
        if tag == 3:
            d = self.default(); e = self.encoding(); lengths = self.int_array().view(np.uint64)
            return Binary(lengths, self.sized(), e, d)
        if tag == 4:
            d = self.default(); lengths = self.int_array().view(np.uint64)
            return Array(lengths, self.column(), d)
        if tag == 5:
            d = self.default(); n = self.u32()
            return Struct([(self.string(), self.column()) for _ in range(n)], d)
        if tag == 6:
            d = self.default(); tags = self.int_array(); n = self.u32()
            variants = []
            for _ in range(n):
                name = self.string(); t = self.u32()
                variants.append((name, t, self.column()))
            return Enum(tags, variants, d)
        if tag == 7:
            lengths = self.int_array().view(np.uint64)
            return Nested(lengths, self.table())
        if tag == 8:
            return Reversed(self.column())
        raise BinaryFormatError(f"InvalidColumnTag({tag})")

    def table(self):                                       # Table::read_from :240-270
        tag = self.u8()
        if tag == 0:
            d = self.default(); e = self.encoding()
            return BinaryTable(self.sized(), e, d)
        if tag == 1:
            return ArrayTable(default=self.default(), column=self.column())
        if tag == 2:
            d = self.default(); k = self.column()
            return MapTable(k, self.column(), d)
        raise BinaryFormatError(f"InvalidTableTag({tag})")


def file_index(image):
    """zbra_hip_file_index: (FileInfo, ctypes array of FileArray).  Host only."""
    buf = np.frombuffer(image, dtype=np.uint8)
    info = FileInfo()
    err = C.c_size_t(0)
    ptr = buf.ctypes.data_as(C.c_void_p) if buf.size else None
    st = lib.zbra_hip_file_index(ptr, buf.size, None, 0, C.byref(info), C.byref(err))
    if st == _lib.ERR_FORMAT:
        raise BinaryFormatError(f"{_lib.strerror(st)} at offset {err.value}")
    if st:
        _gpu._raise(st)
    arrays = (FileArray * max(1, info.n_arrays))()
    st = lib.zbra_hip_file_index(ptr, buf.size, arrays, info.n_arrays, C.byref(info), C.byref(err))
    if st:
        _gpu._raise(st)
    return info, arrays


def file_decode(image, info, arrays):
    """zbra_hip_file_decode: decodes every indexed int array; returns a list of numpy arrays
    (int64, or uint32 for enum tags) in file order."""
    buf = np.frombuffer(image, dtype=np.uint8)
    n = info.n_arrays
    # decompress_int_array returns an empty Vec for empty data and for packed_len == 0 whatever
    # count says (compression.rs:296-298, :182-184): the device writes nothing there, so the buffer
    # is sized by what the header says will come (a corrupt frame's count allocates nothing);
    # a malformed E raises here what the decode would raise
    sizes = [_gpu.decoded_count(buf[arrays[i].offset:arrays[i].offset + arrays[i].len], arrays[i].count) for i in range(n)]
    outs = [np.empty(sizes[i], dtype=np.uint32 if arrays[i].kind == _lib.ELEM_U32 else np.int64) for i in range(n)]
    nothing = np.empty(2, dtype=np.int64)  # where an array that produces no value "goes": never written
    ptrs = (C.c_void_p * max(1, n))(*[o.ctypes.data if o.size else nothing.ctypes.data for o in outs])
    statuses = (C.c_int * max(1, n))()
    st = lib.zbra_hip_file_decode(buf.ctypes.data_as(C.c_void_p), buf.size, arrays, n, ptrs, statuses)
    if st:
        _gpu._raise(st)
    return outs


@dataclass
class BinaryFile:                            # binary.rs:56-59
    schema_json: str
    blocks: List[Block] = field(default_factory=list)
    compression_json: str = COMPRESSION_NONE

    def to_bytes(self, codec=_gpu) -> bytes:               # write_to :103-130
        cfg = json.loads(self.compression_json)
        # a table without byte blobs (int, double, enum, array columns) is written under any config
        w = _Writer(codec, (cfg.get("binary_data") == "None", cfg.get("strings") == "None"))
        w.out += MAGIC_NUMBER
        w.sized(self.schema_json.encode())
        w.sized(self.compression_json.encode())
        w.u32(len(self.blocks))
        for b in self.blocks:                              # Block::write_to :184-192
            w.u32(b.row_count)
            w.table(b.table)
        return w.finish()

    @classmethod
    def from_bytes(cls, image) -> "BinaryFile":            # read_from :139-174
        image = bytes(image)
        info, arrays = file_index(image)
        schema = image[info.schema_offset:info.schema_offset + info.schema_len].decode()
        comp = image[info.compression_offset:info.compression_offset + info.compression_len].decode()
        cfg = json.loads(comp)
        decoded = file_decode(image, info, arrays)
        r = _Reader(image, decoded)
        r.pos = info.blocks_offset
        blocks = []
        for _ in range(r.u32()):
            rows = r.u32()
            blocks.append(Block(rows, r.table()))
        if (cfg.get("binary_data") != "None" or cfg.get("strings") != "None") and _has_blobs(blocks):
            raise NotImplementedError("this image stores byte blobs with zstd; only its int arrays are decoded here")
        return cls(schema, blocks, comp)


def _has_blobs(blocks):
    def col(c):
        if isinstance(c, Binary):
            return len(c.data) > 0
        if isinstance(c, Array):
            return col(c.element)
        if isinstance(c, Struct):
            return any(col(x) for _, x in c.fields)
        if isinstance(c, Enum):
            return any(col(x) for _, _, x in c.variants)
        if isinstance(c, Nested):
            return tab(c.table)
        if isinstance(c, Reversed):
            return col(c.inner)
        return False

    def tab(t):
        if isinstance(t, BinaryTable):
            return len(t.data) > 0
        if isinstance(t, ArrayTable):
            return col(t.column)
        return col(t.key_column) or col(t.value_column)
    return any(tab(b.table) for b in blocks)
