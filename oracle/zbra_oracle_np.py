"""numpy restatement of zbra-core's integer column codec (second, independent reading).

TEST INFRASTRUCTURE ONLY: imported by tests/ (and by scripts that generate
tests/golden/), never by the shipped package.

PARITY PINNING: the reference is Rust and cannot be run here, and it ships no
golden bytes; this file and oracle/zbra_oracle.c are two separately written
readings of zbra-core/src/compression.rs that must agree with each other and
with the hand-derived vectors in tests/golden/.  "parity unpinned" against
bytes produced by the reference itself.

Written array-at-a-time (vectorised) on purpose, so that it shares no loop
structure with the C restatement.
"""
import numpy as np

OK = 0
ERR_SHORT_HEADER = 1  # compression.rs:300-304
ERR_SHORT_PACKED = 2  # compression.rs:312-316
ERR_TRUNCATED = 3     # compression.rs:229 underflow: unspecified in the reference

_U64 = np.uint64
_MASK64 = (1 << 64) - 1


def _to_i64(x: int) -> int:
    x &= _MASK64
    return x - (1 << 64) if x >> 63 else x


def frame_of_reference_encode(values):
    """compression.rs:52-74.  Returns (midpoint:int, deltas:int64 array)."""
    v = np.asarray(values, dtype=np.int64)
    n = v.size
    if n == 0:
        return 0, np.zeros(0, dtype=np.int64)
    s = np.sort(v, kind="stable")
    if n % 2 == 0:
        a = int(s[n // 2 - 1])
        b = int(s[n // 2])
        diff = _to_i64(b - a)                       # wrapping_sub
        half = abs(diff) // 2 * (1 if diff >= 0 else -1)   # truncating division
        mid = _to_i64(a + half)                     # wrapping_add
    else:
        mid = int(s[n // 2])
    deltas = (v.view(np.uint64) - _U64(mid & _MASK64)).view(np.int64)
    return mid, deltas


def frame_of_reference_decode(midpoint, deltas):
    """compression.rs:77-82."""
    d = np.asarray(deltas, dtype=np.int64)
    return (d.view(np.uint64) + _U64(midpoint & _MASK64)).view(np.int64)


def zig_zag_encode(values):
    """compression.rs:88-98."""
    v = np.asarray(values, dtype=np.int64)
    return (v.view(np.uint64) << _U64(1)) ^ (v >> np.int64(63)).view(np.uint64)


def zig_zag_decode(values):
    """compression.rs:101-111."""
    u = np.asarray(values, dtype=np.uint64)
    return ((u >> _U64(1)) ^ (_U64(0) - (u & _U64(1)))).view(np.int64)


def bit_width_of(max_value: int) -> int:
    """compression.rs:124-128."""
    return 1 if max_value == 0 else int(max_value).bit_length()


def bp64_pack(values) -> bytes:
    """compression.rs:117-178."""
    u = np.asarray(values, dtype=np.uint64)
    if u.size == 0:
        return b""
    w = bit_width_of(int(u.max()))
    if w >= 32:
        return bytes([w]) + u.astype("<u8").tobytes()
    # value i occupies bits [i*w, i*w+w) of an LSB-first little-endian stream
    bits = ((u[:, None] >> np.arange(w, dtype=np.uint64)[None, :]) & _U64(1)).astype(np.uint8)
    return bytes([w]) + np.packbits(bits.reshape(-1), bitorder="little").tobytes()


def bp64_unpack(packed: bytes, count: int):
    """compression.rs:181-233.  Returns (status, uint64 array)."""
    packed = bytes(packed)
    if len(packed) == 0:
        return OK, np.zeros(0, dtype=np.uint64)
    w = packed[0]
    if w == 0 or count == 0:
        return OK, np.zeros(count, dtype=np.uint64)
    data = np.frombuffer(packed, dtype=np.uint8, offset=1)
    if w >= 32:
        whole = min(count, data.size // 8)
        out = np.zeros(count, dtype=np.uint64)
        out[:whole] = np.frombuffer(data[: whole * 8].tobytes(), dtype="<u8")
        return OK, out
    need = (count * w + 7) // 8
    if data.size < need:
        return ERR_TRUNCATED, np.zeros(0, dtype=np.uint64)
    bits = np.unpackbits(data[:need], bitorder="little")[: count * w].reshape(count, w).astype(np.uint64)
    out = (bits << np.arange(w, dtype=np.uint64)[None, :]).sum(axis=1, dtype=np.uint64)
    return OK, out


def compress_int_array(values) -> bytes:
    """compression.rs:271-292."""
    v = np.asarray(values, dtype=np.int64)
    if v.size == 0:
        return b""
    mid, deltas = frame_of_reference_encode(v)
    packed = bp64_pack(zig_zag_encode(deltas))
    return (mid & _MASK64).to_bytes(8, "little") + (len(packed) & 0xFFFFFFFF).to_bytes(4, "little") + packed


def decompress_int_array(data: bytes, count: int):
    """compression.rs:295-330.  Returns (status, int64 array)."""
    data = bytes(data)
    empty = np.zeros(0, dtype=np.int64)
    if len(data) == 0:
        return OK, empty
    if len(data) < 12:
        return ERR_SHORT_HEADER, empty
    mid = int.from_bytes(data[0:8], "little", signed=True)
    packed_len = int.from_bytes(data[8:12], "little")
    if len(data) < 12 + packed_len:
        return ERR_SHORT_PACKED, empty
    st, u = bp64_unpack(data[12 : 12 + packed_len], count)
    if st != OK:
        return st, empty
    return OK, frame_of_reference_decode(mid, zig_zag_decode(u))


def write_int_array_compressed(values) -> bytes:
    """binary.rs:639-645: u32 count | u32 clen | E."""
    v = np.asarray(values, dtype=np.int64)
    e = compress_int_array(v)
    return (v.size & 0xFFFFFFFF).to_bytes(4, "little") + (len(e) & 0xFFFFFFFF).to_bytes(4, "little") + e


def sm(seed: int, i):
    """Counter-based splitmix64: finalise(seed + (i+1)*0x9E3779B97F4A7C15) mod 2^64."""
    with np.errstate(over="ignore"):
        z = _U64(seed & _MASK64) + (np.asarray(i, dtype=np.uint64) + _U64(1)) * _U64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> _U64(30))) * _U64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> _U64(27))) * _U64(0x94D049BB133111EB)
        return z ^ (z >> _U64(31))


def enum_rows(tags, variant_tags):
    """The row loop of Column::from_values for an Enum column (zbra-core/src/striped.rs:344-359):
    row i goes to the FIRST variant whose tag equals tags[i] (`position`, :349) and lands at the end
    of that variant's list (:351), so its position there is the number of earlier rows of the same
    variant.  Returns (status, variant[n] u32, rank[n] u64, counts[V] u64, first_bad_row); status 8
    and first_bad_row = the row whose tag no variant carries (:352-358), where the reference stops."""
    tags = np.asarray(tags).astype(np.uint32)
    vt = [int(t) & 0xFFFFFFFF for t in variant_tags]
    n = tags.size
    variant = np.full(n, 0xFFFFFFFF, dtype=np.uint32)
    for v in range(len(vt) - 1, -1, -1):          # later variants first, so the first match wins
        variant[tags == np.uint32(vt[v])] = v
    bad = np.flatnonzero(variant == 0xFFFFFFFF)
    if bad.size:
        return 8, variant, None, None, int(bad[0])
    rank = np.zeros(n, dtype=np.uint64)
    counts = np.zeros(len(vt), dtype=np.uint64)
    for v in range(len(vt)):
        m = variant == v
        c = int(m.sum())
        rank[m] = np.arange(c, dtype=np.uint64)
        counts[v] = c
    return 0, variant, rank, counts, -1


def enum_rows_loop(tags, variant_tags):
    """The same, statement by statement as the reference's loop (small inputs only)."""
    lists = [[] for _ in variant_tags]
    variant, rank = [], []
    for i, tag in enumerate(tags):
        pos = next((v for v, t in enumerate(variant_tags) if int(t) == int(tag)), None)   # striped.rs:349
        if pos is None:
            return 8, variant, rank, None, i                                              # :352-358
        variant.append(pos)
        rank.append(len(lists[pos]))
        lists[pos].append(i)                                                              # :351
    return 0, variant, rank, [len(x) for x in lists], -1
