"""The wire format of one encoded int array, written down as arithmetic on ONE big integer.

TEST INFRASTRUCTURE ONLY: imported by tests/ and scripts/make_golden.py, never by the shipped
package.  A third reading beside oracle/zbra_oracle.c (byte-at-a-time bit buffer, as the reference
is written) and oracle/zbra_oracle_np.py (vectorised): this one has no bit buffer at all.  It states
what the loop of bp64_pack (zbra-core/src/compression.rs:152-175) amounts to,

    payload = sum(z[i] << (i * w))  as ceil(n * w / 8) little-endian bytes          (w <= 31)
    payload = z[0] | z[1] | ... as 8 little-endian bytes each                        (w >= 32, :143-149)

and decodes by shifting and masking the same integer.  PARITY PINNING: like the other two it is
derived from the reference's source text, not from bytes the reference produced ("parity
unpinned"); what it adds is that the byte layout of every golden vector, including the 643-byte
vector M whose bytes nobody typed by hand, now rests on three formulations that share no code.
Python integers only: meant for small inputs."""

_M64 = (1 << 64) - 1


def _wrap(x: int) -> int:
    """Two's-complement wrap to i64 (wrapping_add / wrapping_sub)."""
    x &= _M64
    return x - (1 << 64) if x >> 63 else x


def midpoint(values):
    """frame_of_reference_encode's midpoint (compression.rs:56-68): median of the signed-sorted copy;
    for an even count a.wrapping_add(b.wrapping_sub(a) / 2) with Rust's truncating division."""
    s = sorted(int(v) for v in values)
    n = len(s)
    if n == 0:
        return 0
    if n % 2:
        return s[n // 2]
    a, b = s[n // 2 - 1], s[n // 2]
    diff = _wrap(b - a)
    half = abs(diff) // 2 * (1 if diff >= 0 else -1)  # toward zero
    return _wrap(a + half)


def zigzag(d: int) -> int:
    """zig_zag_encode (compression.rs:88-98) on an i64."""
    return ((d << 1) ^ (d >> 63)) & _M64  # Python's >> on a negative int is arithmetic


def unzigzag(z: int) -> int:
    """zig_zag_decode (compression.rs:101-111)."""
    return _wrap((z >> 1) ^ -(z & 1))


def compress_int_array(values) -> bytes:
    """compress_int_array (compression.rs:271-292) as one expression per field."""
    vals = [int(v) for v in values]
    if not vals:
        return b""                                                      # :272-274
    mid = midpoint(vals)
    z = [zigzag(_wrap(v - mid)) for v in vals]
    zmax = max(z)
    w = zmax.bit_length() if zmax else 1                                # :123-128
    if w >= 32:
        payload = b"".join(x.to_bytes(8, "little") for x in z)          # :143-149
    else:
        stream = sum(x << (i * w) for i, x in enumerate(z))
        payload = stream.to_bytes((len(z) * w + 7) // 8, "little")      # :152-175
    packed = bytes([w]) + payload                                       # :131
    return (mid & _M64).to_bytes(8, "little") + (len(packed) & 0xFFFFFFFF).to_bytes(4, "little") + packed  # :286-289


def decompress_int_array(data: bytes, count: int):
    """decompress_int_array (compression.rs:295-330) for well-formed input: list of ints."""
    data = bytes(data)
    if not data:
        return []                                                       # :296-298
    mid = int.from_bytes(data[0:8], "little", signed=True)
    packed_len = int.from_bytes(data[8:12], "little")
    packed = data[12:12 + packed_len]
    if not packed:
        return []                                                       # :182-184
    w = packed[0]
    if w == 0 or count == 0:
        z = [0] * count                                                 # :187-189
    elif w >= 32:
        z = [int.from_bytes(packed[1 + 8 * i:9 + 8 * i], "little") if 9 + 8 * i <= len(packed) else 0 for i in range(count)]  # :195-207
    else:
        stream = int.from_bytes(packed[1:], "little")
        z = [(stream >> (i * w)) & ((1 << w) - 1) for i in range(count)]
    return [_wrap(mid + unzigzag(x)) for x in z]
