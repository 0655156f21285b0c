"""Synthetic columns of the benchmark configurations, generated in place on the device.

Counter-based splitmix64, sm(seed, i) = finalise(seed + (i+1)*0x9E3779B97F4A7C15): the same value
for index i on the CPU, in numpy and here, so multi-GB inputs never cross PCIe.  torch has no
unsigned 64-bit arithmetic; int64 multiplication wraps and logical shifts are masked arithmetic
shifts, which gives the same bits.
"""
import torch

_M64 = (1 << 64) - 1


def _s(x: int) -> int:
    """Python int -> the int64 with the same low 64 bits."""
    x &= _M64
    return x - (1 << 64) if x >> 63 else x


_GOLDEN = _s(0x9E3779B97F4A7C15)
_C1 = _s(0xBF58476D1CE4E5B9)
_C2 = _s(0x94D049BB133111EB)


def _lsr(z: torch.Tensor, s: int) -> torch.Tensor:
    return (z >> s) & ((1 << (64 - s)) - 1)


def sm(seed: int, idx: torch.Tensor) -> torch.Tensor:
    """splitmix64 of int64 index tensor `idx`; result holds the u64 value's bits as int64."""
    z = (idx + 1) * _GOLDEN + _s(seed)
    z = (z ^ _lsr(z, 30)) * _C1
    z = (z ^ _lsr(z, 27)) * _C2
    return z ^ _lsr(z, 31)


def umod(z: torch.Tensor, m: int) -> torch.Tensor:
    """(z as u64) mod m for m < 2^31."""
    return ((_lsr(z, 1) % m) * 2 + (z & 1)) % m


def narrow_column(seed: int, first: int, n: int, device, chunk: int = 1 << 24) -> torch.Tensor:
    """Headline column: v[i] = 1.7e12 + sm(seed, first + i) mod 4000  (bit width 12 after FOR)."""
    out = torch.empty(n, dtype=torch.int64, device=device)
    for s in range(0, n, chunk):
        e = min(n, s + chunk)
        idx = torch.arange(first + s, first + e, dtype=torch.int64, device=device)
        out[s:e] = umod(sm(seed, idx), 4000) + 1_700_000_000_000
    return out


def mixed_width_chunk(c: int, first: int, n: int, device) -> torch.Tensor:
    """Chunk c of the mixed-width stress column: signed 48-bit base b_c, amplitude a_c in 1..40
    bits, v = b_c + (sm(0x5EED2003, i) mod 2^a_c) - 2^(a_c-1)."""
    cidx = torch.tensor([c], dtype=torch.int64, device=device)
    base = int(sm(0x5EED0003, cidx).item())
    base = ((base & ((1 << 48) - 1)) ^ (1 << 47)) - (1 << 47)  # sign-extend 48 bits
    a = 1 + int(umod(sm(0x5EED1003, cidx), 40).item())
    idx = torch.arange(first, first + n, dtype=torch.int64, device=device)
    r = sm(0x5EED2003, idx) & ((1 << a) - 1)
    return r - (1 << (a - 1)) + base


I64_MIN, I64_MAX = -(1 << 63), (1 << 63) - 1


def mixed_width_column(n_total: int, chunk_rows: int, device):
    """Mixed-width stress column: n_total rows cut into arrays of chunk_rows rows (the last one
    shorter).  Chunk c >= 5 is mixed_width_chunk(c, ...); chunks 0..4 are adversarial:
    0 constant, 1 alternating i64::MIN / i64::MAX, 2 full-range random (width 64), 3 a single
    value, 4 an odd length (chunk_rows - 1).  Returns a list of int64 tensors."""
    out = []
    first, c = 0, 0
    while first < n_total:
        n = min(chunk_rows, n_total - first)
        if c == 0:
            t = torch.full((n,), -987654321012345, dtype=torch.int64, device=device)
        elif c == 1:
            t = torch.where(torch.arange(n, device=device) % 2 == 0, torch.tensor(I64_MIN, device=device),
                            torch.tensor(I64_MAX, device=device))
        elif c == 2:
            t = sm(0x5EED2003, torch.arange(first, first + n, dtype=torch.int64, device=device))
        elif c == 3:
            n = 1
            t = torch.tensor([424242], dtype=torch.int64, device=device)
        elif c == 4:
            n = min(n, chunk_rows - 1)
            t = mixed_width_chunk(c, first, n, device)
        else:
            t = mixed_width_chunk(c, first, n, device)
        out.append(t)
        first += n
        c += 1
    return out


_CLUSTERS = (100, 1000, 10000, 100000)


def enum_block(block: int, rows: int, device):
    """One Block of the sum-typed table: Enum.tags over {0,1,2,7} with probabilities
    50/25/15/10 % (row r of the table draws sm(0x5EED0004, r) mod 100), and per variant a struct
    of two Int columns whose length is that variant's row count: A clustered
    (benches/compression.rs:25-34 pattern), B time series with 1 s steps (:36-40 pattern).
    Returns (tags:int32 tensor, [(tag, colA, colB), ...])."""
    r = torch.arange(block * rows, (block + 1) * rows, dtype=torch.int64, device=device)
    u = umod(sm(0x5EED0004, r), 100)
    tags = torch.where(u < 50, 0, torch.where(u < 75, 1, torch.where(u < 90, 2, 7))).to(torch.int32)
    variants = []
    clusters = torch.tensor(_CLUSTERS, dtype=torch.int64, device=device)
    for tag in (0, 1, 2, 7):
        cnt = int((tags == tag).sum().item())
        i = torch.arange(cnt, dtype=torch.int64, device=device)
        col_a = clusters[i % 4] + (i % 20) - 10
        col_b = 1640995200000 + (block * rows + i) * 1000
        variants.append((tag, col_a, col_b))
    return tags, variants


# ---- the full-file configuration: host-side generator (a .zbra file starts in host memory) ------

def sm_np(seed: int, idx):
    """splitmix64 of a numpy uint64 index array: the same values as sm() above."""
    import numpy as np
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + (idx.astype(np.uint64) + np.uint64(1)) * np.uint64(0x9E3779B97F4A7C15)
        z ^= z >> np.uint64(30)
        z *= np.uint64(0xBF58476D1CE4E5B9)
        z ^= z >> np.uint64(27)
        z *= np.uint64(0x94D049BB133111EB)
        z ^= z >> np.uint64(31)
    return z


FILE_COLUMNS = ("attribute_id", "entity_id", "time", "value")  # struct keys in the order the reference CLI emits them (alphabetical)


def file_block_columns(block: int, rows: int):
    """The four Int columns of Block `block` of the full-file configuration's table (rows
    [block*rows, (block+1)*rows) of it), as int64 numpy arrays keyed by FILE_COLUMNS:
    entity_id clustered (runs of 64 equal ids out of 50000), attribute_id small range (< 200, at
    most 8 bits), time milliseconds, monotone with 1 s steps, value < 4000 (at most 12 bits)."""
    import numpy as np
    i = np.arange(block * rows, (block + 1) * rows, dtype=np.uint64)
    return {
        "attribute_id": (sm_np(0x5EED0005, i) % np.uint64(200)).astype(np.int64),
        "entity_id": (np.uint64(1000) + (i // np.uint64(64)) % np.uint64(50000)).astype(np.int64),
        "time": 1_700_000_000_000 + i.astype(np.int64) * 1000,
        "value": (sm_np(0x5EED1005, i) % np.uint64(4000)).astype(np.int64),
    }


def file_blocks(first_block: int, n_blocks: int, rows: int):
    """Blocks [first_block, first_block + n_blocks) of that table as binary.Block objects:
    array<struct{attribute_id:int, entity_id:int, time:time_milliseconds, value:int}>, the schema
    of the reference CLI's time-series example (zbra-cli/src/main.rs:581-594)."""
    from . import binary as zb
    out = []
    for b in range(first_block, first_block + n_blocks):
        c = file_block_columns(b, rows)
        out.append(zb.Block(rows, zb.ArrayTable(zb.Struct([
            ("attribute_id", zb.Int(c["attribute_id"])), ("entity_id", zb.Int(c["entity_id"])),
            ("time", zb.Int(c["time"], "TimeMilliseconds")), ("value", zb.Int(c["value"]))]))))
    return out
