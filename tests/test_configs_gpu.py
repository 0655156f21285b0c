"""GPU parity on the shapes of the other benchmark configurations, at sizes the oracle finishes in
seconds: mixed per-array widths with the adversarial arrays (config 3) and the sum-typed table's
Blocks — enum tags as u32 plus two Int columns per variant with ragged lengths (config 4).  The
device generators are checked against numpy restatements of the same formulas first."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a device"
    from oracle import zbra_oracle_c as oc
    from oracle import zbra_oracle_np as on
    from zbra_amd import _lib, device, workloads
    return torch, device, workloads, oc, on, _lib


def _np_mixed_chunk(on, c, first, n):
    base = int(on.sm(0x5EED0003, np.array([c], dtype=np.uint64))[0])
    base = ((base & ((1 << 48) - 1)) ^ (1 << 47)) - (1 << 47)
    a = 1 + int(on.sm(0x5EED1003, np.array([c], dtype=np.uint64))[0] % np.uint64(40))
    idx = np.arange(first, first + n, dtype=np.uint64)
    r = (on.sm(0x5EED2003, idx) & np.uint64((1 << a) - 1)).astype(np.int64)
    return r - (1 << (a - 1)) + base


def test_generators_match_numpy(env):
    torch, device, workloads, oc, on, L = env
    dev = torch.device("cuda", 0)
    v = workloads.narrow_column(0x5EED0002, 12345, 5000, dev).cpu().numpy()
    assert np.array_equal(v, oc.gen_narrow(0x5EED0002, 12345, 5000))
    for c in (5, 6, 17, 40):
        t = workloads.mixed_width_chunk(c, 1000 * c, 3000, dev).cpu().numpy()
        assert np.array_equal(t, _np_mixed_chunk(on, c, 1000 * c, 3000)), c
    tags, variants = workloads.enum_block(3, 4096, dev)
    r = np.arange(3 * 4096, 4 * 4096, dtype=np.uint64)
    u = (on.sm(0x5EED0004, r) % np.uint64(100)).astype(np.int64)
    expect = np.where(u < 50, 0, np.where(u < 75, 1, np.where(u < 90, 2, 7))).astype(np.int32)
    assert np.array_equal(tags.cpu().numpy(), expect)
    assert sum(a.numel() for _, a, _ in variants) == 4096


def _batch_roundtrip_vs_oracle(env, cols):
    torch, device, workloads, oc, on, L = env
    dev = cols[0].device
    kinds = [L.ELEM_U32 if c.dtype == torch.int32 else L.ELEM_I64 for c in cols]
    caps = [device.encode_bound(c.numel()) for c in cols]
    offs, total = [], 0
    for cap in caps:
        offs.append(total)
        total += cap + 5
    enc = torch.zeros(total + 64, dtype=torch.uint8, device=dev)
    res = device.new_results(len(cols), dev)
    device.encode_batch_dev(device.make_encode_descs(
        [(c.data_ptr(), c.numel(), enc.data_ptr() + o, cap, k) for c, o, cap, k in zip(cols, offs, caps, kinds)]), res)
    rs = device.read_results(res)
    enc_h = enc.cpu().numpy()
    lens = []
    for i, c in enumerate(cols):
        h = c.cpu().numpy()
        expect = oc.compress_int_array(h.view(np.uint32).astype(np.int64) if c.dtype == torch.int32 else h)
        assert rs[i].status == 0 and rs[i].out_count == len(expect), i
        assert enc_h[offs[i]:offs[i] + len(expect)].tobytes() == expect, i
        lens.append(len(expect))
    outs = [torch.full_like(c, -9) for c in cols]
    dres = device.new_results(len(cols), dev)
    device.decode_batch_dev(device.make_descs(
        [(enc.data_ptr() + o, ln, c.numel(), out.data_ptr(), k) for c, o, ln, out, k in zip(cols, offs, lens, outs, kinds)]), dres)
    for i, (c, o) in enumerate(zip(cols, outs)):
        assert torch.equal(c, o), i
    return sorted(set(r.width for r in rs))


def test_config3_mixed_widths_reduced(env):
    torch, device, workloads, oc, on, L = env
    cols = workloads.mixed_width_column(60 * 4096 + 1234, 4096, torch.device("cuda", 0))
    assert cols[0].numel() == 4096 and cols[3].numel() == 1 and cols[4].numel() == 4095
    widths = _batch_roundtrip_vs_oracle(env, cols)
    assert 1 in widths and 64 in widths and len(widths) > 20  # constant / MIN-MAX -> 1, full range -> 64


def test_config4_enum_blocks_reduced(env):
    torch, device, workloads, oc, on, L = env
    cols = []
    for b in range(3):
        tags, variants = workloads.enum_block(b, 1 << 15, torch.device("cuda", 0))
        cols.append(tags)
        for _, a, bcol in variants:
            cols += [a, bcol]
    widths = _batch_roundtrip_vs_oracle(env, cols)
    assert 4 in widths  # tags {0,1,2,7} around the median


def test_batch_encode_medians_at_scale(env):
    """A round trip succeeds with any midpoint, so at sizes beyond the oracle's reach the medians
    are checked against a device sort.  The mix is the one that exposes a workgroup whose lanes
    disagree about flushing its staged candidates: two-valued tag arrays (their workgroups run at
    another pace) between wide-range arrays that take the candidate-list route."""
    torch, device, _, _, _, L = env
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(11)
    n = 1 << 22
    cols = []
    for b in range(24):
        tags = torch.randint(0, 2, (n,), dtype=torch.int64, device=dev, generator=g)
        cols.append(tags.to(torch.int32))
        cols.append(torch.randint(-(1 << 24), 1 << 24, (n - 1000 - 37 * b,), dtype=torch.int64, device=dev, generator=g))
    outs = [torch.empty(device.encode_bound(c.numel()), dtype=torch.uint8, device=dev) for c in cols]
    descs = device.make_encode_descs([(c.data_ptr(), c.numel(), o.data_ptr(), o.numel(),
                                       L.ELEM_U32 if c.dtype == torch.int32 else L.ELEM_I64) for c, o in zip(cols, outs)])
    want = []
    for c in cols:
        s, _ = torch.sort(c.to(torch.int64))
        m = c.numel()
        a, b = int(s[(m - 1) // 2]), int(s[m // 2])
        d = b - a
        want.append(a + (d // 2 if d >= 0 else -((-d) // 2)))
    for rep in range(2):
        res = device.new_results(len(cols), dev)
        device.encode_batch_dev(descs, res)
        rs = device.read_results(res)
        assert [r.status for r in rs] == [0] * len(cols)
        assert [r.midpoint for r in rs] == want, rep
        assert all((r.width in (1, 2)) if i % 2 == 0 else r.width == 26 for i, r in enumerate(rs))


def test_config5_file_blocks_reduced(env):
    """The full-file configuration's own shape (Blocks of 4 Int columns: attribute id, entity id,
    time, value) at 3 Blocks of 2^15 rows: the image written through the C ABI
    (zbra_hip_encode_frames behind BinaryFile.to_bytes) equals the oracle-built image byte for
    byte, and zbra_hip_file_index + zbra_hip_file_decode give back the generated columns, which
    are regenerated here from the formulas and not taken from the tree."""
    import ctypes as C
    torch, device, workloads, oc, on, L = env
    from zbra_amd import binary as zb

    class OracleCodec:
        compress_int_array = staticmethod(oc.compress_int_array)

    rows, first_block, n_blocks = 1 << 15, 5, 3
    blocks = workloads.file_blocks(first_block, n_blocks, rows)
    f = zb.BinaryFile('{"Array":{"default":"Allow"}}', blocks)
    img = f.to_bytes()
    assert img == f.to_bytes(OracleCodec)
    info, arrays = zb.file_index(img)
    assert info.block_count == n_blocks and info.n_arrays == 4 * n_blocks and info.total_values == 4 * n_blocks * rows
    # the frames alone, through the C ABI with the arrays as the caller holds them: the bytes between
    # each frame's start and end in the image
    want_cols = []
    for b in range(first_block, first_block + n_blocks):
        i = np.arange(b * rows, (b + 1) * rows, dtype=np.uint64)
        want_cols += [(on.sm(0x5EED0005, i) % np.uint64(200)).astype(np.int64),
                      (1000 + (i // np.uint64(64)) % np.uint64(50000)).astype(np.int64),
                      1_700_000_000_000 + i.astype(np.int64) * 1000,
                      (on.sm(0x5EED1005, i) % np.uint64(4000)).astype(np.int64)]
    na = len(want_cols)
    edescs = device.make_encode_descs([(c.ctypes.data, c.size, 0, 0, L.ELEM_I64) for c in want_cols])
    cap = sum(8 + device.encode_bound(c.size) for c in want_cols)
    frames = np.zeros(cap, dtype=np.uint8)
    offs = np.zeros(na + 1, dtype=np.uint64)
    st = L.lib.zbra_hip_encode_frames(edescs, na, frames.ctypes.data_as(C.c_void_p), cap, offs.ctypes.data_as(C.POINTER(C.c_uint64)), None)
    assert st == 0
    widths = []
    for k in range(na):
        a = arrays[k]
        assert a.role == 0 and a.kind == L.ELEM_I64 and a.count == rows and a.block == k // 4
        got = frames[int(offs[k]):int(offs[k + 1])].tobytes()
        assert got == img[a.offset - 8:a.offset + a.len], k
        assert got[8:] == oc.compress_int_array(want_cols[k]), k
        widths.append(got[8 + 12])
    # attribute id <= 8 bits, entity id clustered, time monotone (1 s steps over 2^15 rows), value <= 12 bits
    assert widths[0] == 8 and widths[3] == 12 and widths[2] == 25 and widths[1] <= 10, widths
    outs = [np.full(rows, -1, dtype=np.int64) for _ in range(na)]
    ptrs = (C.c_void_p * na)(*[o.ctypes.data for o in outs])
    statuses = (C.c_int * na)()
    buf = np.frombuffer(img, dtype=np.uint8)
    assert L.lib.zbra_hip_file_decode(buf.ctypes.data_as(C.c_void_p), buf.size, arrays, na, ptrs, statuses) == 0
    assert list(statuses) == [0] * na
    for k in range(na):
        assert np.array_equal(outs[k], want_cols[k]), k
    back = zb.BinaryFile.from_bytes(img)
    assert [b.row_count for b in back.blocks] == [rows] * n_blocks
    assert np.array_equal(dict(back.blocks[2].table.column.fields)["time"].values, want_cols[10])
