"""GPU parity of the device-resident, stream-ordered entry points (the measured path) through
the C ABI on torch-owned device memory: single arrays, batches with mixed widths and ragged
counts, misaligned inputs and outputs, device-side status words, and full-size properties."""
import numpy as np
import pytest

from conftest import KINDS, column

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a device"
    from oracle import zbra_oracle_c as oc
    from zbra_amd import _lib, device
    return torch, device, oc, _lib


def _to_dev(torch, b: bytes, pad_front: int = 0):
    """Device copy of b placed `pad_front` bytes into a fresh allocation (to control alignment)."""
    buf = torch.zeros(pad_front + len(b) + 64, dtype=torch.uint8, device="cuda")
    if len(b):
        buf[pad_front:pad_front + len(b)] = torch.frombuffer(bytearray(b), dtype=torch.uint8).cuda()
    return buf[pad_front:pad_front + len(b)] if len(b) else buf[:0]


@pytest.mark.parametrize("kind", KINDS)
def test_decode_dev_matches_oracle(env, kind):
    torch, device, oc, _ = env
    for n in (1, 2, 2047, 2049, 50001):
        v = column(kind, n, seed=31 + n)
        e = oc.compress_int_array(v)
        for pad in (0, 1, 3, 8, 13):
            d_e = _to_dev(torch, e, pad)
            out = torch.full((n + 2,), -7, dtype=torch.int64, device="cuda")
            res = device.new_results(1, "cuda")
            device.decode_dev(d_e, n, out[1:], res, length=len(e))  # out + 8: not 16-B aligned
            r = device.read_results(res)[0]
            assert r.status == 0 and r.out_count == n and r.width == e[12]
            assert r.midpoint == int.from_bytes(e[:8], "little", signed=True)
            got = out.cpu().numpy()
            assert np.array_equal(got[1:n + 1], v), (kind, n, pad)
            assert got[0] == -7 and got[n + 1] == -7, "wrote outside the output"


def test_decode_dev_status_words(env):
    torch, device, oc, L = env
    e = oc.compress_int_array([1, 2, 3, 4, 5])
    cases = [
        (b"", 5, L.OK, 0),
        (b"\0" * 11, 1, L.ERR_SHORT_HEADER, 0),
        (e[:-1], 5, L.ERR_SHORT_PACKED, 0),
        (e[:8] + (len(e) - 13).to_bytes(4, "little") + e[12:-1], 5, L.ERR_TRUNCATED, 0),
        (b"\x05" + b"\0" * 11, 4, L.OK, 0),
        (e, 5, L.OK, 5),
        (e, 0, L.OK, 0),
    ]
    for data, count, status, out_count in cases:
        d_e = _to_dev(torch, data)
        out = torch.full((8,), -1, dtype=torch.int64, device="cuda")
        res = device.new_results(1, "cuda")
        device.decode_dev(d_e, count, out, res, length=len(data))
        r = device.read_results(res)[0]
        assert (r.status, r.out_count) == (status, out_count), (data.hex(), count)
        if status != L.OK or out_count == 0:
            assert (out.cpu().numpy() == -1).all(), "a rejected / empty array must write nothing"


def test_decode_batch_mixed(env):
    """Mixed widths, raw-fallback arrays, ragged counts, empty and rejected arrays, i64 and u32
    outputs in one submission, packed back to back in one device buffer at odd offsets."""
    torch, device, oc, L = env
    rng = np.random.default_rng(99)
    cols, encs, kinds = [], [], []
    specs = [("narrow", 5000), ("constant", 1), ("full", 4097), ("w31", 2048), ("w32", 2049), ("extremes", 333),
             ("binary", 100000), ("small_signed", 0), ("time_series", 6143), ("clustered", 2047)]
    for i in range(40):
        specs.append(("width_sweep", int(rng.integers(1, 9000))))
    for i, (kind, n) in enumerate(specs):
        v = column(kind, n, seed=i)
        cols.append(v)
        encs.append(oc.compress_int_array(v))
        kinds.append(L.ELEM_I64)
    tags = rng.choice(np.array([0, 1, 2, 7], dtype=np.uint32), 70001, p=[0.5, 0.25, 0.15, 0.1])
    cols.append(tags)
    encs.append(oc.compress_int_array(tags.astype(np.int64)))
    kinds.append(L.ELEM_U32)
    # a rejected array in the middle of the batch must not disturb its neighbours
    bad_at = 4
    blob = bytearray()
    offs = []
    for e in encs:
        blob += b"\xAA" * int(rng.integers(0, 5))  # odd offsets between arrays
        offs.append(len(blob))
        blob += e
    d_blob = _to_dev(torch, bytes(blob))
    outs = []
    items = []
    for i, (v, e) in enumerate(zip(cols, encs)):
        dt = torch.int64 if kinds[i] == L.ELEM_I64 else torch.int32
        o = torch.full((max(v.size, 1),), -3, dtype=dt, device="cuda")
        outs.append(o)
        ln = len(e) - 1 if i == bad_at else len(e)
        items.append((d_blob.data_ptr() + offs[i], ln, v.size, o.data_ptr(), kinds[i]))
    res = device.new_results(len(items), "cuda")
    device.decode_batch_dev(device.make_descs(items), res)
    rs = device.read_results(res)
    for i, (v, e) in enumerate(zip(cols, encs)):
        got = outs[i].cpu().numpy()
        if i == bad_at:
            assert rs[i].status == L.ERR_SHORT_PACKED and (got == -3).all()
            continue
        assert rs[i].status == L.OK and rs[i].out_count == v.size, i
        if v.size:
            assert rs[i].width == e[12]
            expect = v if kinds[i] == L.ELEM_I64 else v.view(np.int32)
            assert np.array_equal(got[: v.size], expect), (i, specs[i] if i < len(specs) else "tags")


def test_encode_dev_matches_oracle(env):
    torch, device, oc, L = env
    for kind in KINDS:
        for n in (1, 2, 2049, 50001):
            v = column(kind, n, seed=77 + n)
            expect = oc.compress_int_array(v)
            d_v = torch.from_numpy(v).cuda()
            cap = device.encode_bound(n)
            out = torch.full((cap + 3 + 32,), 0xEE, dtype=torch.uint8, device="cuda")
            res = device.new_results(1, "cuda")
            device.encode_dev(d_v, out[3:3 + cap], res)  # misaligned destination
            r = device.read_results(res)[0]
            assert r.status == 0 and r.out_count == len(expect), (kind, n)
            got = out.cpu().numpy()
            assert got[3:3 + len(expect)].tobytes() == expect, (kind, n)
            assert (got[:3] == 0xEE).all() and (got[3 + len(expect):] == 0xEE).all(), "wrote outside E"


def test_encode_batch_matches_oracle(env):
    torch, device, oc, L = env
    specs = [("narrow", 5000), ("constant", 1), ("full", 4097), ("w31", 2048), ("w32", 2049), ("extremes", 334),
             ("binary", 100000), ("small_signed", 0), ("time_series", 6143), ("clustered", 2047), ("w40", 12345)]
    cols = [column(k, n, seed=200 + i) for i, (k, n) in enumerate(specs)]
    d_cols = [torch.from_numpy(c).cuda() for c in cols]
    outs = [torch.zeros(max(device.encode_bound(c.size), 1), dtype=torch.uint8, device="cuda") for c in cols]
    items = [(d.data_ptr(), c.size, o.data_ptr(), device.encode_bound(c.size), L.ELEM_I64)
             for c, d, o in zip(cols, d_cols, outs)]
    res = device.new_results(len(items), "cuda")
    device.encode_batch_dev(device.make_encode_descs(items), res)
    rs = device.read_results(res)
    for i, c in enumerate(cols):
        expect = oc.compress_int_array(c)
        assert rs[i].status == 0 and rs[i].out_count == len(expect), specs[i]
        assert outs[i].cpu().numpy()[: len(expect)].tobytes() == expect, specs[i]
    # capacity too small -> status word, nothing written
    small = torch.full((20,), 0xEE, dtype=torch.uint8, device="cuda")
    res1 = device.new_results(1, "cuda")
    device.encode_batch_dev(device.make_encode_descs([(d_cols[0].data_ptr(), cols[0].size, small.data_ptr(), 20, L.ELEM_I64)]), res1)
    assert device.read_results(res1)[0].status == L.ERR_CAPACITY
    assert (small.cpu().numpy() == 0xEE).all()


def test_full_size_roundtrip_properties(env):
    """2^26 values of the headline column (a quarter of the full 2^28, which bench.py runs):
    encode on device, decode on device, exact round trip; the encoded bytes of the first 2^20
    values' worth of stream are compared with the oracle's, and the header fields with an oracle
    header computed from the full column's median (numpy partition)."""
    torch, device, oc, L = env
    n = 1 << 26
    v = oc.gen_narrow(0x5EED0002, 0, n)
    d_v = torch.from_numpy(v).cuda()
    cap = device.encode_bound(n)
    out = torch.zeros(cap, dtype=torch.uint8, device="cuda")
    res = device.new_results(1, "cuda")
    device.encode_dev(d_v, out, res)
    r = device.read_results(res)[0]
    s = np.partition(v, [n // 2 - 1, n // 2])
    a, b = int(s[n // 2 - 1]), int(s[n // 2])
    mid = a + int((b - a) / 2)
    assert r.status == 0 and r.midpoint == mid and r.width == 12 and r.out_count == 13 + n * 12 // 8
    back = torch.zeros(n, dtype=torch.int64, device="cuda")
    res2 = device.new_results(1, "cuda")
    device.decode_dev(out, n, back, res2, length=r.out_count)
    assert device.read_results(res2)[0].out_count == n
    assert torch.equal(back, d_v)
    # width/midpoint fixed -> the stream of the first 2^20 values must equal what the oracle's
    # packer produces for those values with the same midpoint
    m = 1 << 20
    z = oc.zig_zag_encode(v[:m] - mid)
    pk = oc.bp64_pack(np.concatenate([z, np.array([4095], dtype=np.uint64)]))  # pin w = 12
    assert pk[0] == 12
    head = out[13:13 + m * 12 // 8].cpu().numpy().tobytes()
    assert head == pk[1:1 + m * 12 // 8]


def test_decode_garbage_never_reads_or_writes_outside(env):
    """Random and adversarial header bytes: whatever the 13 header bytes claim, the decoder only
    reads inside [data, data + len) and writes inside out[0:count].  Every case is checked against
    the oracle's answer (status, and values when it accepts)."""
    torch, device, oc, L = env
    rng = np.random.default_rng(4242)
    guard = 64
    cases = []
    for _ in range(200):
        ln = int(rng.integers(0, 300))
        data = bytearray(rng.integers(0, 256, ln, dtype=np.uint8).tobytes())
        if ln >= 13 and rng.random() < 0.8:  # plausible packed_len and width so that many cases decode
            pl = int(rng.integers(0, ln - 11))
            data[8:12] = pl.to_bytes(4, "little")
            data[12] = int(rng.choice([0, 1, 2, 7, 12, 31, 32, 33, 64, 65, 200, 255]))
        cases.append((bytes(data), int(rng.integers(0, 400))))
    # packed_len far beyond the buffer, and maximal
    cases.append((bytes(8) + (0xFFFFFFFF).to_bytes(4, "little") + bytes([12]) + bytes(50), 10))
    cases.append((bytes(8) + (51).to_bytes(4, "little") + bytes([31]) + bytes(50), 1 << 20))   # count >> payload
    cases.append((bytes(8) + (51).to_bytes(4, "little") + bytes([64]) + bytes(50), 1000))      # raw, mostly missing
    for data, count in cases:
        rc, expect = oc.decompress_int_array(data, count)
        d = torch.full((guard + len(data) + guard,), 0x5A, dtype=torch.uint8, device="cuda")
        if data:
            d[guard:guard + len(data)] = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
        out = torch.full((count + 2 * 8,), -77, dtype=torch.int64, device="cuda")
        res = device.new_results(1, "cuda")
        device.decode_dev(d[guard:guard + len(data)], count, out[8:8 + count] if count else out[8:8], res, length=len(data))
        r = device.read_results(res)[0]
        got = out.cpu().numpy()
        assert r.status == rc, (data[:16].hex(), len(data), count, r.status, rc)
        assert (got[:8] == -77).all() and (got[8 + count:] == -77).all(), "wrote outside out[0:count]"
        if rc == 0:
            assert r.out_count == expect.size
            assert np.array_equal(got[8:8 + expect.size], expect), (data[:16].hex(), len(data), count)
            assert (got[8 + expect.size:8 + count] == -77).all()
        else:
            assert (got == -77).all()


def test_dev_calls_from_threads_on_own_streams(env):
    """Stream-ordered calls may be issued from several threads on different streams: each borrows
    its own workspace slot (descriptor tables, histograms) and every stream has its own scratch
    buffer (candidate lists), so concurrent submissions must not see each other's state."""
    import threading
    torch, device, oc, L = env
    kinds = ["narrow", "w31", "full", "small_signed", "w40", "binary"]
    n = 300_001
    cols = [column(k, n, 70 + i) for i, k in enumerate(kinds)]
    want = [oc.compress_int_array(c) for c in cols]
    errs = []

    def work(i):
        try:
            st = torch.cuda.Stream()
            v = torch.from_numpy(cols[i]).cuda()
            enc = torch.empty(device.encode_bound(n), dtype=torch.uint8, device="cuda")
            out = torch.empty(n, dtype=torch.int64, device="cuda")
            res, dres = device.new_results(1, "cuda"), device.new_results(1, "cuda")
            torch.cuda.synchronize()
            for _ in range(20):
                device.encode_dev(v, enc, res, stream=st)
                device.decode_dev(enc, n, out, dres, length=len(want[i]), stream=st)
            st.synchronize()
            r = device.read_results(res)[0]
            assert r.status == 0 and r.out_count == len(want[i])
            assert enc[:len(want[i])].cpu().numpy().tobytes() == want[i]
            assert torch.equal(out, v)
        except Exception as ex:  # noqa: BLE001
            errs.append((i, repr(ex)))

    ts = [threading.Thread(target=work, args=(i,)) for i in range(len(kinds))]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs


def test_dev_encodes_from_threads_on_one_stream(env):
    """Encodes enqueued on ONE stream share that stream's scratch buffer (candidate lists of the
    median selection); the library serialises their enqueueing, so calls from several threads on
    the same stream still give exact results, each in its own output.  Sizes differ so that the
    buffer is regrown while other threads use the stream."""
    import threading
    torch, device, oc, L = env
    kinds = ["w31", "full", "w40", "narrow"]
    sizes = [300_001, 1_200_003, 150_000, 2_000_001]
    cols = [column(k, n, 90 + i) for i, (k, n) in enumerate(zip(kinds, sizes))]
    want = [oc.compress_int_array(c) for c in cols]
    st = torch.cuda.Stream()
    vs = [torch.from_numpy(c).cuda() for c in cols]
    encs = [torch.zeros(device.encode_bound(c.size), dtype=torch.uint8, device="cuda") for c in cols]
    ress = [device.new_results(1, "cuda") for _ in cols]
    torch.cuda.synchronize()
    errs = []

    def work(i):
        try:
            for _ in range(10):
                device.encode_dev(vs[i], encs[i], ress[i], stream=st)
        except Exception as ex:  # noqa: BLE001
            errs.append((i, repr(ex)))

    ts = [threading.Thread(target=work, args=(i,)) for i in range(len(cols))]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    st.synchronize()
    assert not errs, errs
    for i in range(len(cols)):
        r = device.read_results(ress[i])[0]
        assert r.status == 0 and r.out_count == len(want[i]), (i, r)
        assert encs[i][:len(want[i])].cpu().numpy().tobytes() == want[i], i


def test_dev_encodes_on_more_streams_than_keep_scratch(env):
    """Only 16 streams keep a scratch buffer; a seventeenth takes over the least recently used one
    after a device synchronisation.  Twenty streams, twice round, every output exact."""
    torch, device, oc, L = env
    n = 200_003
    col = column("w31", n, 123)
    want = oc.compress_int_array(col)
    v = torch.from_numpy(col).cuda()
    streams = [torch.cuda.Stream() for _ in range(20)]
    encs = [torch.zeros(device.encode_bound(n), dtype=torch.uint8, device="cuda") for _ in streams]
    ress = [device.new_results(1, "cuda") for _ in streams]
    torch.cuda.synchronize()
    for _ in range(2):
        for st, enc, res in zip(streams, encs, ress):
            device.encode_dev(v, enc, res, stream=st)
    torch.cuda.synchronize()
    for enc, res in zip(encs, ress):
        r = device.read_results(res)[0]
        assert r.status == 0 and r.out_count == len(want)
        assert enc[:len(want)].cpu().numpy().tobytes() == want


@pytest.mark.parametrize("dtype_name", ["int32", "int64"])
def test_more_than_2_31_values(env, dtype_name):
    """One array of 2^31 + 5 values: every value index, bit offset and count beyond 32 bits
    (`values.len() as u32` allows up to 2^32 - 1, binary.rs:640).  Device-resident, 8 / 16 GiB."""
    torch, device, _, L = env
    dt = getattr(torch, dtype_name)
    n = (1 << 31) + 5
    v = torch.empty(n, dtype=dt, device="cuda")
    step = 1 << 28
    for i0 in range(0, n, step):
        i1 = min(n, i0 + step)
        idx = torch.arange(i0, i1, dtype=torch.int64, device="cuda")
        v[i0:i1] = ((idx * 2654435761) % 1000 + (idx >> 30)).to(dt)   # < 1003: width 10 or 11; the tail differs from the head
        del idx
    enc = torch.empty(13 + 2 * n + 64, dtype=torch.uint8, device="cuda")     # far more than 11 bits per value
    res = device.new_results(1, "cuda")
    device.encode_dev(v, enc, res)
    r = device.read_results(res)[0]
    srt = torch.sort(v[: 1 << 20]).values  # the values are equidistributed: a prefix pins the median well enough to bound it
    assert r.status == 0 and r.width in (10, 11) and r.out_count == 13 + (n * r.width + 7) // 8
    assert int(srt[0]) <= r.midpoint <= int(srt[-1]) + 2
    del srt
    # exact median: count the values below and above the reported midpoint
    below = int((v < r.midpoint).sum()); above = int((v > r.midpoint).sum())
    assert below <= (n - 1) // 2 and above <= n // 2, (below, above)
    out = torch.empty_like(v)
    dres = device.new_results(1, "cuda")
    device.decode_dev(enc, n, out, dres, length=int(r.out_count))
    d = device.read_results(dres)[0]
    assert d.status == 0 and d.out_count == n and d.width == r.width
    assert torch.equal(out, v)


def test_encode_dev_misaligned_inputs(env):
    """Values that start 8 bytes (int64) or 4 / 8 / 12 bytes (u32 tags) past a 16-byte boundary:
    the tile loads are 16 bytes per lane and must not care."""
    torch, device, oc, L = env
    n = 3 * 2048 + 77
    big = torch.from_numpy(column("narrow", n + 4, seed=5)).cuda()
    tags_np = (np.random.default_rng(6).integers(0, 9, n + 4, dtype=np.int64) % 4).astype(np.int32)
    tags = torch.from_numpy(tags_np).cuda()
    for v, shifts in ((big, (0, 1)), (tags, (0, 1, 2, 3))):
        for s in shifts:
            part = v[s:s + n]
            assert part.data_ptr() % 16 == (s * v.element_size()) % 16
            expect = oc.compress_int_array(part.cpu().numpy().astype(np.int64))
            out = torch.zeros(device.encode_bound(n), dtype=torch.uint8, device="cuda")
            res = device.new_results(1, "cuda")
            device.encode_dev(part, out, res)
            r = device.read_results(res)[0]
            assert r.status == 0 and r.out_count == len(expect)
            assert out[: len(expect)].cpu().numpy().tobytes() == expect, (v.dtype, s)
