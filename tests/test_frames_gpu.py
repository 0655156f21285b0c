"""GPU parity of the framed, batched encode (write_int_array_compressed for many arrays at once,
binary.rs:639-645, 666-669): the frames must be the bytes the oracle's framing writes, back to
back, whatever the mix of lengths, widths and element kinds."""
import ctypes as C

import numpy as np
import pytest

from conftest import KINDS, column

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a device"
    from oracle import zbra_oracle_c as oc
    from zbra_amd import _lib, compression as zc, device
    return torch, oc, zc, device, _lib


def _mixed_arrays():
    arrs = []
    for i, kind in enumerate(KINDS):
        for n in (0, 1, 9, 2049, 70001):
            arrs.append(column(kind, n, 900 + 7 * i + n % 13))
    rng = np.random.default_rng(1)
    arrs.append(rng.integers(0, 5, 300001, dtype=np.int64).astype(np.uint32))   # enum tags
    arrs.append(np.zeros(0, dtype=np.uint32))
    arrs.append((rng.integers(0, 1 << 32, 5000, dtype=np.int64)).astype(np.uint32))
    arrs.append(column("narrow", (1 << 21) + 3, 77))
    return arrs


def _want(oc, arrs):
    frames = [oc.write_int_array_compressed(a.astype(np.int64)) for a in arrs]
    offs = np.concatenate([[0], np.cumsum([len(f) for f in frames])])
    return b"".join(frames), offs


def test_host_frames_match_oracle(env):
    _, oc, zc, _, _ = env
    arrs = _mixed_arrays()
    want, offs = _want(oc, arrs)
    got, got_offs = zc.write_int_arrays_compressed(arrs)
    assert np.array_equal(got_offs, offs)
    assert got == want
    empty, empty_offs = zc.write_int_arrays_compressed([])
    assert empty == b"" and list(empty_offs) == [0]


def test_many_small_arrays(env):
    """More arrays than one scan step of the layout kernel handles (256), ragged and empty ones mixed in."""
    _, oc, zc, _, _ = env
    rng = np.random.default_rng(5)
    arrs = []
    for i in range(1500):
        n = int(rng.integers(0, 40)) if i % 7 else int(rng.integers(2000, 6000))
        arrs.append(column(KINDS[i % len(KINDS)], n, 3000 + i))
    want, offs = _want(oc, arrs)
    got, got_offs = zc.write_int_arrays_compressed(arrs)
    assert np.array_equal(got_offs, offs) and got == want


def test_host_frames_capacity(env):
    _, oc, zc, _, L = env
    arrs = [column("narrow", 5000, 1), column("w40", 3000, 2), column("small_signed", 100, 3), np.zeros(0, dtype=np.int64)]
    want, offs = _want(oc, arrs)
    vs = [np.ascontiguousarray(a) for a in arrs]
    descs = (L.EncodeDesc * len(vs))()
    for i, v in enumerate(vs):
        descs[i].values = v.ctypes.data if v.size else None
        descs[i].n = v.size
        descs[i].kind = L.ELEM_I64
    cap = int(offs[2]) - 1                                   # array 1 misses one byte
    out = np.full(len(want) + 64, 0xAB, dtype=np.uint8)
    offsets = (C.c_uint64 * (len(vs) + 1))()
    statuses = (C.c_int * len(vs))()
    st = L.lib.zbra_hip_encode_frames(descs, len(vs), out.ctypes.data_as(C.c_void_p), cap, offsets, statuses)
    assert st == L.ERR_CAPACITY
    # frames are laid out in order: everything after the first frame that does not fit fails too
    # (even the 8-byte empty frame) and is reported as the empty range at the end of what was written
    assert list(statuses) == [0, L.ERR_CAPACITY, L.ERR_CAPACITY, L.ERR_CAPACITY]
    assert list(offsets) == [0] + [int(offs[1])] * 4
    assert out[: offs[1]].tobytes() == want[: offs[1]]
    assert (out[offs[1]:] == 0xAB).all(), "bytes of frames that were not laid out reached the caller"
    # exact capacity is enough
    st = L.lib.zbra_hip_encode_frames(descs, len(vs), out.ctypes.data_as(C.c_void_p), len(want), offsets, statuses)
    assert st == 0 and list(offsets) == list(offs) and out[: len(want)].tobytes() == want
    assert (out[len(want):] == 0xAB).all()


def test_host_frames_many_rounds_and_capacities(env):
    """Enough values (about 270 MB) that the host write path cuts them into several rounds on its two
    lanes, with a second thread copying frames back while the next round goes up.  The frames must be
    the oracle's whatever the capacity allows the rounds to do: run ahead (generous capacity), wait
    for each other (exact capacity), or stop at the first frame that does not fit (in a middle round:
    every later array fails, nothing of it reaches the caller)."""
    _, oc, zc, _, L = env
    rng = np.random.default_rng(12)
    arrs = [column("narrow", 5_000_003, 1), column("time_series", 4_194_304, 2), column("w40", 3_000_000, 3), np.zeros(0, dtype=np.int64),
            column("constant", 6_000_000, 4), rng.choice(np.array([0, 1, 2, 7], dtype=np.uint32), 9_000_001), column("small_signed", 4_500_000, 5),
            column("full", 2_000_000, 6), column("clustered", 5_000_000, 7), column("binary", 1, 8)]
    want, offs = _want(oc, arrs)
    n = len(arrs)
    vs = [np.ascontiguousarray(a) for a in arrs]
    descs = (L.EncodeDesc * n)()
    for i, v in enumerate(vs):
        descs[i].values = v.ctypes.data if v.size else None
        descs[i].n = v.size
        descs[i].kind = L.ELEM_U32 if v.dtype == np.uint32 else L.ELEM_I64
    bound = sum(8 + L.lib.zbra_hip_encode_bound(v.size) for v in vs)
    out = np.empty(bound + 64, dtype=np.uint8)
    offsets = (C.c_uint64 * (n + 1))()
    statuses = (C.c_int * n)()

    def run(cap):
        out[:] = 0xAB
        for i in range(n):
            statuses[i] = -1
        return L.lib.zbra_hip_encode_frames(descs, n, out.ctypes.data_as(C.c_void_p), cap, offsets, statuses)

    for cap in (bound, len(want), len(want) + 12345):          # rounds run ahead / wait for each other / a mix
        assert run(cap) == 0
        assert list(statuses) == [0] * n and list(offsets) == [int(o) for o in offs]
        assert out[: len(want)].tobytes() == want and (out[len(want):] == 0xAB).all(), cap
    for k in (6, 2, 0):                                       # frame k misses one byte
        cap = int(offs[k + 1]) - 1
        assert run(cap) == L.ERR_CAPACITY
        assert list(statuses) == [0] * k + [L.ERR_CAPACITY] * (n - k), k
        assert list(offsets) == [int(o) for o in offs[: k + 1]] + [int(offs[k])] * (n - k), k
        assert out[: offs[k]].tobytes() == want[: offs[k]], k
        assert (out[offs[k]:] == 0xAB).all(), "bytes of frames that were not laid out reached the caller"


def test_frames_dev_misaligned_with_guards(env):
    torch, oc, _, device, L = env
    arrs = [column(k, n, 40 + i) for i, (k, n) in enumerate([("narrow", 4097), ("full", 2048), ("binary", 1), ("constant", 0),
                                                             ("w31", 10001), ("time_series", 5000)])]
    want, offs = _want(oc, arrs)
    d_vals = [torch.from_numpy(a).cuda() for a in arrs]
    descs = device.make_encode_descs([(v.data_ptr() if v.numel() else None, v.numel(), None, 0, L.ELEM_I64) for v in d_vals])
    for pad in (0, 1, 5):
        buf = torch.full((pad + len(want) + 32,), 0xCD, dtype=torch.uint8, device="cuda")
        d_off = torch.zeros(len(arrs) + 1, dtype=torch.int64, device="cuda")
        res = device.new_results(len(arrs), "cuda")
        device.encode_frames_dev(descs, buf[pad:pad + len(want)], d_off, res)
        torch.cuda.synchronize()
        assert np.array_equal(d_off.cpu().numpy(), offs)
        got = buf.cpu().numpy()
        assert got[pad:pad + len(want)].tobytes() == want, pad
        assert (got[:pad] == 0xCD).all() and (got[pad + len(want):] == 0xCD).all(), "wrote outside the frames"
        rs = device.read_results(res)
        for i, a in enumerate(arrs):
            assert rs[i].status == 0 and rs[i].out_count == offs[i + 1] - offs[i] - 8


def test_host_batch_decode_of_scattered_arrays(env):
    """zbra_hip_decode_batch: compressed arrays in separate host buffers (what read_exact per array
    leaves behind), mixed kinds and sizes, an empty one, and a corrupted one whose error must not
    disturb the rest."""
    _, oc, zc, _, L = env
    arrs = _mixed_arrays()
    items = [(oc.compress_int_array(a.astype(np.int64)), a.size, a.dtype) for a in arrs]
    got = zc.read_int_arrays_compressed(items)
    for g, a in zip(got, arrs):
        assert g.dtype == a.dtype and np.array_equal(g, a)
    # count larger than what was encoded for an empty array: empty result, not an error
    assert zc.read_int_arrays_compressed([(b"", 7, np.int64)])[0].size == 0
    # a 12-byte E with packed_len == 0 yields an empty Vec whatever count says (compression.rs:182-184
    # through :318-321), also in the middle of a batch and for u32 outputs
    hollow = b"\x05" + b"\0" * 11
    assert oc.decompress_int_array(hollow, 9)[1].size == 0
    mixed = zc.read_int_arrays_compressed([items[0], (hollow, 9, np.int64), (hollow + b"\xee\xee", 5, np.uint32), items[1]])
    assert [m.size for m in mixed] == [arrs[0].size, 0, 0, arrs[1].size]
    assert np.array_equal(mixed[0], arrs[0]) and np.array_equal(mixed[3], arrs[1])
    assert zc.decoded_count(hollow, 9) == 0 and zc.decoded_count(items[0][0], arrs[0].size) == arrs[0].size
    bad = list(items)
    k = next(i for i, it in enumerate(items) if len(it[0]) > 20)
    e = bytearray(items[k][0])
    e[8:12] = (len(e)).to_bytes(4, "little")              # packed_len beyond the bytes that are there
    bad[k] = (bytes(e), items[k][1], items[k][2])
    with pytest.raises(zc.DecompressionError, match="Insufficient data"):
        zc.read_int_arrays_compressed(bad)
    # the C call reports per-array statuses and still decodes the others
    descs = (L.ArrayDesc * len(bad))()
    keep, outs = [], []
    for i, (data, count, dtype) in enumerate(bad):
        d = np.frombuffer(data, dtype=np.uint8)
        o = np.full(count, 77, dtype=dtype)
        keep.append(d); outs.append(o)
        descs[i].data = d.ctypes.data if d.size else None
        descs[i].len = d.size; descs[i].count = count
        descs[i].out = o.ctypes.data if o.size else None
        descs[i].kind = L.ELEM_U32 if o.dtype == np.uint32 else L.ELEM_I64
    statuses = (C.c_int * len(bad))()
    st = L.lib.zbra_hip_decode_batch(descs, len(bad), statuses)
    assert st == L.ERR_SHORT_PACKED and statuses[k] == L.ERR_SHORT_PACKED and sum(1 for s in statuses if s) == 1
    assert (outs[k] == 77).all()
    for i, a in enumerate(arrs):
        if i != k:
            assert np.array_equal(outs[i], a), i
