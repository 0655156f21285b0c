"""zbra_hip_decode_batch_dev_noalloc (the entry point that allocates nothing) against
zbra_hip_decode_batch_dev on the same submissions, through the C ABI: bit-identical values and
result words, on two calls in a row of which the second is smaller (the library's fixed scratch
keeps the first call's contents), and a clear error beyond its limits."""
import ctypes as C

import numpy as np
import pytest

from conftest import column

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a device"
    from oracle import zbra_oracle_c as oc
    from zbra_amd import _lib, device
    return torch, device, oc, _lib


def _submission(torch, oc, L, specs, seed):
    """Encoded arrays back to back in one device buffer at odd offsets; returns (blob, items builder)."""
    rng = np.random.default_rng(seed)
    cols, encs, kinds = [], [], []
    for i, (kind, n) in enumerate(specs):
        if kind == "tags":
            v = rng.choice(np.array([0, 1, 2, 7], dtype=np.uint32), n, p=[0.5, 0.25, 0.15, 0.1])
            cols.append(v); encs.append(oc.compress_int_array(v.astype(np.int64))); kinds.append(L.ELEM_U32)
        else:
            v = column(kind, n, seed=seed * 1000 + i)
            cols.append(v); encs.append(oc.compress_int_array(v)); kinds.append(L.ELEM_I64)
    blob, offs = bytearray(), []
    for e in encs:
        blob += b"\xAA" * int(rng.integers(0, 5))
        offs.append(len(blob))
        blob += e
    blob += b"\0" * 64
    d_blob = torch.frombuffer(blob, dtype=torch.uint8).cuda()
    return cols, encs, kinds, offs, d_blob


def _run(torch, device, L, fn, cols, encs, kinds, offs, d_blob, bad_at=None):
    outs, items = [], []
    for i, (v, e) in enumerate(zip(cols, encs)):
        dt = torch.int64 if kinds[i] == L.ELEM_I64 else torch.int32
        o = torch.full((v.size + 2,), -3, dtype=dt, device="cuda")
        outs.append(o)
        ln = len(e) - 1 if i == bad_at else len(e)
        items.append((d_blob.data_ptr() + offs[i], ln, v.size, o.data_ptr() + o.element_size(), kinds[i]))
    res = device.new_results(len(items), "cuda")
    fn(device.make_descs(items), res)
    torch.cuda.synchronize()
    return [o.cpu().numpy() for o in outs], res.cpu().numpy().tobytes()


BIG = [("narrow", 300001), ("constant", 1), ("full", 4097), ("w31", 2048), ("w32", 2049), ("extremes", 333), ("binary", 100000),
       ("small_signed", 0), ("time_series", 6143), ("clustered", 2047), ("tags", 70001), ("w40", 12345)] + \
      [("width_sweep", 1 + 37 * k) for k in range(120)]          # 132 arrays: three chunks of kernel-argument descriptors
SMALL = [("w31", 5000), ("tags", 4099), ("narrow", 17)]


def test_noalloc_equals_batch_dev_twice_second_smaller(env):
    torch, device, oc, L = env
    assert L.lib.zbra_hip_init() == 0
    for specs, seed, bad_at in ((BIG, 1, 4), (SMALL, 2, None)):
        sub = _submission(torch, oc, L, specs, seed)
        want_outs, want_res = _run(torch, device, L, device.decode_batch_dev, *sub, bad_at=bad_at)
        got_outs, got_res = _run(torch, device, L, device.decode_batch_dev_noalloc, *sub, bad_at=bad_at)
        assert got_res == want_res
        cols = sub[0]
        for i, (g, w) in enumerate(zip(got_outs, want_outs)):
            assert np.array_equal(g, w), (specs[i], i)
            if i != bad_at:  # and both equal the oracle's input, guards intact
                v = cols[i] if cols[i].dtype == np.int64 else cols[i].view(np.int32)
                assert np.array_equal(g[1:-1], v) and g[0] == -3 and g[-1] == -3, (specs[i], i)
            else:
                assert (g == -3).all()


def test_noalloc_single_array_and_empty_submission(env):
    torch, device, oc, L = env
    sub = _submission(torch, oc, L, [("narrow", 1 << 21)], 3)
    want_outs, want_res = _run(torch, device, L, device.decode_batch_dev, *sub)
    got_outs, got_res = _run(torch, device, L, device.decode_batch_dev_noalloc, *sub)
    assert got_res == want_res and np.array_equal(got_outs[0], want_outs[0])
    assert L.lib.zbra_hip_decode_batch_dev_noalloc(None, 0, None, None) == 0


def test_noalloc_rejects_what_its_scratch_cannot_hold(env):
    torch, device, oc, L = env
    out = torch.zeros(8, dtype=torch.int64, device="cuda")
    e = torch.zeros(64, dtype=torch.uint8, device="cuda")
    too_many = device.make_descs([(e.data_ptr(), 0, 0, out.data_ptr(), L.ELEM_I64)] * (L.NOALLOC_MAX_ARRAYS + 1))
    assert L.lib.zbra_hip_decode_batch_dev_noalloc(too_many, len(too_many), None, None) == L.ERR_NOALLOC_LIMIT
    # tiles: 2^20 tiles of 2048 int64 values fit, one more value does not (nothing is launched, so
    # the pointers are never touched)
    fits = device.make_descs([(e.data_ptr(), 0, 2048 << 19, out.data_ptr(), L.ELEM_I64)] * 2)
    assert L.lib.zbra_hip_decode_batch_dev_noalloc(fits, 2, None, None) == 0      # len 0: empty data, nothing written
    over = device.make_descs([(e.data_ptr(), 0, 2048 << 19, out.data_ptr(), L.ELEM_I64), (e.data_ptr(), 0, (2048 << 19) + 1, out.data_ptr(), L.ELEM_I64)])
    assert L.lib.zbra_hip_decode_batch_dev_noalloc(over, 2, None, None) == L.ERR_NOALLOC_LIMIT
    with pytest.raises(ValueError, match="no-allocation"):
        device.decode_batch_dev_noalloc(over)
    torch.cuda.synchronize()
    assert "scratch" in L.strerror(L.ERR_NOALLOC_LIMIT)


def test_noalloc_under_graph_capture_and_replay(env):
    """The callers the entry point is for: a stream under HIP graph capture, where an allocation, a
    copy from pageable memory or a synchronisation fails the capture.  One call is captured after
    zbra_hip_init(), then replayed twice over CHANGED encoded bytes (same frame lengths, so the
    descriptors baked into the graph still hold); each replay must produce the oracle's values, and
    the device's free memory must be what it was before the capture."""
    torch, device, oc, L = env
    assert L.lib.zbra_hip_init() == 0
    specs = [("narrow", 200001), ("tags", 70001), ("w31", 40961), ("narrow", 2049)]
    subs = [_submission(torch, oc, L, specs, seed) for seed in (11, 12, 13)]
    # same layout for all three data sets: equal encoded lengths, blob rebuilt at the first one's offsets
    lens = [[len(e) for e in sub[1]] for sub in subs]
    assert lens[0] == lens[1] == lens[2], lens
    offs = subs[0][3]
    blobs = []
    for sub in subs:
        b = bytearray(subs[0][4].numel())
        for e, o in zip(sub[1], offs):
            b[o:o + len(e)] = e
        blobs.append(torch.frombuffer(b, dtype=torch.uint8).cuda())
    d_blob = blobs[0].clone()
    cols0, _, kinds, _, _ = subs[0]
    outs, items = [], []
    for i, v in enumerate(cols0):
        o = torch.full((v.size + 2,), -3, dtype=torch.int64 if kinds[i] == L.ELEM_I64 else torch.int32, device="cuda")
        outs.append(o)
        items.append((d_blob.data_ptr() + offs[i], lens[0][i], v.size, o.data_ptr() + o.element_size(), kinds[i]))
    descs = device.make_descs(items)
    res = device.new_results(len(items), "cuda")
    torch.cuda.synchronize()
    free_before = torch.cuda.mem_get_info()[0]
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=torch.cuda.Stream()):
        device.decode_batch_dev_noalloc(descs, res)          # on the capturing stream
    for o in outs:
        assert (o == -3).all()                                # captured, not run
    for k in (1, 2, 0):
        d_blob.copy_(blobs[k])
        for o in outs:
            o.fill_(-3)
        graph.replay()
        torch.cuda.synchronize()
        for i, v in enumerate(subs[k][0]):
            got = outs[i].cpu().numpy()
            want = v if v.dtype == np.int64 else v.view(np.int32)
            assert np.array_equal(got[1:-1], want) and got[0] == -3 and got[-1] == -3, (k, specs[i])
        for r in device.read_results(res):
            assert r.status == 0
    assert torch.cuda.mem_get_info()[0] >= free_before - (4 << 20)   # the graph object itself is all that was added
