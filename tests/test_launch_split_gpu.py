"""A dispatch holds at most 2^32 - 1 work-items, and a larger launch is not refused: it silently
runs fewer workgroups.  One workgroup per 16 KiB tile meets that limit at 137 GB of decoded values
in one submission (a 64 GB .zbra file decoded in HBM passes it), so the library cuts such a
submission into several launches (decode) or lets workgroups take several tiles (pack).  These
tests lower the limit through the tuning hook so that small inputs reach both, and check the
results against the oracle."""
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


@pytest.fixture()
def tight_limit(env):
    import ctypes as C
    _, _, _, L = env
    L.lib.zbra_tuning_set_launch_limit.argtypes = [C.c_uint64]
    L.lib.zbra_tuning_set_launch_limit.restype = C.c_int

    def set_limit(work_items):
        assert L.lib.zbra_tuning_set_launch_limit(int(work_items)) == 0
    yield set_limit
    L.lib.zbra_tuning_set_launch_limit(0)  # back to the hardware's limit


def _batch(env, specs, seed):
    torch, device, oc, L = env
    rng = np.random.default_rng(seed)
    cols, encs, kinds = [], [], []
    for i, (kind, n) in enumerate(specs):
        if kind == "tags":
            v = rng.choice(np.array([0, 1, 2, 7], dtype=np.uint32), n, p=[0.5, 0.25, 0.15, 0.1])
            cols.append(v); encs.append(oc.compress_int_array(v.astype(np.int64))); kinds.append(L.ELEM_U32)
        else:
            v = column(kind, n, seed=seed + i)
            cols.append(v); encs.append(oc.compress_int_array(v)); kinds.append(L.ELEM_I64)
    return cols, encs, kinds


SPECS = [("narrow", 70000), ("w31", 2048), ("full", 40001), ("tags", 150001), ("constant", 1), ("time_series", 123457),
         ("small_signed", 0), ("w32", 8192), ("clustered", 99999), ("width_sweep", 31000), ("tags", 4097)]


@pytest.mark.parametrize("limit", [512 * 8, 512 * 24, 512 * 1000])
def test_decode_batch_cut_into_launches(env, tight_limit, limit):
    torch, device, oc, L = env
    cols, encs, kinds = _batch(env, SPECS, seed=5)
    blob, offs = bytearray(), []
    for e in encs:
        blob += b"\x55" * (len(blob) % 3)
        offs.append(len(blob))
        blob += e
    d_blob = torch.frombuffer(bytearray(bytes(blob) + b"\0" * 64), dtype=torch.uint8).cuda()
    outs, items = [], []
    for i, (v, e) in enumerate(zip(cols, encs)):
        o = torch.full((v.size + 2,), -3, dtype=torch.int64 if kinds[i] == L.ELEM_I64 else torch.int32, device="cuda")
        outs.append(o)
        items.append((d_blob.data_ptr() + offs[i], len(e), v.size, o.data_ptr() + o.element_size(), kinds[i]))
    tight_limit(limit)
    res = device.new_results(len(items), "cuda")
    device.decode_batch_dev(device.make_descs(items), res)
    # the entry point that allocates nothing shares the launch code
    outs2 = [torch.full_like(o, -3) for o in outs]
    items2 = [(it[0], it[1], it[2], o.data_ptr() + o.element_size(), it[4]) for it, o in zip(items, outs2)]
    device.decode_batch_dev_noalloc(device.make_descs(items2), None)
    rs = device.read_results(res)
    for i, v in enumerate(cols):
        assert rs[i].status == L.OK and rs[i].out_count == v.size, i
        expect = v if kinds[i] == L.ELEM_I64 else v.view(np.int32)
        for got in (outs[i].cpu().numpy(), outs2[i].cpu().numpy()):
            assert np.array_equal(got[1:v.size + 1], expect), (i, SPECS[i], limit)
            assert got[0] == -3 and got[-1] == -3, "wrote outside the output"


def test_single_array_cut_into_launches(env, tight_limit):
    torch, device, oc, L = env
    for kind, n in (("narrow", 300001), ("full", 65537), ("w31", 100000)):
        v = column(kind, n, seed=n)
        e = oc.compress_int_array(v)
        d_e = torch.frombuffer(bytearray(e), dtype=torch.uint8).cuda()
        out = torch.full((n + 2,), -9, dtype=torch.int64, device="cuda")
        tight_limit(512 * 16)
        device.decode_dev(d_e, n, out[1:], None)
        got = out.cpu().numpy()
        assert np.array_equal(got[1:n + 1], v), kind
        assert got[0] == -9 and got[n + 1] == -9


@pytest.mark.parametrize("limit", [256 * 3, 256 * 64])
def test_pack_grid_capped(env, tight_limit, limit):
    torch, device, oc, L = env
    cols, encs, kinds = _batch(env, SPECS, seed=11)
    tight_limit(limit)
    d_vals = [torch.from_numpy(v if k == L.ELEM_I64 else v.view(np.int32)).cuda() for v, k in zip(cols, kinds)]
    d_outs = [torch.full((device.encode_bound(v.size) + 16,), 0xEE, dtype=torch.uint8, device="cuda") for v in cols]
    items = [(dv.data_ptr(), v.size, do.data_ptr(), device.encode_bound(v.size), k) for dv, v, do, k in zip(d_vals, cols, d_outs, kinds)]
    res = device.new_results(len(items), "cuda")
    device.encode_batch_dev(device.make_encode_descs(items), res)
    rs = device.read_results(res)
    for i, e in enumerate(encs):
        assert rs[i].status == L.OK and rs[i].out_count == len(e), i
        got = d_outs[i].cpu().numpy()
        assert got[:len(e)].tobytes() == e, (i, SPECS[i])
        assert (got[len(e):] == 0xEE).all()
    # a single long array: its pack launch is capped too
    v = column("narrow", 500001, seed=3)
    e = oc.compress_int_array(v)
    d_v = torch.from_numpy(v).cuda()
    out = torch.full((device.encode_bound(v.size),), 0xEE, dtype=torch.uint8, device="cuda")
    r1 = device.new_results(1, "cuda")
    device.encode_dev(d_v, out, r1)
    assert device.read_results(r1)[0].out_count == len(e)
    assert out.cpu().numpy()[:len(e)].tobytes() == e
