"""GPU parity of the Enum row split (zbra_hip_enum_rows_dev) against the restatement of the
reference's row loop (Column::from_values, striped.rs:340-387): every row's variant index and its
position in the variant's column, the variants' row counts, the unknown-tag error with its row."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a device"
    from oracle import zbra_oracle_np as onp
    from zbra_amd import _lib, device
    return torch, device, onp, _lib


def _run(torch, device, tags, vt, pad=0):
    n = tags.size
    # `pad` elements in front: the tag column then starts off a 16-byte boundary
    d_tags = torch.from_numpy(np.concatenate([np.zeros(pad, np.uint32), tags.astype(np.uint32)]).view(np.int32)).cuda()[pad:]
    variant = torch.full((n + 2,), -5, dtype=torch.int32, device="cuda")
    rank = torch.full((n + 2,), -5, dtype=torch.int64, device="cuda")
    counts = torch.full((len(vt) + 2,), -5, dtype=torch.int64, device="cuda")
    res = device.new_results(1, "cuda")
    device.enum_rows_dev(d_tags, vt, variant[1:n + 1], rank[1:n + 1], counts[1:len(vt) + 1], res)
    r = device.read_results(res)[0]
    variant, rank, counts = variant.cpu().numpy(), rank.cpu().numpy(), counts.cpu().numpy()
    assert variant[0] == -5 and variant[n + 1] == -5 and rank[0] == -5 and rank[n + 1] == -5, "wrote outside the outputs"
    assert counts[0] == -5 and counts[len(vt) + 1] == -5
    return r, variant[1:n + 1].view(np.uint32), rank[1:n + 1].view(np.uint64), counts[1:len(vt) + 1].view(np.uint64)


@pytest.mark.parametrize("n", [0, 1, 63, 64, 65, 511, 512, 513, 2047, 2048, 2049, 100003, (1 << 21) + 77, 30_000_000])
def test_rows_match_reference_loop(env, n):
    torch, device, onp, _ = env
    rng = np.random.default_rng(n + 1)
    shapes = [([0, 1, 2, 7], rng.choice([0, 1, 2, 7], n, p=[.5, .25, .15, .1])),             # the sum-typed table's tags
              ([5], np.full(n, 5)),                                                         # one variant
              (list(range(32)), rng.integers(0, 32, n)),                                    # as many variants as fit
              ([3, 900, 1024, 70000, 0xFFFFFFFF, 3], rng.choice([3, 900, 1024, 70000, 0xFFFFFFFF], n))]  # searched tags; a duplicate variant never wins
    if n:
        runs = np.repeat(rng.integers(0, 3, n // 700 + 1), 700)[:n]                          # long runs of one variant
        shapes.append(([0, 1, 2], runs))
    for vt, tags in shapes:
        tags = np.asarray(tags, dtype=np.uint64).astype(np.uint32)
        st, variant, rank, counts, _ = onp.enum_rows(tags, vt)
        assert st == 0
        r, got_v, got_r, got_c = _run(torch, device, tags, vt, pad=n % 4)
        assert r.status == 0 and r.out_count == n
        assert np.array_equal(got_v, variant), (n, vt)
        assert np.array_equal(got_r, rank), (n, vt)
        assert np.array_equal(got_c, counts), (n, vt)
        # what the pair is for: scattering a payload column by (variant, rank) rebuilds the variant columns
        if n and n <= 100003:
            payload = rng.integers(-1 << 40, 1 << 40, n)
            for v in range(len(vt)):
                col = np.empty(int(got_c[v]), dtype=np.int64)
                m = got_v == v
                col[got_r[m].astype(np.int64)] = payload[m]
                assert np.array_equal(col, payload[variant == v])


def test_small_cases_against_the_statement_by_statement_loop(env):
    torch, device, onp, _ = env
    rng = np.random.default_rng(9)
    for trial in range(40):
        nv = int(rng.integers(1, 9))
        vt = [int(x) for x in rng.integers(0, 12, nv)]       # duplicates among the variants happen
        n = int(rng.integers(1, 400))
        tags = rng.choice(vt, n).astype(np.uint32)
        st, variant, rank, counts, _ = onp.enum_rows_loop(tags, vt)
        assert st == 0
        r, got_v, got_r, got_c = _run(torch, device, tags, vt)
        assert r.status == 0
        assert got_v.tolist() == variant and got_r.tolist() == rank and got_c.tolist() == counts


def test_unknown_tag_is_reported_with_its_row(env):
    torch, device, onp, L = env
    rng = np.random.default_rng(3)
    for n, bad_rows in ((10, [4]), (5000, [4999]), (300000, [77777, 12345, 299999]), (2048, [0])):
        tags = rng.choice([0, 1, 2, 7], n).astype(np.uint32)
        for b in bad_rows:
            tags[b] = 9
        st, _, _, _, first = onp.enum_rows(tags, [0, 1, 2, 7])
        assert st == 8 and first == min(bad_rows)
        r, _, _, _ = _run(torch, device, tags, [0, 1, 2, 7])
        assert r.status == L.ERR_UNKNOWN_TAG and r.out_count == first and r.midpoint == 9
    # no variants at all: the first row is already unknown
    r, _, _, _ = _run(torch, device, np.array([1, 2], dtype=np.uint32), [])
    assert r.status == L.ERR_UNKNOWN_TAG and r.out_count == 0 and r.midpoint == 1


def test_invalid_arguments(env):
    torch, device, onp, L = env
    tags = torch.zeros(8, dtype=torch.int32, device="cuda")
    variant = torch.zeros(8, dtype=torch.int32, device="cuda")
    rank = torch.zeros(8, dtype=torch.int64, device="cuda")
    counts = torch.zeros(40, dtype=torch.int64, device="cuda")
    with pytest.raises(ValueError):                                   # more variants than the kernels take
        device.enum_rows_dev(tags, list(range(33)), variant, rank, counts)


def _run_batch(torch, device, arrays, vt):
    """One batched submission over `arrays` (numpy uint32 tag arrays, one variant list): per array
    (result, variant, rank, counts), guard words checked."""
    bufs = []
    for k, tags in enumerate(arrays):
        n, pad = tags.size, k % 4
        d_tags = torch.from_numpy(np.concatenate([np.zeros(pad, np.uint32), tags.astype(np.uint32)]).view(np.int32)).cuda()[pad:]
        bufs.append((d_tags, torch.full((n + 2,), -5, dtype=torch.int32, device="cuda"), torch.full((n + 2,), -5, dtype=torch.int64, device="cuda"),
                     torch.full((len(vt) + 2,), -5, dtype=torch.int64, device="cuda")))
    res = device.new_results(len(arrays), "cuda")
    device.enum_rows_batch_dev([(t, v[1:t.numel() + 1], r[1:t.numel() + 1], c[1:len(vt) + 1]) for t, v, r, c in bufs], vt, res)
    rs = device.read_results(res)
    out = []
    for (t, v, r, c), res_k in zip(bufs, rs):
        n = t.numel()
        v, r, c = v.cpu().numpy(), r.cpu().numpy(), c.cpu().numpy()
        assert v[0] == -5 and v[n + 1] == -5 and r[0] == -5 and r[n + 1] == -5 and c[0] == -5 and c[len(vt) + 1] == -5, "wrote outside the outputs"
        out.append((res_k, v[1:n + 1].view(np.uint32), r[1:n + 1].view(np.uint64), c[1:len(vt) + 1].view(np.uint64)))
    return out


@pytest.mark.parametrize("vt", [[0, 1, 2, 7], list(range(20)), [3, 900, 70000, 3]])
def test_batch_of_tag_arrays_each_ranked_on_its_own(env, vt):
    """Ragged tag arrays of one Enum column in many Blocks, incl. empty ones, one that ends on a tile
    boundary and one with a tag no variant carries: every array's variants, ranks and counts equal the
    restated row loop run on that array alone; the bad array is reported and the others unaffected."""
    torch, device, onp, L = env
    rng = np.random.default_rng(len(vt))
    pool = sorted(set(vt))
    sizes = [0, 1, 2048, 4096 + 1, 70001, 0, 333, 2047, 1 << 17, 5, 12288, 100003, 0]
    arrays = [rng.choice(pool, n).astype(np.uint64).astype(np.uint32) for n in sizes]
    bad_at, bad_row = 6, 200
    arrays[bad_at][bad_row] = 123456
    got = _run_batch(torch, device, arrays, vt)
    for k, (tags, (r, gv, gr, gc)) in enumerate(zip(arrays, got)):
        st, variant, rank, counts, first = onp.enum_rows(tags, vt)
        if k == bad_at:
            assert st == 8 and r.status == L.ERR_UNKNOWN_TAG and r.out_count == first == bad_row and r.midpoint == 123456
            continue
        assert st == 0 and r.status == 0 and r.out_count == tags.size, k
        assert np.array_equal(gv, variant) and np.array_equal(gr, rank) and np.array_equal(gc, counts), (k, tags.size)


def test_batch_equals_single_calls_and_empty_batch(env):
    torch, device, onp, L = env
    rng = np.random.default_rng(31)
    vt = [0, 1, 2, 7]
    arrays = [rng.choice(vt, n, p=[.5, .25, .15, .1]).astype(np.uint32) for n in (1 << 15, (1 << 15) + 77, 9, 1 << 16)]
    got = _run_batch(torch, device, arrays, vt)
    for tags, (r, gv, gr, gc) in zip(arrays, got):
        r1, v1, k1, c1 = _run(torch, device, tags, vt)
        assert r.status == r1.status == 0 and np.array_equal(gv, v1) and np.array_equal(gr, k1) and np.array_equal(gc, c1)
    assert L.lib.zbra_hip_enum_rows_batch_dev(None, 0, None, 0, None, None) == 0
    # only empty arrays: counts are written (zeros), nothing else
    got = _run_batch(torch, device, [np.zeros(0, np.uint32)] * 3, vt)
    assert all(r.status == 0 and r.out_count == 0 and gc.tolist() == [0, 0, 0, 0] for r, _, _, gc in got)
