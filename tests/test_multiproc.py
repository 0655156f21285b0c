"""world_size-2 gloo test of the multi-process path on CPU: the partition functions bench.py uses,
the max-time / sum-units reduction, and that the union of what the ranks decode (with the oracle
standing in for the GPU decode) is the whole column, in order, with no unit decoded twice."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, tmp):
    import sys
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import zbra_oracle_c as oc
    from zbra_amd import sharding
    rows = 1 << 14
    first = sharding.first_row(rank, rows)
    v = oc.gen_narrow(0x5EED0002, first, rows)          # this rank's column of the weak-scaling job
    e = oc.compress_int_array(v)
    st, back = oc.decompress_int_array(e, rows)
    assert st == 0 and np.array_equal(back, v)
    np.save(os.path.join(tmp, f"rank{rank}.npy"), back)
    # Blocks partitioned two ways
    lo, hi = sharding.shard_range(11, world, rank)
    rr = list(sharding.shard_round_robin(11, world, rank))
    got = [None] * world
    dist.all_gather_object(got, (lo, hi, rr))
    dist.barrier()
    elapsed, units = sharding.job_totals(0.5 + rank, float(rows))
    if rank == 0:
        assert elapsed == 0.5 + (world - 1) and units == rows * world
        covered = []
        for lo_, hi_, _ in got:
            covered += list(range(lo_, hi_))
        assert covered == list(range(11))
        assert sorted(sum((g[2] for g in got), [])) == list(range(11))
    dist.destroy_process_group()


def test_two_ranks_gloo(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    from oracle import zbra_oracle_c as oc
    whole = np.concatenate([np.load(tmp_path / f"rank{r}.npy") for r in range(world)])
    assert np.array_equal(whole, oc.gen_narrow(0x5EED0002, 0, world * (1 << 14)))


def test_shard_range_properties():
    from zbra_amd import sharding
    for n in (0, 1, 7, 8, 9, 1000):
        for world in (1, 2, 3, 8):
            parts = [sharding.shard_range(n, world, r) for r in range(world)]
            assert parts[0][0] == 0 and parts[-1][1] == n
            assert all(parts[i][1] == parts[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in parts]
            assert max(sizes) - min(sizes) <= 1
