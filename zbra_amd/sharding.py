"""Partitioning of independent units (columns, Blocks, files) over ranks, and the whole-job
timing reduction.  Arrays, Blocks (binary.rs:124-127, 167-171) and files never exchange data, so
the only communication of a multi-GPU job is the barrier around the timed region and the
max-over-ranks / sum-over-ranks of its scalars."""
import torch
import torch.distributed as dist


def shard_range(n_units: int, world: int, rank: int):
    """Contiguous, balanced partition of range(n_units): rank r gets [lo, hi).  The first
    n_units % world ranks get one extra unit."""
    q, r = divmod(n_units, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def shard_round_robin(n_units: int, world: int, rank: int):
    """Unit b -> rank b mod world (Blocks of a file dealt like cards)."""
    return range(rank, n_units, world)


def first_row(rank: int, rows_per_rank: int) -> int:
    """Weak scaling: rank r owns rows [r * rows_per_rank, (r+1) * rows_per_rank) of the generator."""
    return rank * rows_per_rank


def job_totals(elapsed_s: float, units: float, device="cpu"):
    """(max elapsed over ranks, sum of units over ranks); identity when not distributed."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return elapsed_s, units
    t = torch.tensor([elapsed_s], dtype=torch.float64, device=device)
    u = torch.tensor([units], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(u, op=dist.ReduceOp.SUM)
    return float(t.item()), float(u.item())
