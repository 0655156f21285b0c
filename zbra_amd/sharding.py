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
    """(max elapsed over ranks, sum of units over ranks); identity when no process group exists.
    A group of one rank still runs the two all-reduces: that is how a single GPU rehearses the
    collective path of the many-GPU job."""
    if not (dist.is_available() and dist.is_initialized()):
        return elapsed_s, units
    t = torch.tensor([elapsed_s], dtype=torch.float64, device=device)
    u = torch.tensor([units], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(u, op=dist.ReduceOp.SUM)
    return float(t.item()), float(u.item())


def gather_ranks(entry):
    """Every rank's `entry` (any picklable object) in rank order, on every rank; [entry] when no
    process group exists."""
    if not (dist.is_available() and dist.is_initialized()):
        return [entry]
    got = [None] * dist.get_world_size()
    dist.all_gather_object(got, entry)
    return got


def ranks_seen() -> int:
    """Size of the process group as the collective library reports it (1 without a group)."""
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def shared_devices(devices):
    """Device identities that more than one rank reports, sorted: a one-rank-per-GPU job must have none."""
    seen, twice = set(), set()
    for d in devices:
        (twice if d in seen else seen).add(d)
    return sorted(twice)
