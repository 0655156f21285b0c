#!/usr/bin/env python3
"""Decode rates of the other device-resident configurations (parity cases, not bench lines):
3 = mixed per-array widths, 1e9 rows in 2^20-row arrays; 4 = the sum-typed table, 2^29 rows per
GPU as Blocks of 2^22 (enum tags + two Int columns per variant).  Everything is generated,
encoded (batched) and decoded (batched) on the device; the check is the exact round trip.  Prints
one JSON line per configuration.

Under torch.distributed.run (one process per GPU) config 4 deals Block b of the 128 x N Blocks to
rank b mod N (Blocks never exchange data, binary.rs:124-127, 167-171), and rank 0 adds a line
with the whole-job rate: every rank's decoded bytes over the slowest rank's time.
ZBRA_BENCH_BACKEND=gloo lets several ranks rehearse that on one GPU."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def timed(fn, reps):
    import torch
    for _ in range(max(3, reps)):  # the clocks drop while the host checks results: bring them back before timing
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


ENCODE_ONLY = False


def run_batch(cols, label, reps):
    """cols: list of int64 / int32 tensors.  Batched encode, batched decode, round trip."""
    import torch

    from zbra_amd import _lib, device
    dev = cols[0].device
    kinds = [_lib.ELEM_U32 if c.dtype == torch.int32 else _lib.ELEM_I64 for c in cols]
    caps = [device.encode_bound(c.numel()) for c in cols]
    offs, total = [], 0
    for cap in caps:
        offs.append(total)
        total += (cap + 63) // 64 * 64 + 3  # odd offsets, as in a file image
    enc = torch.empty(total + 64, dtype=torch.uint8, device=dev)
    res = device.new_results(len(cols), dev)
    edesc = device.make_encode_descs([(c.data_ptr(), c.numel(), enc.data_ptr() + o, cap, k)
                                      for c, o, cap, k in zip(cols, offs, caps, kinds)])
    enc_ms = timed(lambda: device.encode_batch_dev(edesc, res), max(1, reps // 4))
    rs = device.read_results(res)
    assert all(r.status == 0 for r in rs)
    if ENCODE_ONLY:  # for kernel traces: the last dispatches of the process are one encode
        return {"case": label, "arrays": len(cols), "encode_ms": round(enc_ms, 4)}
    lens = [int(r.out_count) for r in rs]
    # a round trip succeeds with ANY midpoint: check the medians themselves against a sort
    for i, (c, r) in enumerate(zip(cols, rs)):
        if c.numel() == 0:
            continue
        s, _ = torch.sort(c.to(torch.int64) & 0xFFFFFFFF if c.dtype == torch.int32 else c)
        m = c.numel()
        a, b = int(s[(m - 1) // 2]), int(s[m // 2])
        d = b - a
        if d >= 1 << 63:
            d -= 1 << 64                                   # b.wrapping_sub(a)
        want = a + (d // 2 if d >= 0 else -((-d) // 2))    # truncating division
        want = (want + (1 << 63)) % (1 << 64) - (1 << 63)  # a.wrapping_add(..)
        assert r.midpoint == want, (label, i, r.midpoint, want)
    outs = [torch.empty_like(c) for c in cols]
    ddesc = device.make_descs([(enc.data_ptr() + o, ln, c.numel(), out.data_ptr(), k)
                               for c, o, ln, out, k in zip(cols, offs, lens, outs, kinds)])
    dres = device.new_results(len(cols), dev)
    dec_ms = timed(lambda: device.decode_batch_dev(ddesc, dres), reps)
    assert all(r.status == 0 and r.out_count == c.numel() for r, c in zip(device.read_results(dres), cols))
    for c, o in zip(cols, outs):
        assert torch.equal(c, o), "round trip"
    rows = sum(c.numel() for c in cols)
    out_bytes = sum(c.numel() * c.element_size() for c in cols)
    enc_bytes = sum(lens)
    widths = sorted(set(r.width for r in rs))
    return {"case": label, "arrays": len(cols), "rows": rows, "encoded_bytes": enc_bytes, "decoded_bytes": out_bytes,
            "widths": widths, "decode_ms": round(dec_ms, 4), "decoded_GBps": round(out_bytes / dec_ms / 1e6, 1),
            "algorithmic_GBps": round((enc_bytes + out_bytes) / dec_ms / 1e6, 1),
            "hbm_frac": round((enc_bytes + out_bytes) / dec_ms / 1e6 / 8000.0, 4),
            "encode_ms": round(enc_ms, 4), "encode_GBps_in": round(out_bytes / enc_ms / 1e6, 1)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, nargs="+", default=[3, 4])
    ap.add_argument("--scale-log2", type=int, default=0, help="divide the row counts by 2^this (smoke runs)")
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--encode-only", action="store_true", help="no median check, no decode (kernel traces)")
    ap.add_argument("--tile-order", type=int, default=2, help="force the decode's tile order: 0 plain, 1 one eighth per XCD, N groups of N (tuning; 2 = the library's choice)")
    ap.add_argument("--only", choices=["a", "b", "tags"], help="config 4: only the clustered (a) or the time-series (b) Int columns, or only the tag columns")
    args = ap.parse_args()
    global ENCODE_ONLY
    ENCODE_ONLY = args.encode_only
    import torch
    import torch.distributed as dist

    from zbra_amd import _lib, sharding, workloads
    if args.tile_order != 2:
        assert _lib.lib.zbra_tuning_set_decode_variant(2048, 512, 3, 0, args.tile_order) == 0
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    backend = os.environ.get("ZBRA_BENCH_BACKEND", "nccl")
    dev = torch.device("cuda", local_rank if backend == "nccl" else local_rank % torch.cuda.device_count())
    torch.cuda.set_device(dev)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend=backend)
    if 3 in args.config and world == 1:
        n = 1_000_000_000 >> args.scale_log2
        cols = workloads.mixed_width_column(n, 1 << 20, dev)
        print(json.dumps(run_batch(cols, "config 3: mixed widths, %d rows in 2^20-row arrays" % n, args.reps)), flush=True)
        del cols
        torch.cuda.empty_cache()
    if 4 in args.config:
        blocks = max(1, 128 >> args.scale_log2)  # per GPU
        tags, ints = [], []
        for b in sharding.shard_round_robin(blocks * world, world, rank):
            t, variants = workloads.enum_block(b, 1 << 22, dev)
            tags.append(t)
            for _, a, bcol in variants:
                ints += [a, bcol] if not args.only else [a] if args.only == "a" else [bcol]
        print(json.dumps(run_batch(tags, "config 4 (one GPU's 2^29 rows): enum tags -> u32, %d Blocks" % blocks, args.reps)), flush=True)
        if args.only == "tags":
            return
        print(json.dumps(run_batch(ints, "config 4: Int columns of the variants, %d arrays" % len(ints), args.reps)), flush=True)
        if args.only:
            return
        if world > 1:
            dist.barrier()
        both = run_batch(tags + ints, "config 4: tags + Int columns in one submission", args.reps)
        print(json.dumps(both), flush=True)
        if world > 1:
            ms, out_bytes = sharding.job_totals(both["decode_ms"], float(both["decoded_bytes"]), dev if backend == "nccl" else "cpu")
            if rank == 0:
                print(json.dumps({"case": "config 4, whole job: Block b on GPU b mod %d, no collective on the data path" % world,
                                  "n_gpus": world, "blocks": blocks * world, "decode_ms_slowest_rank": round(ms, 4),
                                  "decoded_GBps": round(out_bytes / ms / 1e6, 1)}), flush=True)
            dist.destroy_process_group()


if __name__ == "__main__":
    main()
