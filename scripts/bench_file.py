#!/usr/bin/env python3
"""The full-file configuration on one GPU, scaled down: a .zbra image of `--blocks` Blocks of
2^`--rows-log2` rows x 4 Int columns (entity id: clustered runs, attribute id: <= 8 bits, time:
milliseconds, monotone, value: <= 12 bits), written by this package's batched writer.  Measures
(i) the batched decode alone, image and values resident in HBM, and (ii) image in pageable host
memory -> values in host memory through zbra_hip_file_index + zbra_hip_file_decode.  The scale
(64 GB per file in the full configuration) is stated in the output.

Under torch.distributed.run (one process per GPU) every rank writes and decodes its OWN file
(file f on GPU f; files share nothing), and rank 0 adds a line with the whole-job rates.
ZBRA_BENCH_BACKEND=gloo lets several ranks rehearse that on one GPU."""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--blocks", type=int, default=24)
    ap.add_argument("--rows-log2", type=int, default=22)
    ap.add_argument("--reps", type=int, default=3)
    args = ap.parse_args()
    import torch
    import torch.distributed as dist

    from zbra_amd import _lib, binary as zb, device, sharding
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    backend = os.environ.get("ZBRA_BENCH_BACKEND", "nccl")
    dev = torch.device("cuda", local_rank if backend == "nccl" else local_rank % torch.cuda.device_count())
    torch.cuda.set_device(dev)  # also the device of the host-pointer calls of this thread
    if world > 1:
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend=backend)
    n = 1 << args.rows_log2
    from zbra_amd import workloads
    blocks = workloads.file_blocks(rank * args.blocks, args.blocks, n)  # file `rank` holds these rows of the table
    t0 = time.perf_counter()
    img = zb.BinaryFile("{}", blocks).to_bytes()
    t_write = time.perf_counter() - t0
    info, arrays = zb.file_index(img)
    na = info.n_arrays
    out_bytes = 8 * int(info.total_values)
    # (i) device-resident: image in HBM, one batched submission
    d_img = torch.frombuffer(bytearray(img), dtype=torch.uint8).to(dev)
    outs = [torch.empty(arrays[k].count, dtype=torch.int64, device=dev) for k in range(na)]
    descs = device.make_descs([(d_img.data_ptr() + arrays[k].offset, arrays[k].len, arrays[k].count, outs[k].data_ptr(), _lib.ELEM_I64)
                               for k in range(na)])
    res = device.new_results(na, dev)
    device.decode_batch_dev(descs, res)
    torch.cuda.synchronize()
    rs = device.read_results(res)
    assert all(r.status == 0 for r in rs)
    widths = sorted(set(r.width for r in rs))
    assert np.array_equal(outs[3].cpu().numpy(), blocks[0].table.column.fields[3][1].values)
    for _ in range(30):  # the clocks drop while the host compares results: bring them back before timing
        device.decode_batch_dev(descs, res)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        device.decode_batch_dev(descs, res)
    e1.record()
    torch.cuda.synchronize()
    k_ms = e0.elapsed_time(e1) / 20
    enc_bytes = sum(arrays[k].len for k in range(na))
    del outs, d_img
    # (ii) host image -> host values
    h_outs = [np.empty(arrays[k].count, dtype=np.int64) for k in range(na)]
    ptrs = (C.c_void_p * na)(*[o.ctypes.data for o in h_outs])
    statuses = (C.c_int * na)()
    buf = np.frombuffer(img, dtype=np.uint8)
    best = None
    if world > 1:
        dist.barrier()  # the ranks share the host's memory bandwidth and PCIe root complexes: run together
    for _ in range(args.reps + 1):
        t0 = time.perf_counter()
        st = _lib.lib.zbra_hip_file_decode(buf.ctypes.data_as(C.c_void_p), buf.size, arrays, na, ptrs, statuses)
        dt = time.perf_counter() - t0
        assert st == 0
        best = dt if best is None or dt < best else best
    assert np.array_equal(h_outs[-1], blocks[-1].table.column.fields[3][1].values)
    print(json.dumps({
        "what": f"full-file configuration, one GPU's file scaled to {args.blocks} Blocks of 2^{args.rows_log2} rows x 4 Int columns "
                f"({len(img) / 1e9:.2f} GB image, {out_bytes / 1e9:.2f} GB of values; the full configuration is 64 GB per file)",
        "arrays": int(na), "widths": widths, "write_to_bytes_s": round(t_write, 3),
        "kernel_only_ms": round(k_ms, 4), "kernel_only_decoded_GBps": round(out_bytes / k_ms / 1e6, 1),
        "kernel_only_algorithmic_GBps": round((enc_bytes + out_bytes) / k_ms / 1e6, 1),
        "hbm_frac": round((enc_bytes + out_bytes) / k_ms / 1e6 / 8000, 4),
        "host_to_host_ms": round(best * 1e3, 2), "host_to_host_decoded_GBps": round(out_bytes / best / 1e9, 2)}), flush=True)
    if world > 1:
        red = dev if backend == "nccl" else "cpu"
        k_all, bytes_all = sharding.job_totals(k_ms * 1e-3, float(out_bytes), red)
        h_all, _ = sharding.job_totals(best, float(out_bytes), red)
        if rank == 0:
            print(json.dumps({"what": "whole job: file f on GPU f, no collective on the data path", "n_gpus": world,
                              "kernel_only_decoded_GBps": round(bytes_all / k_all / 1e9, 1),
                              "host_to_host_decoded_GBps": round(bytes_all / h_all / 1e9, 2)}), flush=True)
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
