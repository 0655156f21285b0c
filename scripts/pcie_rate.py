#!/usr/bin/env python3
"""PCIe-inclusive decode rate: encoded arrays start in pinned host memory, decoded values end in
pinned host memory.  Blocks of 2^22 rows (the file layout of the full-file configuration) are
pipelined over two streams: H2D(block k+1) | decode(block k) | D2H(block k-1).

Prints one JSON line (a PCIe-inclusive rate: never bench.py's `value`)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--blocks", type=int, default=64)
    ap.add_argument("--rows-log2", type=int, default=22)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--streams", type=int, default=2)
    args = ap.parse_args()
    import torch

    from zbra_amd import device, workloads
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    n = 1 << args.rows_log2
    # untimed: build the encoded blocks on the device, park them in pinned host memory
    enc_host, lens = [], []
    cap = device.encode_bound(n)
    buf = torch.empty(cap, dtype=torch.uint8, device=dev)
    res = device.new_results(1, dev)
    for b in range(args.blocks):
        v = workloads.narrow_column(0x5EED0005, b * n, n, dev)
        device.encode_dev(v, buf, res)
        r = device.read_results(res)[0]
        assert r.status == 0
        ln = int(r.out_count)
        h = torch.empty(ln, dtype=torch.uint8).pin_memory()
        h.copy_(buf[:ln])
        enc_host.append(h)
        lens.append(ln)
    check = workloads.narrow_column(0x5EED0005, (args.blocks - 1) * n, n, dev).cpu()
    S = args.streams
    streams = [torch.cuda.Stream(device=dev) for _ in range(S)]
    d_enc = [torch.empty(cap, dtype=torch.uint8, device=dev) for _ in range(S)]
    d_out = [torch.empty(n, dtype=torch.int64, device=dev) for _ in range(S)]
    h_out = [torch.empty(n, dtype=torch.int64).pin_memory() for _ in range(S)]  # reused ring, not retained
    enc_bytes = sum(lens)

    def run():
        for b in range(args.blocks):
            s = b % S
            with torch.cuda.stream(streams[s]):
                d_enc[s][:lens[b]].copy_(enc_host[b], non_blocking=True)
                device.decode_dev(d_enc[s], n, d_out[s], None, length=lens[b], stream=streams[s])
                h_out[s].copy_(d_out[s], non_blocking=True)
        torch.cuda.synchronize()

    run()
    assert torch.equal(h_out[(args.blocks - 1) % S], check)
    best = None
    for _ in range(args.reps):
        t0 = time.perf_counter()
        run()
        dt = time.perf_counter() - t0
        best = dt if best is None or dt < best else best
    out_bytes = 8.0 * n * args.blocks
    print(json.dumps({"what": "H2D(pinned) + decode + D2H(pinned), %d blocks of 2^%d rows, w=12, %d streams" % (args.blocks, args.rows_log2, S),
                      "seconds": round(best, 5), "decoded_GBps": round(out_bytes / best / 1e9, 2),
                      "h2d_GB": round(enc_bytes / 1e9, 3), "d2h_GB": round(out_bytes / 1e9, 3),
                      "pcie_GBps_both_directions": round((enc_bytes + out_bytes) / best / 1e9, 2)}))


if __name__ == "__main__":
    main()
