#!/usr/bin/env python3
"""Benchmark of the hot path: device-resident int64 column decode (FOR + zig-zag + bit-pack).

Workload (N=1): one synthetic 2^28-row int64 column per GPU, narrow value range
(v = 1.7e12 + sm(seed, i) mod 4000, bit width 12 after the frame of reference), encoded on the
device by this library, then decoded K times.  A step = one decode of the whole column:
402,653,197 encoded bytes read, 2 GiB of int64 written.  With --gpus N every rank holds its own
column (rows [rank*2^28, (rank+1)*2^28) of the same generator): weak scaling, no collectives on
the data path.

`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment starts the N ranks
itself (one child process per GPU, before anything in this process touches the GPU) and relays
rank 0's line; under `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` the
ranks already exist and each process is one of them.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak
SEED = 0x5EED0002


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--rows-log2", type=int, default=28, help="rows per GPU = 2^this (default 28: the named config)")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="CPU-baseline budget; 0 disables it")
    ap.add_argument("--no-encode-bench", action="store_true")
    return ap.parse_args()


def spawn_ranks(n: int, argv, child=None, env=None, timeout_s: float = 3000.0) -> int:
    """Starts n child processes, one per rank (RANK = LOCAL_RANK = 0..n-1, WORLD_SIZE = n, rendezvous
    on 127.0.0.1), lets rank 0 write to this process's stdout, waits for all of them and returns 0
    only if every one exited with 0.  When one fails the others are terminated (by PID) and its exit
    code is returned.  Must be called before this process initialises the GPU: the children are new
    processes, nothing is exec'ed over this one."""
    child = list(child) if child is not None else [sys.executable, os.path.abspath(__file__)]
    base = dict(os.environ if env is None else env)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    base.update(WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    procs = []
    for r in range(n):
        e = dict(base, RANK=str(r), LOCAL_RANK=str(r))
        # ranks other than 0 print nothing to stdout (rank 0 owns the ONE JSON line); stderr is shared
        procs.append(subprocess.Popen(child + list(argv), env=e, stdout=None if r == 0 else subprocess.DEVNULL))
    deadline = time.monotonic() + timeout_s
    rc, left = 0, set(range(n))
    while left and rc == 0:
        for r in sorted(left):
            code = procs[r].poll()
            if code is None:
                continue
            left.discard(r)
            if code != 0:
                rc = code if code > 0 else 128 - code
                print(f"bench.py: rank {r} exited with {code}", file=sys.stderr, flush=True)
                break
        if time.monotonic() > deadline:
            rc = 124
            print("bench.py: ranks timed out", file=sys.stderr, flush=True)
        if left and rc == 0:
            time.sleep(0.05)
    for r in left:  # only after a failure: end the ranks that would wait for the failed one forever
        procs[r].terminate()
    for r in left:
        try:
            procs[r].wait(timeout=10)
        except subprocess.TimeoutExpired:
            procs[r].kill()
            procs[r].wait()
    return rc


def pmc_traffic(rows_log2: int):
    """HBM bytes per decode launch from the PMC counters, and where the figure comes from.
    rocprofv3 cannot run inside the timed process, so it is read from the newest
    profiles/*_pmc_traffic.json: separate --pmc FETCH_SIZE / --pmc WRITE_SIZE passes over this
    same command (with the gfx950 FETCH_SIZE x2 correction applied there), not this run."""
    if rows_log2 != 28:
        return None, None
    try:
        names = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_pmc_traffic.json"))
        path = os.path.join(ROOT, "profiles", names[-1])
        with open(path) as f:
            doc = json.load(f)
        src = {"file": "profiles/" + names[-1], "collected": doc.get("collected"),
               "note": "separate rocprofv3 --pmc passes over this command, not this run"}
        return int(doc["hbm_bytes_per_launch"]["total"]), src
    except Exception:
        return None, None


def cpu_baseline(n_sample: int, budget_s: float):
    """The C restatement of the reference codec (oracle/, kind 'port'), one core, decoding the
    first n_sample rows of the same column.  Test infrastructure used as a reported baseline."""
    import numpy as np

    from oracle import zbra_oracle_c as oc
    v = oc.gen_narrow(SEED, 0, n_sample)
    enc = np.frombuffer(oc.compress_int_array(v), dtype=np.uint8).copy()
    out = np.empty(n_sample, dtype=np.int64)
    oc.decompress_int_array_into(enc, n_sample, out)  # warm
    assert np.array_equal(out, v)
    reps, t0 = 0, time.perf_counter()
    while True:
        oc.decompress_int_array_into(enc, n_sample, out)
        reps += 1
        el = time.perf_counter() - t0
        if el >= budget_s or reps >= 200:
            break
    dec_gbps = 8.0 * n_sample * reps / el / 1e9
    # encode leg (the median's sort dominates): one timed call each on a smaller slice, with the
    # port's qsort and with an inlined-compare stable merge sort, which is what Rust's Vec::sort
    # (compression.rs:58-59) is: the second is the fair stand-in for the reference's encode
    m = min(n_sample, 1 << 22)
    buf = np.empty(13 + 8 * m, dtype=np.uint8)
    t1 = time.perf_counter()
    oc.compress_int_array_into(v[:m], buf)
    enc_gbps = 8.0 * m / (time.perf_counter() - t1) / 1e9
    buf2 = np.empty(13 + 8 * m, dtype=np.uint8)
    t1 = time.perf_counter()
    got = oc.compress_int_array_stable_sort_into(v[:m], buf2)
    enc_fair_gbps = 8.0 * m / (time.perf_counter() - t1) / 1e9
    assert np.array_equal(buf[:got], buf2[:got])
    # all host cores the box gives this job, one independent chunk per thread (ctypes releases the GIL)
    from concurrent.futures import ThreadPoolExecutor
    threads = max(1, min(16, len(os.sched_getaffinity(0))))
    chunk = n_sample // threads
    parts = [(enc_chunk, np.empty(chunk, dtype=np.int64)) for enc_chunk in
             (np.frombuffer(oc.compress_int_array(v[i * chunk:(i + 1) * chunk]), dtype=np.uint8).copy() for i in range(threads))]
    with ThreadPoolExecutor(threads) as pool:
        t2, rounds = time.perf_counter(), 0
        while True:
            list(pool.map(lambda pr: oc.decompress_int_array_into(pr[0], chunk, pr[1]), parts))
            rounds += 1
            if time.perf_counter() - t2 >= min(budget_s, 5.0) or rounds >= 200:
                break
        all_gbps = 8.0 * chunk * threads * rounds / (time.perf_counter() - t2) / 1e9
    return {"value": round(dec_gbps, 4), "unit": "GB/s", "cores": 1, "kind": "port",
            "sample": f"first 2^{n_sample.bit_length() - 1} rows of the same column, {reps} reps, "
                      f"C restatement of zbra-core compression.rs (gcc -O3), decode only",
            "encode_GBps": round(enc_fair_gbps, 4),
            "encode_sample": f"first 2^{m.bit_length() - 1} rows, one call, median by std::stable_sort (inlined compare, as Rust's "
                             f"Vec::sort); the statement-by-statement port with C qsort: {enc_gbps:.4f} GB/s",
            "all_cores": {"value": round(all_gbps, 3), "unit": "GB/s", "cores": threads,
                          "sample": "same rows cut into one chunk per thread"}}


def claim_stdout() -> int:
    """stdout carries ONE JSON line.  Libraries print there too (RCCL its version banner when the
    first communicator is made, gloo its peer count), so from here on file descriptor 1 of this
    process is stderr, and the line goes to the returned duplicate of the real stdout."""
    sys.stdout.flush()
    real = os.dup(1)
    os.dup2(2, 1)
    return real


def emit_line(real_stdout: int, line: dict):
    os.write(real_stdout, (json.dumps(line) + "\n").encode())


def force_dist() -> bool:
    """ZBRA_BENCH_FORCE_DIST=1: a single rank also creates its process group and goes through every
    collective of the many-rank job (barriers, the two all-reduces, the gather of the ranks'
    reports): the one-GPU rehearsal of the code an 8-GPU run executes."""
    return os.environ.get("ZBRA_BENCH_FORCE_DIST") == "1"


def rendezvous_env():
    """A process group of one rank that no launcher prepared still needs a rendezvous address."""
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if "MASTER_PORT" not in os.environ:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            os.environ["MASTER_PORT"] = str(s.getsockname()[1])
    os.environ.setdefault("RANK", "0")
    os.environ.setdefault("WORLD_SIZE", "1")


def device_identity(torch, gpu: int) -> str:
    """PCI address of the device this rank decodes on (the same GPU has the same address in every
    process of the node, whatever index it has there) and its UUID, as far as torch tells them."""
    p = torch.cuda.get_device_properties(gpu)
    uuid = str(getattr(p, "uuid", "") or "")
    try:
        pci = "%04x:%02x:%02x" % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
    except AttributeError:
        return uuid or f"index:{gpu}"
    # partitions of one card share its PCI address and differ in their UUID: both, where torch tells both
    return f"{pci}/{uuid}" if uuid else pci


def job_report(elapsed_s: float, units: float, step_ms: float, device_id: str, red_dev, one_gpu_per_rank: bool):
    """The whole job from every rank's numbers: (slowest rank's time, all ranks' units, the fields
    the JSON line carries about the ranks).  Every rank gets the same answer.  Two ranks of a
    one-rank-per-GPU job on ONE device would report a rate no 8-GPU node has: that ends the job with
    exit code 3 on every rank."""
    from zbra_amd import sharding
    total_s, total_units = sharding.job_totals(elapsed_s, units, red_dev)
    per = sharding.gather_ranks({"rank": int(os.environ.get("RANK", "0")), "ms": round(step_ms, 4), "device": device_id})
    fields = {"ranks_seen": sharding.ranks_seen(), "per_rank_ms": [p["ms"] for p in per],
              "per_rank_device": [p["device"] for p in per]}
    assert [p["rank"] for p in per] == list(range(len(per))), per
    clash = sharding.shared_devices(fields["per_rank_device"]) if one_gpu_per_rank else []
    if clash:
        print(f"bench.py: ranks share a device: {clash} in {fields['per_rank_device']}", file=sys.stderr, flush=True)
        raise SystemExit(3)
    return total_s, total_units, fields


def dry_rank():
    """Launcher and reduction rehearsal on a machine without a GPU (ZBRA_BENCH_DRY_RANK=1): the rank
    joins a gloo group, goes through the barriers and the whole-job reduction with made-up times and
    a made-up device name (ZBRA_BENCH_DRY_CLASH=1: the same name on every rank), and rank 0 prints
    where it sits in the job."""
    real_stdout = claim_stdout()
    import torch.distributed as dist
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    grouped = world > 1 or force_dist()
    if grouped:
        rendezvous_env()
        dist.init_process_group(backend="gloo")
        dist.barrier()
    name = "dry:0" if os.environ.get("ZBRA_BENCH_DRY_CLASH") == "1" else f"dry:{local_rank}"
    try:
        elapsed, units, fields = job_report(0.001 * (rank + 1), 100.0, 1.0 + rank, name, "cpu", True)
        if rank == 0:
            line = {"dry": True, "rank": rank, "n_gpus": world, "job_seconds": elapsed, "job_units": units,
                    "config": {"partitioning": f"columns x{world}, no collectives"}}
            line.update(fields)
            emit_line(real_stdout, line)
    finally:
        if grouped:
            dist.destroy_process_group()


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # the parent only launches: torch is not imported and HIP is not touched in this process
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))
    if os.environ.get("ZBRA_BENCH_DRY_RANK") == "1":
        dry_rank()
        return
    real_stdout = claim_stdout()
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: zbra_amd has no CPU path")
    # one rank per GPU; ZBRA_BENCH_BACKEND=gloo lets several ranks rehearse on one GPU (the
    # collective then runs on CPU tensors), the default nccl (= RCCL) needs a GPU per rank
    backend = os.environ.get("ZBRA_BENCH_BACKEND", "nccl")
    if backend == "nccl" and local_rank >= torch.cuda.device_count():
        raise SystemExit(f"bench.py: rank {rank} has no GPU of its own (LOCAL_RANK {local_rank}, {torch.cuda.device_count()} visible)")
    gpu = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(gpu)
    dev = torch.device("cuda", gpu)
    grouped = world > 1 or force_dist()
    if grouped:
        rendezvous_env()
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend=backend)
    red_dev = dev if backend == "nccl" else "cpu"

    from zbra_amd import device, sharding, workloads

    n = 1 << args.rows_log2
    # -- setup (untimed): generate on device, encode on device ------------------------------
    values = workloads.narrow_column(SEED, sharding.first_row(rank, n), n, dev)
    cap = device.encode_bound(n)
    enc_buf = torch.empty(cap, dtype=torch.uint8, device=dev)
    res = device.new_results(1, dev)
    device.encode_dev(values, enc_buf, res)
    r = device.read_results(res)[0]
    assert r.status == 0 and r.width == 12, (r.status, r.width)
    # decode(encode(v)) == v holds for ANY midpoint, so the median is checked on its own (untimed)
    srt = torch.sort(values).values
    lo_med, hi_med = int(srt[n // 2 - 1]), int(srt[n // 2])
    del srt
    assert r.midpoint == lo_med + (hi_med - lo_med) // 2, (r.midpoint, lo_med, hi_med)  # both positive: trunc = floor
    enc_len = int(r.out_count)
    assert enc_len == 13 + (n * 12 + 7) // 8
    encoded = enc_buf[:enc_len].clone()  # exact-size copy: E at its own allocation
    del enc_buf
    out = torch.empty(n, dtype=torch.int64, device=dev)
    dres = device.new_results(1, dev)

    def step():
        device.decode_dev(encoded, n, out, dres)

    # correctness first (untimed), so that nothing but the warm-up steps precedes the timed region:
    # a comparison kernel and a copy back in between let the clocks drop again
    step()
    torch.cuda.synchronize()
    assert torch.equal(out, values), "decode(encode(v)) != v"
    assert device.read_results(dres)[0].out_count == n
    # Part of the set-up: the device has just generated, sorted and compared gigabytes and then sat
    # idle through a copy back; a few dozen decodes bring it to the steady clocks a long-running
    # reader sees.  The W warm-up steps and the K timed steps follow as asked (reported as
    # config.setup_decodes; ZBRA_BENCH_PREWARM=0 leaves them out).
    prewarm = int(os.environ.get("ZBRA_BENCH_PREWARM", "64"))
    for _ in range(prewarm):
        step()
    for _ in range(args.warmup):
        step()

    # -- timed region ------------------------------------------------------------------------
    ev0 = torch.cuda.Event(enable_timing=True)  # events on the stream the kernels run on
    ev1 = torch.cuda.Event(enable_timing=True)
    if grouped:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(args.steps):
        step()
    ev1.record()
    torch.cuda.synchronize()
    own_elapsed = time.perf_counter() - t0  # this rank's K steps, before it waits for the others
    if grouped:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    kernel_ms = ev0.elapsed_time(ev1) / args.steps  # per decode launch (plan + tile kernel)

    # whole job: slowest rank's time, every rank's decoded bytes, and what each rank ran on
    elapsed, decoded_bytes, rank_fields = job_report(elapsed, 8.0 * n * args.steps, own_elapsed / args.steps * 1e3,
                                                     device_identity(torch, gpu), red_dev, backend == "nccl")

    # -- encode timing (secondary; same config names encode + decode) -------------------------
    encode = None
    if not args.no_encode_bench:
        enc_buf = torch.empty(cap, dtype=torch.uint8, device=dev)
        for _ in range(5):
            device.encode_dev(values, enc_buf, res)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 30
        e0.record()
        for _ in range(reps):
            device.encode_dev(values, enc_buf, res)
        e1.record()
        torch.cuda.synchronize()
        enc_ms = e0.elapsed_time(e1) / reps
        assert torch.equal(enc_buf[:enc_len], encoded)
        encode = {"ms": round(enc_ms, 4), "GBps_in": round(8.0 * n / enc_ms / 1e6, 2),
                  "algorithmic_GBps": round((16.0 * n + enc_len) / enc_ms / 1e6, 2)}
        del enc_buf

    if rank == 0:
        traffic, traffic_src = pmc_traffic(args.rows_log2)
        value = decoded_bytes / elapsed / 1e9
        algo_bytes = enc_len + 8.0 * n  # per launch: encoded bytes read + int64 written
        achieved = algo_bytes / (kernel_ms * 1e-3) / 1e9
        line = {
            "metric": "device-resident int64 column decode GB/s (FOR+zigzag+bitpack)",
            "value": round(value, 2),
            "unit": "GB/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "warmup_total": prewarm + args.warmup,  # untimed decodes before the timed region: set-up decodes + --warmup
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "int64",
            "data": "synthetic",
            "config": {"workload": f"single 2^{args.rows_log2}-row int64 column per GPU, v=1.7e12+sm(0x5EED0002,i) mod 4000, "
                                   "bit width 12, decode of device-resident bytes",
                       "rows_per_gpu": n, "bit_width": 12, "encoded_bytes": enc_len, "partitioning": f"columns x{world}, no collectives",
                       "setup_decodes": prewarm},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": "decode_tiles_kernel", "kernel_ms": round(kernel_ms, 4),
                         "algorithmic_bytes_per_launch": int(algo_bytes)},
        }
        line.update(rank_fields)
        line["collectives"] = (f"{dist.get_backend()} process group: barrier around the timed region, max/sum all-reduce and a gather "
                               "of the ranks' reports; none on the data path") if grouped else "none (single process, no group)"
        if encode is not None:
            line["encode"] = encode
        if world == 1 and args.cpu_seconds > 0:
            line["cpu_baseline"] = cpu_baseline(min(n, 1 << 25), args.cpu_seconds)
        emit_line(real_stdout, line)
    if grouped:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
