#!/usr/bin/env python3
"""The full-file configuration at its full size on ONE GPU: one .zbra file of `--target-gb` GB
(64 in the configuration: 8 such files, one per GPU).  Blocks of 2^22 rows x 4 Int columns
(attribute id <= 8 bits, entity id in clustered runs, time in milliseconds, value <= 12 bits):
`--blocks` distinct Blocks are written by this package's batched writer and their bytes repeated
until the image has the target size (the configuration says "repeated to 64 GB/file"); the block
count in the header is patched, so the image is one valid file that zbra_hip_file_index walks.

Measures
  (i)  the whole file decoded in ONE batched submission, image and values resident in HBM
       (64 GB of image + 164 GB of values fit the 288 GB of one MI355X), and
  (ii) image in pageable host memory -> values in host memory, one zbra_hip_file_decode call per
       group of `--blocks` Blocks into a REUSED set of output buffers (the values are consumed and
       not retained: 164 GB of them do not fit beside the image in the box's memory share), and
  (iii) the write path at the same size: the file's 4896 int arrays in host memory (the distinct
       Blocks' vectors, repeated as in the file) -> their frames in host memory through
       zbra_hip_encode_frames, checked against the frames of the image.
If the device or host does not have the memory, the file is scaled down and the scale is stated."""
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
    ap.add_argument("--target-gb", type=float, default=64.0)
    ap.add_argument("--blocks", type=int, default=24, help="distinct Blocks (and Blocks per host-to-host call)")
    ap.add_argument("--rows-log2", type=int, default=22)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--skip-host", action="store_true")
    ap.add_argument("--skip-write", action="store_true")
    ap.add_argument("--group-mult", type=int, nargs="+", default=[1], help="host-to-host calls take this many x --blocks Blocks each")
    args = ap.parse_args()
    import torch

    from zbra_amd import _lib, binary as zb, device
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    n = 1 << args.rows_log2
    from zbra_amd import workloads
    blocks = workloads.file_blocks(0, args.blocks, n)
    cols = [dict(b.table.column.fields)[name].values for b in blocks for name in workloads.FILE_COLUMNS]
    base = np.frombuffer(zb.BinaryFile("{}", blocks).to_bytes(), dtype=np.uint8)
    binfo, barrays = zb.file_index(base)
    per = 4 * args.blocks                               # int arrays per group of Blocks
    assert binfo.n_arrays == per and binfo.end_offset == base.size
    hdr = int(binfo.blocks_offset) + 4                  # header + u32 block_count
    body = base.size - hdr                              # the Blocks' bytes
    values_per_group = int(binfo.total_values)
    want = max(1, int(round((args.target_gb * 1e9 - hdr) / body)))
    # what fits: image + values + workspace in HBM
    free_b, total_b = torch.cuda.mem_get_info()
    fit = int((free_b - (6 << 30)) // (body + 8 * values_per_group))
    reps_of_body = max(1, min(want, fit))
    total = hdr + body * reps_of_body
    print(f"# base image {base.size / 1e9:.3f} GB ({args.blocks} Blocks), body x{reps_of_body} -> {total / 1e9:.2f} GB image, "
          f"{8 * values_per_group * reps_of_body / 1e9:.1f} GB of values; HBM free {free_b / 1e9:.0f} of {total_b / 1e9:.0f} GB",
          flush=True)

    # -- the image in host memory (pageable, as a file read leaves it) --------------------------
    t0 = time.perf_counter()
    img = np.empty(total, dtype=np.uint8)
    img[:hdr] = base[:hdr]
    img[hdr - 4:hdr] = np.frombuffer(np.uint32(args.blocks * reps_of_body).tobytes(), dtype=np.uint8)
    for r in range(reps_of_body):
        img[hdr + r * body: hdr + (r + 1) * body] = base[hdr:]
    t_img = time.perf_counter() - t0
    info, arrays = zb.file_index(img)
    na = int(info.n_arrays)
    assert na == per * reps_of_body and info.block_count == args.blocks * reps_of_body and info.end_offset == total
    out_bytes = 8 * int(info.total_values)
    enc_bytes = sum(arrays[k].len for k in range(na))
    print(f"# image built in {t_img:.1f} s; index: {na} int arrays in {info.block_count} Blocks", flush=True)

    line = {}
    # (ii) and (iii) run first: freeing the 228 GB of part (i) leaves the driver unmapping memory for
    # seconds, and host copies that run meanwhile take twice as long
    # -- (ii) host image -> host values, group by group into reused buffers ----------------------
    if not args.skip_host:
        FileArray = type(arrays[0])
        img_p = img.ctypes.data_as(C.c_void_p)
        for mult in args.group_mult:                    # groups of mult x --blocks Blocks per call
            per_call = per * mult
            h_outs = [np.zeros(n, dtype=np.int64) for _ in range(per_call)]  # touched: a reader reuses its buffers
            ptrs = (C.c_void_p * per_call)(*[o.ctypes.data for o in h_outs])
            statuses = (C.c_int * per_call)()
            best, calls = None, []
            for rep in range(2):
                t0 = time.perf_counter()
                for first in range(0, na, per_call):
                    cnt = min(per_call, na - first)
                    grp = (FileArray * cnt).from_address(C.addressof(arrays) + first * C.sizeof(FileArray))
                    t1 = time.perf_counter()
                    st = _lib.lib.zbra_hip_file_decode(img_p, img.size, grp, cnt, ptrs, statuses)
                    calls.append(time.perf_counter() - t1)
                    assert st == 0, (first, st)
                dt = time.perf_counter() - t0
                best = dt if best is None or dt < best else best
                last = (na - 1) % per_call              # where the file's last array landed
                assert np.array_equal(h_outs[last], cols[(na - 1) % per]) and np.array_equal(h_outs[0], cols[(na - (na - 1) % per_call - 1) % per])
            calls.sort()
            line.setdefault("host_to_host", []).append({
                "blocks_per_call": args.blocks * mult, "values_per_call_GB": round(8 * values_per_group * mult / 1e9, 2),
                "calls": len(calls) // 2, "s": round(best, 3), "decoded_GBps": round(out_bytes / best / 1e9, 2),
                "link_GBps_both_directions": round((total + out_bytes) / best / 1e9, 2),
                "call_ms_min_median_max": [round(calls[0] * 1e3, 1), round(calls[len(calls) // 2] * 1e3, 1), round(calls[-1] * 1e3, 1)]})
            del h_outs
    # -- (iii) the write path at the same size: the file's int arrays (host memory) -> its frames ---
    if not args.skip_host and not args.skip_write:
        frames_of_base = [bytes(base[barrays[k].offset - 8: barrays[k].offset + barrays[k].len]) for k in range(per)]
        del img, img_p                                  # 64 GB back before the frame buffer is allocated
        img = None
        items = [(cols[k % per].ctypes.data, n, 0, 0, _lib.ELEM_I64) for k in range(na)]  # the 24 Blocks' vectors, repeated as in the file
        edescs = device.make_encode_descs(items)
        room = sum(len(f) for f in frames_of_base) * reps_of_body
        out = np.empty(room + 4096, dtype=np.uint8)
        offs = np.zeros(na + 1, dtype=np.uint64)
        statuses = (C.c_int * na)()
        t0 = time.perf_counter()
        st = _lib.lib.zbra_hip_encode_frames(edescs, na, out.ctypes.data_as(C.c_void_p), out.size, offs.ctypes.data_as(C.POINTER(C.c_uint64)), statuses)
        t_enc = time.perf_counter() - t0
        assert st == 0 and int(offs[na]) == room, (st, int(offs[na]), room)
        group = int(offs[per])
        first = out[:group].tobytes()
        assert first == b"".join(frames_of_base)
        assert all(out[r * group:(r + 1) * group].tobytes() == first for r in range(1, reps_of_body, max(1, reps_of_body // 8)))
        line["write_path"] = {"arrays": na, "values_GB": round(out_bytes / 1e9, 1), "frames_GB": round(room / 1e9, 2), "s": round(t_enc, 3),
                              "values_GBps": round(out_bytes / t_enc / 1e9, 2),
                              "note": "zbra_hip_encode_frames, host values -> host frames, first call of the process (buffers fresh)"}
        del out, first
    # -- (i) device-resident: the whole file in one batched submission ---------------------------
    d_img = torch.empty(total, dtype=torch.uint8, device=dev)
    d_img[:base.size].copy_(torch.from_numpy(base))
    d_img[hdr - 4:hdr].copy_(torch.from_numpy(np.frombuffer(np.uint32(args.blocks * reps_of_body).tobytes(), dtype=np.uint8).copy()))
    for r in range(1, reps_of_body):                    # the same bytes the host image holds
        d_img[hdr + r * body: hdr + (r + 1) * body].copy_(d_img[hdr:hdr + body])
    d_out = torch.empty(int(info.total_values), dtype=torch.int64, device=dev)
    pos, items = 0, []
    for k in range(na):
        items.append((d_img.data_ptr() + arrays[k].offset, arrays[k].len, arrays[k].count, d_out.data_ptr() + 8 * pos, _lib.ELEM_I64))
        pos += arrays[k].count
    descs = device.make_descs(items)
    res = device.new_results(na, dev)
    device.decode_batch_dev(descs, res)
    torch.cuda.synchronize()
    rs = device.read_results(res)
    assert all(r.status == 0 and r.out_count == n for r in rs)
    widths = sorted(set(r.width for r in rs))
    # every copy of the Blocks decodes to the values they were written from
    first = d_out[:values_per_group]
    for k in range(per):
        assert torch.equal(first[k * n:(k + 1) * n], torch.from_numpy(cols[k]).to(dev)), k
    for r in range(1, reps_of_body):
        assert torch.equal(d_out[r * values_per_group:(r + 1) * values_per_group], first), r
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    device.decode_batch_dev(descs, res)
    e0.record()
    for _ in range(args.reps):
        device.decode_batch_dev(descs, res)
    e1.record()
    torch.cuda.synchronize()
    k_ms = e0.elapsed_time(e1) / args.reps
    line.update({
        "what": f"full-file configuration, ONE file of {total / 1e9:.2f} GB ({info.block_count} Blocks of 2^{args.rows_log2} rows x 4 Int "
                f"columns, {args.blocks} distinct Blocks repeated x{reps_of_body}; the configuration's file is 64 GB), "
                f"{out_bytes / 1e9:.1f} GB of values",
        "arrays": na, "widths": widths,
        "device_resident_one_submission_ms": round(k_ms, 3),
        "device_resident_decoded_GBps": round(out_bytes / k_ms / 1e6, 1),
        "device_resident_algorithmic_GBps": round((enc_bytes + out_bytes) / k_ms / 1e6, 1),
        "hbm_frac": round((enc_bytes + out_bytes) / k_ms / 1e6 / 8000, 4),
    })
    del d_out, d_img, first
    torch.cuda.empty_cache()

    print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
