#!/usr/bin/env python3
"""Randomised batches through the device-resident encode and decode, at sizes beyond the CPU
oracle's reach.  A round trip alone proves little (it succeeds with any midpoint), so every array's
midpoint is checked against a device sort and its width against the zig-zag maximum, then the
decode of the batch against the input.

Usage: stress_encode.py [--trials N] [--seed S] [--max-log2 L] [--max-arrays A]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

I64_MIN, I64_MAX = -(1 << 63), (1 << 63) - 1


def make_array(torch, g, dev, kind, n):
    ar = torch.arange(n, dtype=torch.int64, device=dev)
    rnd = lambda lo, hi: torch.randint(lo, hi, (n,), dtype=torch.int64, device=dev, generator=g)  # noqa: E731
    if kind == "constant":
        return torch.full((n,), 987654321, dtype=torch.int64, device=dev)
    if kind == "two":
        return rnd(0, 2)
    if kind == "two_far":                      # span >= 2^63: wrap-around in the deltas
        return torch.where(rnd(0, 2) == 0, I64_MIN + 5, I64_MAX - 7)
    if kind == "narrow":
        return 1_700_000_000_000 + rnd(0, 4000)
    if kind == "wide":
        return rnd(-(1 << 24), 1 << 24)
    if kind == "w40":
        return rnd(-(1 << 38), 1 << 38) - (1 << 50)
    if kind == "full":
        return rnd(-(1 << 62), 1 << 62) * 2 + rnd(0, 2)
    if kind == "sorted":
        return ar * 3 - n
    if kind == "reversed":
        return (n - ar) * 1000
    if kind == "sawtooth":                     # period = the sampling stride's neighbourhood
        return (ar % 4096) * 17
    if kind == "median_run":                   # half the values equal: the median sits in a long run
        return torch.where(rnd(0, 2) == 0, 42, rnd(-100000, 100000))
    if kind == "outliers":                     # narrow bulk + a few far values on both sides
        v = rnd(0, 1000)
        v[:: max(1, n // 7)] = I64_MAX - 3
        v[1:: max(1, n // 5)] = I64_MIN + 11
        return v
    if kind == "skewed":                       # sample-based brackets see a lopsided distribution
        return (rnd(0, 1 << 20) * rnd(0, 1 << 20)) >> 10
    if kind == "full64":                       # every bit random: the widest brackets there are
        return (rnd(-(1 << 62), 1 << 62) << 1) ^ rnd(0, 4)
    if kind == "three_far":                    # three values far apart: the candidate list overflows
        return torch.where(rnd(0, 3) == 0, -(1 << 62), torch.where(rnd(0, 2) == 0, 12345, (1 << 62) + 99))
    if kind == "popular_in_wide":              # a few per cent of one value at the median of wide data
        v = torch.where(rnd(0, 2) == 0, -rnd(1 << 40, 1 << 62), rnd(1 << 40, 1 << 62))
        return torch.where(rnd(0, 100) < 3, 12345, v)
    if kind == "cluster_in_wide":              # a dense cluster of different values at the median of wide data
        v = torch.where(rnd(0, 2) == 0, -rnd(1 << 40, 1 << 62), rnd(1 << 40, 1 << 62))
        return torch.where(rnd(0, 100) < 4, 500_000_000 + rnd(0, 5000), v)
    if kind == "heavy_ends":                   # a dense bracket of three keys: 45/3/3/49 %
        u = rnd(0, 100)
        return torch.where(u < 45, 0, torch.where(u < 48, 1, torch.where(u < 51, 2, 3)))
    if kind == "few_tags":                     # a bracket of up to eight keys with a rare key at the median
        u = rnd(0, 100)
        return torch.where(u < 40, 7, torch.where(u < 47, 8 + u % 4, torch.where(u < 53, 12 + u % 3, 15)))
    if kind == "doubles":                      # f64 bit patterns of a normal-ish spread around zero
        x = (torch.rand(n, device=dev, generator=g, dtype=torch.float64) - 0.5) * 1e6
        return x.view(torch.int64)
    raise ValueError(kind)


KINDS = ["constant", "two", "two_far", "narrow", "wide", "w40", "full", "sorted", "reversed", "sawtooth", "median_run",
         "outliers", "skewed", "full64", "three_far", "popular_in_wide", "cluster_in_wide", "heavy_ends", "few_tags", "doubles"]


def expected(torch, v):
    """(midpoint, width) by the reference's rules (compression.rs:52-74, 117-131), computed with a sort."""
    m = v.numel()
    s = torch.sort(v).values
    a, b = int(s[(m - 1) // 2]), int(s[m // 2])
    d = b - a
    if d >= 1 << 63:
        d -= 1 << 64                                        # b.wrapping_sub(a)
    mid = a + (d // 2 if d >= 0 else -((-d) // 2))          # truncating division
    mid = (mid + (1 << 63)) % (1 << 64) - (1 << 63)         # a.wrapping_add(..)
    delta = v - mid                                         # int64 arithmetic wraps
    zz = (delta << 1) ^ (delta >> 63)
    if bool((zz < 0).any()):
        w = 64
    else:
        w = max(1, int(zz.max()).bit_length())
    return mid, w


def run_trial(torch, device, L, seed, max_log2, max_arrays, verbose=False):
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    pick = lambda hi: int(torch.randint(0, hi, (1,), generator=g, device=dev))  # noqa: E731
    # shape of the trial: one array alone (the single-array launch sequence), arrays of one length
    # (u32 arrays of one length are packed by a kernel that numbers its tiles per array), or ragged lengths
    shape = ("single", "equal", "equal", "ragged", "ragged", "ragged")[pick(6)]
    # a quarter of the trials hold u32 arrays only (enum tags, or any u32 column): such submissions
    # have a bracket pass and, when single or of one length, a pack kernel of their own
    all_u32 = pick(4) == 0
    na = 1 if shape == "single" else 1 + pick(max_arrays)
    budget = 1 << 27                                        # values per trial
    cols, kinds = [], []
    same_lg = 1 + pick(max_log2 if shape == "single" else max(1, max_log2 - 4))
    same_n = max(1, (1 << same_lg) - pick(1 << max(0, same_lg - 1)))
    for _ in range(na):
        lg = 1 + pick(max_log2)
        n = max(1, (1 << lg) - pick(1 << max(0, lg - 1)))
        if shape != "ragged":
            n = same_n
        if n > budget:
            break
        budget -= n
        k = KINDS[pick(len(KINDS))]
        v = make_array(torch, g, dev, k, n)
        if all_u32:
            v = (((v & 0xFFFFFFFF) ^ 0x80000000) - 0x80000000).to(torch.int32)   # the low 32 bits as a u32 pattern
        elif k in ("two",) and pick(2):
            v = v.to(torch.int32)                           # enum tags
        cols.append(v)
        kinds.append(k)
    outs = [torch.empty(device.encode_bound(c.numel()), dtype=torch.uint8, device=dev) for c in cols]
    ek = [L.ELEM_U32 if c.dtype == torch.int32 else L.ELEM_I64 for c in cols]
    edesc = device.make_encode_descs([(c.data_ptr(), c.numel(), o.data_ptr(), o.numel(), k) for c, o, k in zip(cols, outs, ek)])
    res = device.new_results(len(cols), dev)
    device.encode_batch_dev(edesc, res)
    rs = device.read_results(res)
    for i, (c, r) in enumerate(zip(cols, rs)):
        mid, w = expected(torch, c.to(torch.int64) & 0xFFFFFFFF if c.dtype == torch.int32 else c)
        assert r.status == 0 and r.midpoint == mid and r.width == w, (seed, i, kinds[i], c.numel(), r.status, r.midpoint, mid, r.width, w)
        n = c.numel()
        assert r.out_count == 13 + (8 * n if w >= 32 else (n * w + 7) // 8)
    back = [torch.empty_like(c) for c in cols]
    ddesc = device.make_descs([(o.data_ptr(), int(r.out_count), c.numel(), b.data_ptr(), k) for o, r, c, b, k in zip(outs, rs, cols, back, ek)])
    dres = device.new_results(len(cols), dev)
    device.decode_batch_dev(ddesc, dres)
    for i, (c, b, r) in enumerate(zip(cols, back, device.read_results(dres))):
        assert r.status == 0 and r.out_count == c.numel() and torch.equal(c, b), (seed, i, kinds[i])
    if verbose:
        print(f"seed {seed}: {shape}{' u32' if all_u32 else ''}, {len(cols)} arrays, {sum(c.numel() for c in cols)} values ok", flush=True)
    return len(cols)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--trials", type=int, default=40)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--max-log2", type=int, default=23)
    ap.add_argument("--max-arrays", type=int, default=300)
    args = ap.parse_args()
    import torch

    from zbra_amd import _lib, device
    total = 0
    for t in range(args.trials):
        total += run_trial(torch, device, _lib, args.seed + t, args.max_log2, args.max_arrays, verbose=True)
    print(f"{args.trials} trials, {total} arrays: all midpoints, widths and round trips exact")


if __name__ == "__main__":
    main()
