"""Batch encode at scale: midpoints and widths of a many-array submission against a sort per array."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from zbra_amd import device, _lib

dev = torch.device("cuda", 0)
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 42
n = int(sys.argv[2]) if len(sys.argv) > 2 else (1 << 22)
mix = sys.argv[3] if len(sys.argv) > 3 else "tvw"
g = torch.Generator(device=dev); g.manual_seed(1)
vals = []
for b in range(nb):
    tags = torch.randint(0, 2, (n,), dtype=torch.int64, device=dev, generator=g)
    n0 = int((tags == 0).sum())
    if "t" in mix: vals.append(tags.to(torch.int32))
    if "T" in mix: vals.append(tags.clone())
    if "u" in mix: vals.append(torch.randint(0, 4000, (n,), dtype=torch.int64, device=dev, generator=g).to(torch.int32))
    if "h" in mix: vals.append(tags[: n // 2].to(torch.int32))
    if "v" in mix: vals.append(1_700_000_000_000 + torch.randint(0, 4000, (n0,), dtype=torch.int64, device=dev, generator=g))
    if "w" in mix: vals.append(torch.randint(-(1 << 24), 1 << 24, (n - n0,), dtype=torch.int64, device=dev, generator=g))
na = len(vals)
outs = [torch.empty(device.encode_bound(v.numel()), dtype=torch.uint8, device=dev) for v in vals]
want = []
for v in vals:
    s, _ = torch.sort(v.to(torch.int64))
    m = v.numel()
    if m % 2: want.append(int(s[m // 2]))
    else:
        a, b = int(s[m // 2 - 1]), int(s[m // 2])
        d = b - a
        want.append(a + (d // 2 if d >= 0 else -((-d) // 2)))
descs = device.make_encode_descs([(v.data_ptr(), v.numel(), o.data_ptr(), o.numel(), _lib.ELEM_I64 if v.dtype == torch.int64 else _lib.ELEM_U32)
                                  for v, o in zip(vals, outs)])
for rep in range(3):
    res = device.new_results(na, dev)
    device.encode_batch_dev(descs, res)
    rs = device.read_results(res)
    bad = [(i, rs[i].midpoint, want[i], rs[i].width, rs[i].status) for i in range(na) if rs[i].midpoint != want[i]]
    print("mix", mix, "arrays", na, "batch rep", rep, "bad", len(bad), bad[:4])
