"""Candidate rate of the multi-pattern stride-2 filter on the bench text, modelled on the host (development aid).
One plane: bit of the exact class 4-gram (c(p-3) .. c(p)).  Two planes: slot of (c(p-2), c(p-1), c(p)) hashed to 14 bits,
plane 0 bit c(p-3), plane 1 bit c(p+1) (kg_ac.hip, table build).  usage: python tools/ac_filter_model.py [MiB]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from krep_amd.engine import Engine
import bench

n = int(sys.argv[1]) << 20 if len(sys.argv) > 1 else 64 << 20
wl = bench.workload("ac1000")
buf = torch.empty(n + 64, dtype=torch.uint8, device="cuda")
Engine().generate(buf.data_ptr(), n, 0, wl["kind"], 42, wl["plant"], wl["period"])
t = buf[:n].cpu().numpy()
c = (t & 31).astype(np.uint32)
pats = [np.frombuffer(p, dtype=np.uint8) & 31 for p in wl["patterns"]]

def reg(c0, c1, c2, c3):
    return c0 | (c1 << 5) | (c2 << 16) | (c3 << 21)
def slot64(u):
    return (((u >> 2) ^ (u >> 12)) & 0x1fff8) >> 3

one = np.zeros(1 << 20, dtype=bool)              # exact 4-gram (c0 | c1<<5 | c2<<10 | c3<<15)
p0 = np.zeros((1 << 14, 32), dtype=bool)
p1 = np.zeros((1 << 14, 32), dtype=bool)
kinds = {"A": 0, "B5": 0, "B4": 0}
def g20(c0, c1, c2, c3):
    return int(c0) | (int(c1) << 5) | (int(c2) << 10) | (int(c3) << 15)
for p in pats:
    m = len(p)
    a = p[m - 4:]
    one[g20(*a)] = True
    s = slot64(reg(*[int(x) for x in a]))
    p0[s, a[0]] = True; p1[s, :] = True
    if m >= 5:
        b = p[m - 5:m - 1]
        one[g20(*b)] = True
        s = slot64(reg(*[int(x) for x in b]))
        p0[s, b[0]] = True; p1[s, p[m - 1]] = True
    else:
        for x in range(32):
            one[g20(x, p[0], p[1], p[2])] = True
        s = slot64(reg(0, int(p[0]), int(p[1]), int(p[2])))
        p0[s, :] = True; p1[s, p[3]] = True
pos = np.arange(3, n - 2, 2, dtype=np.int64) | 1    # odd tested positions
pos = pos[(pos >= 3) & (pos < n - 1)]
c0, c1, c2, c3, c4 = c[pos - 3], c[pos - 2], c[pos - 1], c[pos], c[pos + 1]
h1 = one[c0 | (c1 << 5) | (c2 << 10) | (c3 << 15)]
s = slot64(reg(c0, c1, c2, c3))
h2 = p0[s, c0] & p1[s, c4]
print(f"tested positions {len(pos)}  one plane: {h1.sum()} candidates ({h1.mean() * 100:.3f} %)   "
      f"two planes: {h2.sum()} ({h2.mean() * 100:.3f} %)  plane 0 alone {p0[s, c0].mean() * 100:.3f} %")
print(f"table load: one plane {one.sum()} bits of 2^20; plane 0 {p0.sum()} of 2^19, plane 1 {p1.sum()} of 2^19")
