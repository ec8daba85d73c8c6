"""One arena (text + record buffer in ONE allocation, made first in a fresh process) against separate allocations: does it take
the placement lottery out of the offsets-producing scans?  (development aid)  usage: python tools/arena_probe.py <gib> <mode: arena|separate|posfirst>"""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from krep_amd import abi
from krep_amd.engine import Engine
import bench

gib, mode = float(sys.argv[1]), sys.argv[2]
n = int(gib * (1 << 30))
e = Engine()
cap = n // 80 + 4096
pos_bytes = cap * 16
if mode == "arena":
    arena = torch.empty(n + (1 << 21) + pos_bytes, dtype=torch.uint8, device="cuda")
    text_ptr = arena.data_ptr()
    pos_ptr = (arena.data_ptr() + n + 64 + (1 << 21) - 1) & ~((1 << 21) - 1)
elif mode == "posfirst":
    pos = torch.empty(2 * cap, dtype=torch.int64, device="cuda")
    buf = torch.empty(n + 64, dtype=torch.uint8, device="cuda")
    text_ptr, pos_ptr = buf.data_ptr(), pos.data_ptr()
else:
    buf = torch.empty(n + 64, dtype=torch.uint8, device="cuda")
    pos = torch.empty(2 * cap, dtype=torch.int64, device="cuda")
    text_ptr, pos_ptr = buf.data_ptr(), pos.data_ptr()
for name in ("memchr1", "literal8"):
    wl = bench.workload(name)
    e.generate(text_ptr, n, 0, wl["kind"], bench.SEED, wl["plant"], wl["period"])
    pl = e.plan(abi.Params(wl["patterns"]))
    ts = []
    for rep in range(8):
        out = pl.scan(text_ptr, n, 0, n, 0, pos_ptr, cap, time_it=True)
        if rep >= 2:
            ts.append(out.kernel_ms)
    print(f"{mode:9s} {name}: median {statistics.median(ts):.3f} ms  min {min(ts):.3f}  frac {n / statistics.median(ts) / 1e6 / 8000:.4f}", flush=True)
    pl.close()
