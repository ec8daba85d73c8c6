"""Does the speed of the offsets-producing scan depend on WHERE its scratch lands?  One process, one library; the scratch is
released and re-allocated behind junk allocations of varying size, and each placement is timed (development aid).
usage: python tools/placement_probe.py <gib> <trials>"""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["KREP_GPU_DEBUG_ALLOC"] = "1"
import torch
from krep_amd import abi
from krep_amd.engine import Engine
import bench

gib, trials = float(sys.argv[1]), int(sys.argv[2])
n = int(gib * (1 << 30))
e = Engine()
wl = bench.workload("literal8")
buf = torch.empty(n + 64, dtype=torch.uint8, device="cuda")
e.generate(buf.data_ptr(), n, 0, wl["kind"], 42, wl["plant"], wl["period"])
cap = n // 1500 + 4096
pos = torch.empty(2 * cap, dtype=torch.int64, device="cuda")
print(f"text {buf.data_ptr():#x} pos {pos.data_ptr():#x}", flush=True)
junk = []
for t in range(trials):
    pl = e.plan(abi.Params(wl["patterns"]))
    ts = []
    for rep in range(5):
        out = pl.scan(buf.data_ptr(), n, 0, n, 0, pos.data_ptr(), cap, time_it=True)
        if rep:
            ts.append(out.kernel_ms)
    print(f"trial {t}: median {statistics.median(ts):.3f} ms  min {min(ts):.3f}", flush=True)
    pl.close() if hasattr(pl, "close") else None
    del pl
    e.release_device_resources()
    junk.append(torch.empty((3 + 5 * t) << 20, dtype=torch.uint8, device="cuda"))  # perturb the next placement
