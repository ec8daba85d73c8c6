"""Does the speed of the one-pass single-byte scan depend on WHERE the record buffer (and the ticket arrays) land?  One
process, the text stays; the position buffer is re-allocated behind junk allocations of varying size and each placement is
timed; then the same with the TEXT re-allocated (development aid).  usage: python tools/placement_probe_m1.py <gib> <trials>"""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from krep_amd import abi
from krep_amd.engine import Engine
import bench

gib, trials = float(sys.argv[1]), int(sys.argv[2])
n = int(gib * (1 << 30))
e = Engine()
wl = bench.workload("memchr1")
cap = n // 80 + 4096


def timed(buf, pos, tag):
    pl = e.plan(abi.Params(wl["patterns"]))
    ts = []
    for rep in range(6):
        out = pl.scan(buf.data_ptr(), n, 0, n, 0, pos.data_ptr(), cap, time_it=True)
        if rep:
            ts.append(out.kernel_ms)
    print(f"{tag}: text {buf.data_ptr():#x} pos {pos.data_ptr():#x}  median {statistics.median(ts):.3f} ms  min {min(ts):.3f}", flush=True)
    pl.close()


buf = torch.empty(n + 64, dtype=torch.uint8, device="cuda")
e.generate(buf.data_ptr(), n, 0, wl["kind"], bench.SEED, wl["plant"], wl["period"])
junk = []
for t in range(trials):
    pos = torch.empty(2 * cap, dtype=torch.int64, device="cuda")
    timed(buf, pos, f"pos placement {t}")
    del pos
    torch.cuda.empty_cache()
    junk.append(torch.empty((7 + 61 * t) << 20, dtype=torch.uint8, device="cuda"))
pos = torch.empty(2 * cap, dtype=torch.int64, device="cuda")
for t in range(max(2, trials // 2)):
    del buf
    torch.cuda.empty_cache()
    junk.append(torch.empty((13 + 97 * t) << 20, dtype=torch.uint8, device="cuda"))
    buf = torch.empty(n + 64, dtype=torch.uint8, device="cuda")
    e.generate(buf.data_ptr(), n, 0, wl["kind"], bench.SEED, wl["plant"], wl["period"])
    timed(buf, pos, f"text placement {t}")
