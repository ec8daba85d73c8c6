"""One dictionary, one mode, a few scans (for rocprofv3 / counters).  usage: dict_probe.py <gib> <mode pos|count|lines> pat [pat ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import krep_amd
from krep_amd import abi
n = int(float(sys.argv[1]) * (1 << 30)); mode = sys.argv[2]; pats = [p.encode() for p in sys.argv[3:]]
e = krep_amd.load()
buf = torch.empty(n + 64, dtype=torch.uint8, device="cuda")
e.generate(buf.data_ptr(), n, 0, 2, 42, b"Sherlock", 10000)
cap = n // 4 if mode == "pos" else 0
pos = torch.empty(2 * cap, dtype=torch.int64, device="cuda") if cap else None
kw = dict(count_lines=True, only_match=True) if mode == "count" else dict(count_lines=True) if mode == "lines" else {}
plan = e.plan(abi.Params(pats, **kw))
for _ in range(3):
    out = plan.scan(buf.data_ptr(), n, 0, n, 0, pos.data_ptr() if cap else 0, cap, time_it=True)
    print(pats, mode, f"{out.kernel_ms:.3f} ms {n / out.kernel_ms / 1e6:.0f} GB/s count {out.count}", flush=True)
