"""Dense multi-pattern matches (the candidate queue floods): how slow is the exact fallback?  (development aid)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, krep_amd
from krep_amd import abi
n = 2 << 30
e = krep_amd.load()
buf = torch.empty(n + 64, dtype=torch.uint8, device="cuda")
e.generate(buf.data_ptr(), n, 0, 2, 42, b"Sherlock", 10000)
cap = n // 4
pos = torch.empty(2 * cap, dtype=torch.int64, device="cuda")
for pats in ([b"e", b"t"], [b"e", b"th", b"Sherlock"], [b"er", b"th", b"an"], [b"the", b"and", b"ing"]):
    for name, kw, wp in (("pos", {}, True), ("-c -o", dict(count_lines=True, only_match=True), False)):
        plan = e.plan(abi.Params(pats, **kw))
        best = 1e9
        for _ in range(2):
            out = plan.scan(buf.data_ptr(), n, 0, n, 0, pos.data_ptr() if wp else 0, cap if wp else 0, time_it=True)
            best = min(best, out.kernel_ms)
        print(pats, name, f"{n / best / 1e6:.0f} GB/s", "count", out.count, f"({out.count / n * 100:.2f} % of bytes)", "overflow", out.overflow, flush=True)
        plan.close()
