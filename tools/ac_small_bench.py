"""Device-resident rate of small dictionaries (development aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import krep_amd
from krep_amd import abi

n = int(float(sys.argv[1]) * (1 << 30)) if len(sys.argv) > 1 else 8 << 30
e = krep_amd.load()
buf = torch.empty(n + 64, dtype=torch.uint8, device="cuda")
e.generate(buf.data_ptr(), n, 0, 2, 42, b"Sherlock", 10000)
cap = n // 500
pos = torch.empty(2 * cap, dtype=torch.int64, device="cuda")
for pats in ([b"error", b"warning", b"fatal"], [b"Sherlock", b"Holmes"], [b"he", b"she", b"hers"], [b"a", b"Sherlock"],
             [b"qzx", b"Sherlock", b"Watson"], [b"xq", b"zj"]):
    for name, kw, wp in (("pos", {}, True), ("-c -o", dict(count_lines=True, only_match=True), False)):
        plan = e.plan(abi.Params(pats, **kw))
        best = 1e9
        for _ in range(3):
            out = plan.scan(buf.data_ptr(), n, 0, n, 0, pos.data_ptr() if wp else 0, cap if wp else 0, time_it=True)
            best = min(best, out.kernel_ms)
        print(pats, name, f"{n / best / 1e6:.0f} GB/s", "count", out.count, "overflow", out.overflow, flush=True)
        plan.close()
