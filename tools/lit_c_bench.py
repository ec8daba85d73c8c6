"""-c (distinct lines) and -c -o (matches) of the 8-byte literal, device resident (profiling aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, krep_amd
from krep_amd import abi
n = int(float(sys.argv[1]) * (1 << 30)) if len(sys.argv) > 1 else 8 << 30
e = krep_amd.load()
buf = torch.empty(n + 64, dtype=torch.uint8, device="cuda")
e.generate(buf.data_ptr(), n, 0, 2, 42, b"Sherlock", 10000)
for name, kw in (("-c", dict(count_lines=True)), ("-c -o", dict(count_lines=True, only_match=True))):
    plan = e.plan(abi.Params([b"Sherlock"], **kw))
    best = 1e9
    for _ in range(3):
        out = plan.scan(buf.data_ptr(), n, 0, n, 0, 0, 0, time_it=True)
        best = min(best, out.kernel_ms)
    print(name, f"{n / best / 1e6:.0f} GB/s", out.count, flush=True)
