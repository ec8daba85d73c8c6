"""Can a read-bound scan and a write-bound record gather share the HBM?  (design probe for pipelining dense outputs)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, krep_amd
from krep_amd import abi
n = 32 << 30
e = krep_amd.load()
buf = torch.empty(n + 64, dtype=torch.uint8, device="cuda")
e.generate(buf.data_ptr(), n, 0, 3, 42, b"#", 0)
out = torch.empty(5_500_000_000 // 8, dtype=torch.int64, device="cuda")
plan = e.plan(abi.Params([b"#"], count_lines=True, only_match=True))
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
def scan():
    plan.scan(buf.data_ptr(), n, 0, n, 0, 0, 0, stream=sa.cuda_stream)
def fill():
    with torch.cuda.stream(sb):
        out.fill_(7)
for name, fn in (("scan alone", lambda: scan()), ("fill 5.5 GB alone", lambda: (fill(), sb.synchronize())),
                 ("both", lambda: (fill(), scan(), sb.synchronize()))):
    best = 1e9
    for _ in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    print(f"{name:20s} {best * 1e3:.2f} ms")
