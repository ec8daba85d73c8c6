import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, krep_amd
from krep_amd import abi
n = 8 << 30
e = krep_amd.load()
buf = torch.empty(n + 64, dtype=torch.uint8, device="cuda")
cap = n // 2000
pos = torch.empty(2 * cap, dtype=torch.int64, device="cuda")
for pat in (b"Sherlock", b"abcabcab", b"aaaa", b"abab", b"xyzxyzxyzxyzx"):
    e.generate(buf.data_ptr(), n, 0, 2, 42, pat, 10000)
    torch.cuda.synchronize()
    for name, kw, wp in (("pos", {}, True), ("-c -o", dict(count_lines=True, only_match=True), False), ("-c", dict(count_lines=True), False)):
        plan = e.plan(abi.Params([pat], **kw))
        best = 1e9
        for _ in range(3):
            out = plan.scan(buf.data_ptr(), n, 0, n, 0, pos.data_ptr() if wp else 0, cap if wp else 0, time_it=True)
            best = min(best, out.kernel_ms)
        print(pat, name, "ref_algo", plan.ref_algo, f"{n/best/1e6:.0f} GB/s", out.count, flush=True)
        plan.close()
