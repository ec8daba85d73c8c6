"""Single byte at several densities, with records and counting (development aid).  usage: dense_byte_bench.py <gib>"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import krep_amd
from krep_amd import abi
n = int(float(sys.argv[1]) * (1 << 30)) if len(sys.argv) > 1 else 8 << 30
e = krep_amd.load()
buf = torch.empty(n + 64, dtype=torch.uint8, device="cuda")
e.generate(buf.data_ptr(), n, 0, 2, 42, b"Sherlock", 10000)
cap = n // 8
pos = torch.empty(2 * cap, dtype=torch.int64, device="cuda")
for pat in (b"S", b"k", b"h", b"a", b"e", b" "):
    row = []
    for name, kw, wp in (("offsets", {}, True), ("-c -o", dict(count_lines=True, only_match=True), False), ("-c", dict(count_lines=True), False)):
        plan = e.plan(abi.Params([pat], **kw))
        best = 1e9
        for _ in range(4):
            out = plan.scan(buf.data_ptr(), n, 0, n, 0, pos.data_ptr() if wp else 0, cap if wp else 0, time_it=True)
            best = min(best, out.kernel_ms)
        if name == "-c -o": cnt = out.count
        row.append(f"{name} {n / best / 1e6:5.0f}")
        plan.close()
    print(f"{pat!r:6} {cnt:12d} ({cnt / n * 100:5.2f} % of bytes)   " + "   ".join(row) + "  GB/s", flush=True)
