"""Device-resident timing of the 1000-pattern scan in its three modes (positions, -c lines, -c -o count)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import krep_amd
from krep_amd import abi
import bench

gib = float(sys.argv[1]) if len(sys.argv) > 1 else 8.0
n = int(gib * (1 << 30))
e = krep_amd.load()
pats = bench.ac_patterns()
buf = torch.empty(n + 64, dtype=torch.uint8, device="cuda")
e.generate(buf.data_ptr(), n, 0, 4, bench.SEED, bench.pack_dict(pats), 4096)
cap = n // 1500
pos = torch.empty(2 * cap, dtype=torch.int64, device="cuda")
for name, kw, want_pos in (("positions", {}, True), ("-c (lines)", dict(count_lines=True), False),
                           ("-c -o (matches)", dict(count_lines=True, only_match=True), False),
                           ("-i positions", dict(case_sensitive=False), True), ("-w positions", dict(whole_word=True), True),
                           ("-w -c -o", dict(whole_word=True, count_lines=True, only_match=True), False)):
    plan = e.plan(abi.Params(pats, **kw))
    best = 1e9
    for _ in range(4):
        out = plan.scan(buf.data_ptr(), n, 0, n, 0, pos.data_ptr() if want_pos else 0, cap if want_pos else 0, time_it=True)
        best = min(best, out.kernel_ms)
    print(f"{name:18s} count={out.count} best={best:.3f} ms  {n / best / 1e6:.0f} GB/s", flush=True)
    plan.close()
