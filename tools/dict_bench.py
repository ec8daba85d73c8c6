"""Device-resident rate of small dictionaries in the three modes, at one size (development aid; profiles/r04_dictionaries.txt).
usage: python tools/dict_bench.py <gib>   — set KREP_GPU_AC_NO_TINY=1 for the general kernel on the short ones"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import krep_amd
from krep_amd import abi

n = int(float(sys.argv[1]) * (1 << 30)) if len(sys.argv) > 1 else 8 << 30
e = krep_amd.load()
buf = torch.empty(n + 64, dtype=torch.uint8, device="cuda")
e.generate(buf.data_ptr(), n, 0, 2, 42, b"Sherlock", 10000)
cap = n // 12
pos = torch.empty(2 * cap, dtype=torch.int64, device="cuda")
DICTS = ([b"he", b"she", b"hers"], [b"xq", b"zj"], [b"the", b"and", b"ing"], [b"er", b"th", b"an"], [b"e", b"t"],
         [b"error", b"warning", b"fatal"], [b"Sherlock", b"Holmes"], [b"a", b"Sherlock"],
         [b"if", b"else", b"while"], [b"the", b"quick", b"brown"])  # several long lengths beside a short pattern
if len(sys.argv) > 2:  # a subset: indices, e.g. 8,9
    DICTS = [DICTS[int(i)] for i in sys.argv[2].split(",")]
for pats in DICTS:
    row = []
    for name, kw, wp in (("-c -o", dict(count_lines=True, only_match=True), False), ("offsets", {}, True),
                         ("-c", dict(count_lines=True), False), ("-i -c -o", dict(count_lines=True, only_match=True, case_sensitive=False), False)):
        plan = e.plan(abi.Params(pats, **kw))
        best = 1e9
        for _ in range(4):
            out = plan.scan(buf.data_ptr(), n, 0, n, 0, pos.data_ptr() if wp else 0, cap if wp else 0, time_it=True)
            best = min(best, out.kernel_ms)
        row.append(f"{name} {n / best / 1e6:5.0f}")
        cnt = out.count if name == "-c -o" else cnt
        plan.close()
    print(f"{b' '.join(pats).decode():24s} {cnt:11d} matches ({cnt / n * 100:5.2f} % of bytes)   " + "   ".join(row) + "   GB/s", flush=True)
