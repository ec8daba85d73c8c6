"""Dictionary SHAPES on 32 GiB of BASELINE config 4's text (i.i.d. letters, the 1000 random patterns planted): what a short or a long pattern
beside the thousand does to the end-gram kernel.  A sweep for cliffs (round 6).   usage: python tools/iid_dict_shapes.py [gib]"""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, krep_amd, bench
from krep_amd import abi

gib = float(sys.argv[1]) if len(sys.argv) > 1 else 32.0
n = int(gib * (1 << 30))
e = krep_amd.load()
base = bench.ac_patterns()
buf = torch.empty(n + 64, dtype=torch.uint8, device="cuda")
e.generate(buf.data_ptr(), n, 0, 4, bench.SEED, bench.pack_dict(base), 4096)
cap = n // 200
pos = torch.empty(2 * cap, dtype=torch.int64, device="cuda")
SHAPES = [("BASELINE config 4 (1000 patterns, 4-16 B)", base), ("... + one 3-byte pattern", base + [b"qzx"]), ("... + one 2-byte pattern", base + [b"qz"]),
          ("... + one 1-byte pattern", base + [b"#"]), ("... + one 24-byte pattern", base + [b"abcdefghijklmnopqrstuvwx"]), ("100 of them", base[:100]), ("10 of them", base[:10])]
print(f"# {gib:g} GiB, offsets produced / -c -o / -c (lines): median kernel ms of four scans after the first (GB/s of text)")
for name, pats in SHAPES:
    row = []
    for mname, kw, wp in (("offsets", {}, True), ("-c -o", dict(count_lines=True, only_match=True), False), ("-c", dict(count_lines=True), False)):
        try:
            plan = e.plan(abi.Params(pats, **kw))
            ts = []
            for i in range(5):
                out = plan.scan(buf.data_ptr(), n, 0, n, 0, pos.data_ptr() if wp else 0, cap if wp else 0, time_it=True)
                ts.append(out.kernel_ms)
            st = plan.split_state()
            plan.close()
            t = statistics.median(ts[1:])
            row.append(f"{mname} {t:7.2f} ({n / t / 1e6:5.0f}){' OVERFLOW' if out.overflow else ''}")
        except Exception as ex:
            row.append(f"{mname} failed: {str(ex)[:50]}")
    print(f"{name:44s} {len(pats):5d} patterns  " + "   ".join(row), flush=True)
