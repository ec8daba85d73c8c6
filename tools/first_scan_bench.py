"""First-scan performance of a fresh plan (what a one-shot caller such as the CLI gets) against the same plan's later scans, for
literals whose density decides the road (staging slots / one-pass record writers and their ring shape): VERDICT r05 item 4.
  usage: python tools/first_scan_bench.py [GiB]      ($KREP_GPU_NO_FIRST_LOOK=1: the round-5 behaviour)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import krep_amd
from krep_amd import abi

gib = float(sys.argv[1]) if len(sys.argv) > 1 else 32.0
n = int(gib * (1 << 30))
e = krep_amd.load()
buf = torch.empty(n + 64, dtype=torch.uint8, device="cuda")
cap = n // 48
pos = torch.empty(2 * cap, dtype=torch.int64, device="cuda")
e.generate(buf.data_ptr(), n, 0, 2, 42, b"Sherlock Holmes", 10000)
torch.cuda.synchronize()
print(f"# {gib:g} GiB, records wanted, a FRESH plan per row: GB/s of the 1st, 2nd, 3rd, 4th scan; look = with the first look, r05 = $KREP_GPU_NO_FIRST_LOOK=1")
for pat, kw in ((b"Sh", dict(case_sensitive=False)), (b" a", {}), (b"ab", {}), (b"the", {}), (b"and ", {}), (b"q", {}), (b"e", {}), (b"Sherlock", {})):
    row = []
    for mode in ("look", "r05"):
        if mode == "r05":
            os.environ["KREP_GPU_NO_FIRST_LOOK"] = "1"
        plan = e.plan(abi.Params([pat], **kw))
        ts = []
        for _ in range(4):
            out = plan.scan(buf.data_ptr(), n, 0, n, 0, pos.data_ptr(), cap, time_it=True)
            ts.append(out.kernel_ms)
        plan.close()
        os.environ.pop("KREP_GPU_NO_FIRST_LOOK", None)
        row.append(f"{mode}: " + " / ".join(f"{n / t / 1e6:5.0f}" for t in ts) + f"  (first {ts[0]:.2f} ms = {min(ts) / ts[0] * 100:.0f} % of best)")
        cnt = out.count
    print(f"{pat!r:12} {'-i' if kw else '  '} {cnt:>11} hits   " + "   |   ".join(row), flush=True)
