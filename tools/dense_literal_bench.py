"""Records of DENSE short literals (2..8 bytes): the two-pass staging road against the one-pass record writer (kg_single.hip,
MULTI), in one process on one text, with the count-only rate beside them (development aid).
  usage: python tools/dense_literal_bench.py [GiB]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import krep_amd
from krep_amd import abi

gib = float(sys.argv[1]) if len(sys.argv) > 1 else 8.0
n = int(gib * (1 << 30))
e = krep_amd.load()
buf = torch.empty(n + 64, dtype=torch.uint8, device="cuda")
cap = n // 64
pos = torch.empty(2 * cap, dtype=torch.int64, device="cuda")
e.generate(buf.data_ptr(), n, 0, 2, 42, b"Sherlock Holmes", 10000)
torch.cuda.synchronize()


def best(plan, want_pos, reps=3):
    b, out = 1e9, None
    for _ in range(reps):
        out = plan.scan(buf.data_ptr(), n, 0, n, 0, pos.data_ptr() if want_pos else 0, cap if want_pos else 0, time_it=True)
        b = min(b, out.kernel_ms)
    return b, out


for pat, kw in ((b"Sh", dict(case_sensitive=False)), (b" a", {}), (b"ab", {}), (b"the", {}), (b"th", dict(case_sensitive=False)),
                (b"and ", {}), (b"e t", {}), (b"Sherlock", {}), (b"ing th", {})):
    os.environ["KREP_GPU_NO_FUSED1"] = "1"
    plan = e.plan(abi.Params([pat], **kw))
    best(plan, True, 1)
    two, out2 = best(plan, True)
    plan.close()
    del os.environ["KREP_GPU_NO_FUSED1"]
    plan = e.plan(abi.Params([pat], **kw))
    best(plan, True, 1)  # (learns the density)
    l0 = e.single_launches()
    one, out1 = best(plan, True)
    took = e.single_launches() - l0
    plan.close()
    plan = e.plan(abi.Params([pat], **kw))
    cnt, outc = best(plan, False)
    plan.close()
    assert out1.count == out2.count == outc.count, (pat, out1.count, out2.count, outc.count)
    print(f"{pat!r:12} {'-i' if kw else '  '} hits {out1.count:>11} ({out1.count * 32768 / n:7.1f}/unit)  two-pass {n / two / 1e6:5.0f}  "
          f"one-pass {n / one / 1e6:5.0f} GB/s (launches {took})  count {n / cnt / 1e6:5.0f}", flush=True)
