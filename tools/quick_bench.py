"""Quick device-resident timing of the literal scan (development aid; bench.py is the contract)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import krep_amd
from krep_amd import abi

gib = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
kind = int(sys.argv[2]) if len(sys.argv) > 2 else 2
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
n = int(gib * (1 << 30))
e = krep_amd.load()
dev = torch.device("cuda:0")
buf = torch.empty(n + 64, dtype=torch.uint8, device=dev)
pat = b"Sherlock" if kind == 2 else b"#"
e.generate(buf.data_ptr(), n, 0, kind, 42, pat, 10000 if kind == 2 else 0)
torch.cuda.synchronize()
for mode in ("count", "pos"):
    p = abi.Params([pat], count_lines=(mode == "count"), only_match=(mode == "count"))
    plan = e.plan(p)
    cap = n // (50 if kind == 3 else 4000) + 1024
    pos = torch.empty(cap * 2, dtype=torch.int64, device=dev) if mode == "pos" else None
    best = 1e9
    for r in range(reps):
        out = plan.scan(buf.data_ptr(), n, d_positions=pos.data_ptr() if pos is not None else 0,
                        capacity=cap if pos is not None else 0, time_it=True)
        best = min(best, out.kernel_ms)
    print(f"kind={kind} mode={mode} n={gib}GiB count={out.count} total={out.total_matches} stored={out.stored} "
          f"overflow={out.overflow} best={best:.3f} ms  {n/best/1e6:.1f} GB/s", flush=True)
    plan.close()
