"""Device-resident rate of the literal scan across pattern lengths and modes (development aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import krep_amd
from krep_amd import abi

gib = float(sys.argv[1]) if len(sys.argv) > 1 else 8.0
n = int(gib * (1 << 30))
e = krep_amd.load()
buf = torch.empty(n + 64, dtype=torch.uint8, device="cuda")
cap = n // 2000
pos = torch.empty(2 * cap, dtype=torch.int64, device="cuda")
base = b"Sherlock Holmes and the hound of the Baskervilles went out to sea"
lens = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [2, 3, 4, 5, 7, 8, 9, 12, 16, 17, 24, 32, 33, 48, 64, 65, 128]
for m in lens:
    pat = (base * 2)[:m]
    e.generate(buf.data_ptr(), n, 0, 2, 42, pat, 10000)
    torch.cuda.synchronize()
    row = [f"m={m:3d}"]
    for name, kw, want_pos in (("pos", {}, True), ("pos -i", dict(case_sensitive=False), True),
                               ("pos -w", dict(whole_word=True), True), ("-c", dict(count_lines=True), False),
                               ("-c -o", dict(count_lines=True, only_match=True), False)):
        try:
            plan = e.plan(abi.Params([pat], **kw))
        except Exception as ex:
            row.append(f"{name}: n/a")
            continue
        best = 1e9
        for _ in range(int(os.environ.get("LS_REPS", "3"))):
            out = plan.scan(buf.data_ptr(), n, 0, n, 0, pos.data_ptr() if want_pos else 0, cap if want_pos else 0, time_it=True)
            best = min(best, out.kernel_ms)
        row.append(f"{name}: {n / best / 1e6:5.0f} ({out.count})")
        plan.close()
    print("  ".join(row), flush=True)
