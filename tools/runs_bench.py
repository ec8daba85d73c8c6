"""`-c -o` of a pattern of one repeated byte through the greedy families (kg_runs.hip) against the list road ($KREP_GPU_NO_RUNS=1), one process
(VERDICT r05 weak #8).   usage: python tools/runs_bench.py [GiB]"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, krep_amd
from krep_amd import abi
gib = float(sys.argv[1]) if len(sys.argv) > 1 else 32.0
n = int(gib * (1 << 30))
e = krep_amd.load()
buf = torch.empty(n + 64, dtype=torch.uint8, device="cuda")
print(f"# {gib:g} GiB; count-only (-c -o), simd_sse42_search's greedy set; ms and GB/s, median of 4 after one")
for label, gen in (("i.i.d. text (15.6 % blanks)", lambda: e.generate(buf.data_ptr(), n, 0, 2, 42, b"Sherlock", 10000)), ("all 'a'", lambda: buf.fill_(97))):
    gen(); torch.cuda.synchronize()
    pats = [b"  ", b"   ", b"aa", b"ee"] if "i.i.d" in label else [b"aa", b"aaaaaaa"]
    for pat in pats:
        row = []
        for mode, env in (("runs", {}), ("list", {"KREP_GPU_NO_RUNS": "1"})):
            if mode == "list" and "all" in label and gib > 4:
                row.append("list: skipped (one cluster of 2^35 occurrences)"); continue
            os.environ.update(env)
            plan = e.plan(abi.Params([pat], count_lines=True, only_match=True))
            ts = []
            try:
                for i in range(5):
                    out = plan.scan(buf.data_ptr(), n, time_it=True)
                    if i: ts.append(out.kernel_ms)
                t = statistics.median(ts)
                row.append(f"{mode}: {t:8.2f} ms {n / t / 1e6:6.0f} GB/s (count {out.count})")
            except Exception as ex:
                row.append(f"{mode}: failed {str(ex)[:60]}")
            plan.close()
            for k in env: os.environ.pop(k, None)
        print(f"{label:28s} {pat!r:12} " + "   |   ".join(row), flush=True)
