"""N fresh processes of the default bench line (no CPU baseline): per process the kernel time and roofline fraction of the three
BASELINE workloads and what the placement draw saw.  usage: python tools/run_to_run.py [N] [extra bench.py args...]
       python tools/run_to_run.py 6                      -> first allocations
       python tools/run_to_run.py 6 --placement-tries 4  -> through the library's placed allocator (krep_gpu_alloc_placed)"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
extra = sys.argv[2:]
for i in range(n):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--steps", "10", "--warmup", "2"] + extra,
                       capture_output=True, text=True)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if not lines:
        print(f"process {i}: failed: {r.stderr[-300:]}")
        continue
    j = json.loads(lines[-1])
    parts = [f"literal8 {j['roofline']['kernel_ms']:.3f} ms {j['roofline']['frac']:.4f}"]
    for k, e in j.get("extra", {}).items():
        if isinstance(e, dict) and "roofline" in e:
            parts.append(f"{k} {e['roofline']['kernel_ms']:.3f} ms {e['roofline']['frac']:.4f}")
    pl = j.get("placement") or {}
    print(f"process {i}: " + " | ".join(parts) + (f" | krep_gpu_alloc_placed: records ms {pl['records_ms']} kept {pl['kept']} (count-only {pl['count_only_ms']})"
                                                   if "records_ms" in pl else " | first allocation"), flush=True)
