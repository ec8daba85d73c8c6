"""What alternates from one process to the next?  (literal8 5.17 / 5.29 ms and ac1000 6.74 / 6.96 ms in strictly
alternating processes, profiles/r04_placement.txt §3.)  Per process: the clocks / power state before and after (rocm-smi), then
the same 32 GiB scan timed on the null stream and on six freshly created streams (each a different hardware queue / pipe), and
once more after the haystack was re-allocated.  usage: python tools/mode_probe.py [N processes = 6] [GiB = 32]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if os.environ.get("MODE_PROBE_CHILD"):
    sys.path.insert(0, ROOT)
    import torch, krep_amd, bench
    from krep_amd import abi
    gib = float(sys.argv[2]) if len(sys.argv) > 2 else 32.0
    n = int(gib * (1 << 30))
    e = krep_amd.load()
    def smi():
        try:
            out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showperflevel"], capture_output=True, text=True, timeout=20).stdout
            keep = [l.split(":", 1)[-1].strip() for l in out.splitlines() if any(k in l for k in ("sclk", "mclk", "fclk", "socclk", "Power", "Performance Level"))]
            return "; ".join(keep)[:300]
        except Exception as ex:
            return f"rocm-smi unavailable ({ex})"
    print("  before:", smi(), flush=True)
    buf = torch.empty(n + 64, dtype=torch.uint8, device="cuda")
    wl2, wl4 = bench.workload("literal8"), bench.workload("ac1000")
    pos = torch.empty(2 * (n // 1500 + 4096), dtype=torch.int64, device="cuda")
    def timed(plan, stream, cap):
        ms = sorted(plan.scan(buf.data_ptr(), n, 0, n, 0, pos.data_ptr(), cap, stream, True).kernel_ms for _ in range(7))
        return ms[3]
    for name, wl in (("literal8", wl2), ("ac1000", wl4)):
        e.generate(buf.data_ptr(), n, 0, wl["kind"], 42, wl["plant"], wl["period"])
        plan = e.plan(abi.Params(wl["patterns"]))
        cap = n // 1500 + 4096
        row = [f"null {timed(plan, 0, cap):.3f}"]
        streams = [torch.cuda.Stream() for _ in range(6)]
        for i, s in enumerate(streams):
            torch.cuda.synchronize()
            row.append(f"s{i} {timed(plan, s.cuda_stream, cap):.3f}")
        print(f"  {name:9s} ms by stream: " + "  ".join(row), flush=True)
        plan.close()
    # the same after the haystack moved to a new allocation
    del buf
    torch.cuda.empty_cache()
    junk = torch.empty(3 << 30, dtype=torch.uint8, device="cuda")
    buf = torch.empty(n + 64, dtype=torch.uint8, device="cuda")
    e.generate(buf.data_ptr(), n, 0, wl2["kind"], 42, wl2["plant"], wl2["period"])
    plan = e.plan(abi.Params(wl2["patterns"]))
    print(f"  literal8 after re-allocating the haystack: null {timed(plan, 0, n // 1500 + 4096):.3f}", flush=True)
    print("  after: ", smi(), flush=True)
    sys.exit(0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 6
for i in range(N):
    print(f"process {i}:", flush=True)
    subprocess.run([sys.executable, __file__] + sys.argv[1:], env=dict(os.environ, MODE_PROBE_CHILD="1"))
