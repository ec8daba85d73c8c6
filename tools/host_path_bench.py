"""End-to-end rate of the host-buffer operator (PCIe-inclusive): krep_gpu_literal_search on a host numpy buffer —
one piece (stage everything, then scan) against streamed pieces (H2D of piece k+1 under the scan of piece k), next to the
device-resident scan time of the same bytes: the scan hides under the PCIe time.  usage: host_path_bench.py [GiB]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import krep_amd
from krep_amd import abi

gib = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
n = int(gib * (1 << 30))
e = krep_amd.load()
text = e.generate_host(1 << 20, 0, 2, 42, b"Sherlock", 10000)
text = np.tile(text, n // text.size)
n = text.size
p = abi.Params([b"Sherlock"], count_lines=True, only_match=True)
pp = abi.Params([b"Sherlock"])


def run(label, chunk, params, want_result):
    e.set_stream_chunk(chunk)
    best = 1e9
    for rep in range(4):
        t0 = time.perf_counter()
        ret, pos = e.search(params, text, want_result=want_result)
        best = min(best, time.perf_counter() - t0)
    print(f"{label:58s} {n / 2**30:.1f} GiB  count={ret}  best {best * 1e3:7.1f} ms  {n / best / 1e9:6.1f} GB/s (PCIe-inclusive)", flush=True)
    return best


t_one = run("one piece (stage all, then scan), count", 1 << 42, p, False)
t_def = run("streamed, 128 MiB pieces (default), count", 0, p, False)
t_64 = run("streamed, 64 MiB pieces, count", 64 << 20, p, False)
run("streamed, 64 MiB pieces, offsets", 64 << 20, pp, True)
e.set_stream_chunk(0)
d = torch.from_numpy(text[: min(n, 4 << 30)]).cuda()
plan = e.plan(p)
best = min(plan.scan(d.data_ptr(), d.numel(), time_it=True).kernel_ms for _ in range(5))
scan_ms = best * n / d.numel()
print(f"device-resident scan of the same bytes: {scan_ms:.2f} ms = {100 * scan_ms / (t_64 * 1e3):.1f} % of the streamed end-to-end time "
      f"(one piece spends it AFTER the copy: {t_one * 1e3:.1f} ms vs {t_64 * 1e3:.1f} ms streamed)")
