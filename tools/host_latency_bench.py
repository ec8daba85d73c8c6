"""Latency of the host-buffer operators on small buffers (one search_func_t call, PCIe-inclusive)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import krep_amd
from krep_amd import abi

e = krep_amd.load()
base = e.generate_host(64 << 20, 0, 2, 42, b"Sherlock", 10000)
for pats, kw in (([b"Sherlock"], dict(count_lines=True, only_match=True)), ([b"Sherlock"], {}),
                 ([b"Sherlock", b"Holmes", b"Watson"], {})):
    p = abi.Params(pats, **kw)
    for n in (4 << 10, 256 << 10, 4 << 20, 64 << 20):
        text = base[:n]
        ts = []
        for _ in range(6):
            t0 = time.perf_counter()
            ret, pos = e.search(p, text, want_result=not kw)
            ts.append(time.perf_counter() - t0)
        print(f"{len(pats)} pattern(s) {'count' if kw else 'positions'} n={n >> 10:6d} KiB ret={ret:6d} first={ts[0] * 1e3:8.2f} ms "
              f"best={min(ts[1:]) * 1e3:7.3f} ms ({n / min(ts[1:]) / 1e9:6.2f} GB/s)", flush=True)
