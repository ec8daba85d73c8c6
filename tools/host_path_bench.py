"""End-to-end rate of the host-buffer operator (PCIe-inclusive): krep_gpu_literal_search on a host numpy buffer."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import krep_amd
from krep_amd import abi

gib = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
n = int(gib * (1 << 30))
e = krep_amd.load()
text = e.generate_host(1 << 20, 0, 2, 42, b"Sherlock", 10000)
text = np.tile(text, n // text.size)
p = abi.Params([b"Sherlock"], count_lines=True, only_match=True)
for rep in range(3):
    t0 = time.perf_counter()
    ret, _ = e.search(p, text, want_result=False)
    dt = time.perf_counter() - t0
    print(f"host operator: {text.size / 2**30:.1f} GiB, count={ret}, {dt * 1e3:.1f} ms, {text.size / dt / 1e9:.1f} GB/s (PCIe-inclusive)", flush=True)
