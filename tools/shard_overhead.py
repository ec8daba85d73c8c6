"""Multi-GPU preflight that needs no node: the HOST side of an 8-shard search, timed on one device.
(1) device-resident: a 32 GiB haystack cut into 8 contiguous shards of 4 GiB, each scanned by its own krep_gpu_scan_device call
    (ownership window + halo, as a rank of bench.py --gpus 8 does) and folded with krep_gpu_combine_line_counts — wall clock
    of the 8 calls against the sum of their kernel times: launch + synchronise + counter read-back + fold per shard;
(2) host-resident: search_buffer() over 8 logical shards (num_gpus = 8 on one physical device: eight worker threads, eight
    pinned staging rings, the parallel list merge, one all-reduce) against one shard, PCIe-inclusive.
usage: python tools/shard_overhead.py [GiB total = 32] [host GiB = 4]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import krep_amd
from krep_amd import abi

gib = float(sys.argv[1]) if len(sys.argv) > 1 else 32.0
hgib = float(sys.argv[2]) if len(sys.argv) > 2 else 4.0
n = int(gib * (1 << 30))
G = 8
e = krep_amd.load()
buf = torch.empty(n + 64, dtype=torch.uint8, device="cuda")
e.generate(buf.data_ptr(), n, 0, 2, 20260925, b"Sherlock", 10000)
shard = n // G
cap = shard // 8000 + 4096
pos = torch.empty(2 * cap, dtype=torch.int64, device="cuda")
print(f"# {gib:g} GiB resident, {G} shards of {shard / 2**30:g} GiB, one MI355X; wall clock of the {G} krep_gpu_scan_device calls vs the sum of their kernel times")
for label, kw, want in (("offsets", dict(), True), ("-c -o", dict(count_lines=True, only_match=True), False), ("-c (lines)", dict(count_lines=True), False)):
    plan = e.plan(abi.Params([b"Sherlock"], **kw))
    best = None
    for rep in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        outs = []
        for g in range(G):
            lo, hi = g * shard, (g + 1) * shard
            tl = min(n, hi + 64)
            outs.append(plan.scan(buf.data_ptr(), tl, lo, hi, 0, pos.data_ptr() if want else 0, cap if want else 0, time_it=True, global_len=n))
        arr = (abi.ScanOut * G)(*outs)
        lines = e.lib.krep_gpu_combine_line_counts(arr, G)
        total = sum(o.count for o in outs)
        wall = (time.perf_counter() - t0) * 1e3
        k = sum(o.kernel_ms for o in outs)
        if best is None or wall < best[0]:
            best = (wall, k)
    wall, k = best
    res = lines if kw.get("count_lines") and not kw.get("only_match") else total
    print(f"{label:12s} 8 shards: wall {wall:7.3f} ms, kernels {k:7.3f} ms, host side {wall - k:6.3f} ms = {(wall - k) / G * 1e3:6.1f} us per shard "
          f"({(wall - k) / k * 100:4.2f} % of the scans); result {res}", flush=True)
    plan.close()
del buf, pos
torch.cuda.empty_cache()
hn = int(hgib * (1 << 30))
text = e.generate_host(1 << 22, 0, 2, 42, b"Sherlock", 10000)
text = np.tile(text, hn // text.size)
print(f"# host-resident {text.size / 2**30:g} GiB through search_buffer() (PCIe-inclusive), logical shards on ONE device")
for label, kw, want in (("offsets", dict(), True), ("-c -o", dict(count_lines=True, only_match=True), False)):
    for g in (1, 2, 8):
        best = 1e9
        for rep in range(3):
            t0 = time.perf_counter()
            rc, cnt, p = e.search_buffer(abi.Params([b"Sherlock"], **kw), text, num_gpus=g, want_result=want)
            best = min(best, time.perf_counter() - t0)
        print(f"{label:8s} num_gpus={g}: {best * 1e3:8.1f} ms  {text.size / best / 1e9:6.1f} GB/s  count={cnt}", flush=True)
