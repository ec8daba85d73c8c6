"""A/B of several builds of libkrep_gpu.so IN ONE PROCESS on the same HBM buffers (development aid).
Process-to-process variance of a 32 GiB scan is +-4 % on this part (placement), far above most kernel-level differences.
usage: python tools/ab_bench.py <gib> <kind: 2 literal8 | 3 memchr1 | 4 ac1000> <mode: pos|count|lines> <variant.so> [...]"""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from krep_amd import abi
from krep_amd.engine import Engine
import bench

gib, kind, mode = float(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
libs = sys.argv[4:]  # "<variant.so>" or "<variant.so>:ENV=VAL[,ENV=VAL]" (environment set around that variant's scans)
n = int(gib * (1 << 30))
def _eng(spec):
    path, _, envs = spec.partition(":")
    env = dict(kv.split("=", 1) for kv in envs.split(",") if kv)
    return (spec, Engine(path if os.path.isabs(path) else os.path.join(ROOT, "krep_amd", "lib", "variants", path)), env)
engs = [_eng(p) for p in libs]
engs = [(f"{name}#{i}", e, env) for i, (name, e, env) in enumerate(engs)]
wl = bench.workload({2: "literal8", 3: "memchr1", 4: "ac1000"}[kind])
buf = torch.empty(n + 64, dtype=torch.uint8, device="cuda")
envs = {name: env for name, _, env in engs}
engs = [(name, e) for name, e, _ in engs]
engs[0][1].generate(buf.data_ptr(), n, 0, wl["kind"], 42, wl["plant"], wl["period"])
if os.environ.get("AB_PATTERN"):
    wl["patterns"] = [os.environ["AB_PATTERN"].encode()]
kw = dict(count_lines=True, only_match=True) if mode == "count" else dict(count_lines=True) if mode == "lines" else {}
cap = (n // int(os.environ.get("AB_CAP_DIV", "50" if kind == 3 else "1500"))) + 4096 if mode == "pos" else 0  # (AB_CAP_DIV: denser patterns through AB_PATTERN)
pos = torch.empty(2 * cap, dtype=torch.int64, device="cuda") if cap else None
plans = []
for name, e in engs:  # the environment of a variant also holds while its plan (and tables) are built
    os.environ.update(envs[name])
    plans.append((name, e.plan(abi.Params(wl["patterns"], **kw))))
    for k in envs[name]:
        os.environ.pop(k, None)
times = {name: [] for name, _ in plans}
counts = {}
for rep in range(int(os.environ.get("AB_REPS", "9"))):
    for name, pl in plans:
        os.environ.update(envs[name])
        out = pl.scan(buf.data_ptr(), n, 0, n, 0, pos.data_ptr() if cap else 0, cap, time_it=True)
        for k in envs[name]:
            os.environ.pop(k, None)
        counts[name] = out.count
        if rep:
            times[name].append(out.kernel_ms)
for name, _ in plans:
    t = times[name]
    print(f"{name:40s} {mode:5s} kind={kind} median {statistics.median(t):7.3f} ms  min {min(t):7.3f}  {n / statistics.median(t) / 1e6:7.1f} GB/s   count={counts[name]}")
