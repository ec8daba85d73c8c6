"""Is the 10 % per-allocation mode of the record-writing single-byte scan (profiles/r04_placement.txt) a property of HOW the record buffer
is allocated?  ("compare hipMalloc with hipExtMallocWithFlags / fine-grain variants".)  One process, the 32 GiB text
stays; the record buffer is drawn several times from each allocator and the scan (34.4 GB read + 5.5 GB of records) is timed on each draw.
usage: python tools/alloc_flavour_probe.py [GiB = 32] [draws = 5]"""
import ctypes as C, os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from krep_amd import abi
from krep_amd.engine import Engine
import bench

gib = float(sys.argv[1]) if len(sys.argv) > 1 else 32.0
draws = int(sys.argv[2]) if len(sys.argv) > 2 else 5
n = int(gib * (1 << 30))
e = Engine()
hip = C.CDLL("libamdhip64.so")
wl = bench.workload("memchr1")
cap = n // 80 + 4096
nbytes = 16 * cap
buf = torch.empty(n + 64, dtype=torch.uint8, device="cuda")
e.generate(buf.data_ptr(), n, 0, wl["kind"], bench.SEED, wl["plant"], wl["period"])
torch.cuda.synchronize()
plan = e.plan(abi.Params(wl["patterns"]))
cnt = e.plan(abi.Params(wl["patterns"], count_lines=True, only_match=True))
base = min(cnt.scan(buf.data_ptr(), n, time_it=True).kernel_ms for _ in range(4))
print(f"# {gib:g} GiB, count-only scan {base:.3f} ms; record scan per draw (median of 5 launches, ms) — fast mode ~1.28x, slow mode ~1.41x of that", flush=True)


def timed(ptr):
    ts = [plan.scan(buf.data_ptr(), n, 0, n, 0, ptr, cap, time_it=True).kernel_ms for _ in range(6)][1:]
    return statistics.median(ts)


def flavour(name, alloc, free):
    row = []
    held = []
    for d in range(draws):
        p = alloc()
        if not p:
            row.append("alloc failed")
            break
        al = (p & -p).bit_length() - 1  # log2 of the pointer's alignment
        row.append(f"{timed(p):.3f} (2^{min(al, 40)})")
        held.append(p)  # keep it: the next draw lands elsewhere
    for p in held:
        free(p)
    print(f"{name:58s} " + "  ".join(row), flush=True)


def hip_malloc():
    p = C.c_void_p()
    return p.value if hip.hipMalloc(C.byref(p), C.c_size_t(nbytes)) == 0 else None


def ext(flags):
    def f():
        p = C.c_void_p()
        return p.value if hip.hipExtMallocWithFlags(C.byref(p), C.c_size_t(nbytes), C.c_uint(flags)) == 0 else None
    return f


def hfree(p):
    hip.hipFree(C.c_void_p(p))


tens = []
def torch_alloc():
    t = torch.empty(2 * cap, dtype=torch.int64, device="cuda")
    tens.append(t)
    return t.data_ptr()

def sized(sz):
    def f():
        p = C.c_void_p()
        return p.value if hip.hipMalloc(C.byref(p), C.c_size_t(sz)) == 0 else None
    return f

def up(x, a):
    return (x + a - 1) // a * a

print(f"# record buffer: {nbytes} bytes = {nbytes / 2**20:.3f} MiB", flush=True)
flavour("hipMalloc(exact size)", sized(nbytes), hfree)
flavour("hipMalloc(size rounded up to 2 MiB)", sized(up(nbytes, 2 << 20)), hfree)
flavour("hipMalloc(size rounded up to 64 MiB)", sized(up(nbytes, 64 << 20)), hfree)
flavour("hipMalloc(size rounded up to 1 GiB)", sized(up(nbytes, 1 << 30)), hfree)
flavour("hipMalloc(exact size + 4 KiB)", sized(nbytes + 4096), hfree)
flavour("hipMalloc(exact size) again", sized(nbytes), hfree)
flavour("torch caching allocator (hipMalloc underneath)", torch_alloc, lambda p: None)
tens.clear(); torch.cuda.empty_cache()
flavour("hipMalloc", hip_malloc, hfree)
flavour("hipExtMallocWithFlags(hipDeviceMallocDefault = 0)", ext(0), hfree)
flavour("hipExtMallocWithFlags(hipDeviceMallocFinegrained = 1)", ext(1), hfree)
flavour("hipExtMallocWithFlags(hipDeviceMallocUncached = 3)", ext(3), hfree)
flavour("hipExtMallocWithFlags(hipDeviceMallocContiguous = 4)", ext(4), hfree)
# 2 MiB-aligned sub-buffer of one large hipMalloc at different offsets (same allocation, different physical pages)
big = hip_malloc_big = None
p = C.c_void_p()
if hip.hipMalloc(C.byref(p), C.c_size_t(4 * nbytes)) == 0:
    row = [f"{timed(p.value + k * nbytes):.3f}" for k in range(4)]
    print(f"{'one hipMalloc of 4x the size, quarter k as the record buffer':58s} " + "  ".join(row), flush=True)
    hip.hipFree(p)
