"""The device-resident scan WITHOUT PyTorch in the process (development probe): buffers from hipMalloc of the SYSTEM HIP runtime
(/opt/rocm), the library bound to that runtime.  bench.py and the tests run with PyTorch's bundled HIP runtime (an older ROCm);
this tells whether a number depends on which runtime launched the kernel / allocated the buffers.
usage: KREP_GPU_NO_TORCH=1 python tools/notorch_bench.py <gib> <workload: literal8|memchr1|ac1000>"""
import ctypes as C
import os
import statistics
import sys

os.environ["KREP_GPU_NO_TORCH"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
hip = C.CDLL("/opt/rocm/lib/libamdhip64.so", mode=C.RTLD_GLOBAL)
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipDeviceSynchronize.restype = C.c_int
from krep_amd import abi  # noqa: E402
from krep_amd.engine import Engine  # noqa: E402

gib, name = float(sys.argv[1]), sys.argv[2]
import importlib.util  # noqa: E402
spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)
assert "torch" not in sys.modules, "torch must not be loaded for this probe"
e = Engine()
wl = bench.workload(name)
n = int(gib * (1 << 30))


def dmalloc(nbytes):
    p = C.c_void_p()
    rc = hip.hipMalloc(C.byref(p), nbytes)
    assert rc == 0 and p.value, rc
    return p.value


buf = dmalloc(n + 64)
e.generate(buf, n, 0, wl["kind"], bench.SEED, wl["plant"], wl["period"])
density = 1.0 / 100 if wl["kind"] == 3 else 1.0 / wl["period"] + (2.5e-4 if wl["kind"] == 4 else 0)
cap = int(n * density * 1.25) + 4096
pos = dmalloc(cap * 16)
plan = e.plan(abi.Params(wl["patterns"], **wl["kw"]))
ms = []
for rep in range(12):
    out = plan.scan(buf, n, 0, n, 0, pos, cap, time_it=True)
    if rep >= 2:
        ms.append(out.kernel_ms)
print(f"no-torch (system HIP runtime) {name}: kernel_ms median {statistics.median(ms):.4f} min {min(ms):.4f} count {out.count} "
      f"frac {n / statistics.median(ms) / 1e6 / 8000:.4f}")
