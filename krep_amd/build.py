"""Builds libkrep_gpu.so (hipcc, gfx950 only) in-tree: krep_amd/lib/libkrep_gpu.so.

Every csrc/*.hip is compiled to its own object (in parallel; only the stale ones) and linked into one shared library."""
from __future__ import annotations

import glob
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(LIBDIR, "obj")
LIB = os.path.join(LIBDIR, "libkrep_gpu.so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-cuda-compat"]


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _headers():
    return glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(HERE, "..", "include", "*.h"))


def _obj(src: str) -> str:
    return os.path.join(OBJDIR, os.path.basename(src)[:-4] + ".o")


def _stale_obj(src: str) -> bool:
    o = _obj(src)
    if not os.path.exists(o):
        return True
    t = os.path.getmtime(o)
    return any(os.path.getmtime(d) > t for d in [src] + _headers())


def stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(d) > t for d in sources() + _headers())


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not stale():
        return LIB
    os.makedirs(OBJDIR, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    todo = [s for s in sources() if force or _stale_obj(s)]

    def compile_one(src):
        cmd = [hipcc] + FLAGS + ["-c", src, "-o", _obj(src)]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.run(cmd, check=True, cwd=CSRC)

    with ThreadPoolExecutor(max_workers=min(8, max(1, len(todo)))) as ex:
        list(ex.map(compile_one, todo))
    # drop objects whose source is gone
    keep = {_obj(s) for s in sources()}
    for o in glob.glob(os.path.join(OBJDIR, "*.o")):
        if o not in keep:
            os.remove(o)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + sorted(keep)
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True, cwd=CSRC)
    return LIB


def build_variant(name: str, defines, verbose: bool = False) -> str:
    """Development aid: the same sources with extra -D switches, as krep_amd/lib/exp/libkrep_gpu_<name>.so — an A/B partner
    for one GPU session (KREP_GPU_LIB=<path> selects it in krep_amd.load(), tools/ab_bench.py).  Never the shipped library."""
    out_dir = os.path.join(LIBDIR, "exp")
    obj_dir = os.path.join(out_dir, "obj_" + name)
    os.makedirs(obj_dir, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []

    def compile_one(src):
        o = os.path.join(obj_dir, os.path.basename(src)[:-4] + ".o")
        cmd = [hipcc] + FLAGS + list(defines) + ["-c", src, "-o", o]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.run(cmd, check=True, cwd=CSRC)
        return o

    with ThreadPoolExecutor(max_workers=8) as ex:
        objs = list(ex.map(compile_one, sources()))
    lib = os.path.join(out_dir, f"libkrep_gpu_{name}.so")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + sorted(objs), check=True, cwd=CSRC)
    return lib


if __name__ == "__main__":
    if "--variant" in sys.argv:
        i = sys.argv.index("--variant")
        print(build_variant(sys.argv[i + 1], [a for a in sys.argv[i + 2:] if a.startswith("-D")], verbose=True))
    else:
        print(build(force="--force" in sys.argv, verbose=True))
