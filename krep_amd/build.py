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
    return glob.glob(os.path.join(CSRC, "*.h")) + [os.path.join(HERE, "..", "include", "krep_gpu.h")]


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


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
