#!/usr/bin/env python3
"""Sanitizer pass over the HOST side of the library and over the parity checker (CPU only; SURVEY.md §5 asked for one, VERDICT r05
weak #11).  GPU AddressSanitizer is not available on this pool: the device code is compiled as always (-fno-gpu-sanitize), the host
code of every translation unit — worker threads, realloc growth, pinned ring, chain fix-up, communicator cache, table builders —
with the sanitizer named.

    python tools/sanitize.py asan      # libkrep_gpu_asan.so (ASan + UBSan) under the CPU suite's host-logic tests
    python tools/sanitize.py tsan      # libkrep_gpu_tsan.so under the multi-shard failover tests (8 logical shards, injected failures)
    python tools/sanitize.py oracle    # oracle/krep_oracle.c under ASan + UBSan against the golden vectors and the compiled reference
    python tools/sanitize.py reference # the reference's own krep_test built with ASan + UBSan (findings there are the reference's)

Each mode prints the sanitizer reports it saw (none expected) and exits non-zero on one; profiles/r06_sanitizers.txt is its log."""
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CLANG_RT = sorted(glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux"))[-1]
SAN = os.path.join(ROOT, "oracle", "_san")  # git-ignored scratch, never shipped


def run(cmd, env=None, **kw):
    print("+", " ".join(cmd), flush=True)
    e = dict(os.environ)
    e.update(env or {})
    return subprocess.run(cmd, env=e, **kw)


def lib_variant(name, flags):
    from krep_amd import build
    return build.build_variant(name, flags)


def pytest_under(lib, preload, tests, extra_env=None):
    env = {"LD_PRELOAD": preload, "ASAN_OPTIONS": "detect_leaks=0:abort_on_error=0:halt_on_error=0",
           "UBSAN_OPTIONS": "print_stacktrace=1", "TSAN_OPTIONS": "report_signal_unsafe=0:second_deadlock_stack=1",
           "KREP_GPU_NO_TORCH": "1"}
    if lib:
        env["KREP_GPU_LIB"] = lib
    env.update(extra_env or {})
    r = run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "not gpu", "-p", "no:cacheprovider"] + tests, env=env,
            capture_output=True, text=True, cwd=ROOT)
    out = r.stdout + r.stderr
    print(out[-3000:])
    bad = [l for l in out.splitlines() if "ERROR: AddressSanitizer" in l or "runtime error:" in l or "WARNING: ThreadSanitizer" in l]
    return r.returncode, bad


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "asan"
    os.makedirs(SAN, exist_ok=True)
    if mode == "asan":
        lib = lib_variant("asan", ["-O1", "-g", "-fsanitize=address,undefined", "-fno-gpu-sanitize", "-fno-sanitize=vptr,function",
                                   "-shared-libsan"])
        rc, bad = pytest_under(lib, os.path.join(CLANG_RT, "libclang_rt.asan-x86_64.so"),
                               ["tests/test_failover.py", "tests/test_replay_cpu.py", "tests/test_wordtext_cpu.py", "tests/test_golden_vectors.py"])
    elif mode == "tsan":
        lib = lib_variant("tsan", ["-O1", "-g", "-fsanitize=thread", "-fno-gpu-sanitize", "-shared-libsan"])
        rc, bad = pytest_under(lib, os.path.join(CLANG_RT, "libclang_rt.tsan-x86_64.so"), ["tests/test_failover.py"])
    elif mode == "oracle":
        so = os.path.join(SAN, "liboracle_krep_asan.so")
        r = run(["gcc", "-O1", "-g", "-std=c11", "-D_GNU_SOURCE", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-fPIC", "-shared",
                 "-pthread", "-o", so, os.path.join(ROOT, "oracle", "krep_oracle.c")])
        if r.returncode:
            return r.returncode
        pre = subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()
        rc, bad = pytest_under("", pre, ["tests/test_oracle_kat.py", "tests/test_golden_vectors.py", "tests/test_oracle_vs_ref.py"],
                               {"KREP_ORACLE_LIB": so})
    elif mode == "reference":
        ref = "/root/reference"
        if not os.path.exists(os.path.join(ref, "krep.c")):
            print("no /root/reference here")
            return 0
        exe = os.path.join(SAN, "krep_test_asan")
        r = run(["gcc", "-O1", "-g", "-std=c11", "-D_GNU_SOURCE", "-D_DEFAULT_SOURCE", "-pthread", "-w", "-DTESTING", "-mavx2", "-msse4.2",
                 "-DKREP_USE_SSE42=1", "-DKREP_USE_AVX2=1", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", f"-I{ref}", "-o", exe,
                 f"{ref}/krep.c", f"{ref}/aho_corasick.c", f"{ref}/test/test_krep.c", f"{ref}/test/test_regex.c",
                 f"{ref}/test/test_multiple_patterns.c", "-lm"])
        if r.returncode:
            return r.returncode
        r = run([exe], env={"ASAN_OPTIONS": "detect_leaks=0:halt_on_error=0", "UBSAN_OPTIONS": "print_stacktrace=0"}, capture_output=True, text=True,
                cwd=SAN)
        out = r.stdout + r.stderr
        bad = [l for l in out.splitlines() if "ERROR: AddressSanitizer" in l or "runtime error:" in l]
        print(out[-1500:])
        rc = r.returncode
    else:
        print(__doc__)
        return 2
    print(f"== sanitize {mode}: exit code {rc}, {len(bad)} sanitizer report line(s)")
    for l in bad[:40]:
        print("   ", l)
    return 1 if (rc or bad) else 0


if __name__ == "__main__":
    sys.exit(main())
