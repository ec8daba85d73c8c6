"""Failure -> CPU fallback at the drop-in boundary (SURVEY §5 "Failure detection", §8b "Errors"; the reference's idiom
krep.c:1944-1948 and its error channel krep.c:2940-2947).  CPU suite: this container has no GPU, which IS the scenario.

A backend that answers 0 when it could not look turns "failed" into "no match" (`KREP_GPU=1 krep_gpu_cli -c
Sherlock file` printed file:0, rc 1, on a box without a device).  Pinned here:
  * without a usable device the selector returns NULL / the CLI switch stays off: output byte-identical to the CPU CLI;
  * an operator that fails at run time hands the search to the host's registered CPU function (status FELL_BACK), or
    reports status FAILED, appends nothing, and the patched CLI re-runs the chunk with the CPU pointer.
The same tests pass on a GPU box: the failures are forced there ($KREP_GPU_DISABLE, $KREP_GPU_INJECT_FAILURE)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import oracle_lib as ol
from krep_amd import abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "oracle", "_ref", "krep_gpu_cli")
needs_cli = pytest.mark.skipif(not os.path.exists(CLI), reason="oracle/_ref/krep_gpu_cli not built (needs /root/reference)")

_CLEAN = ("KREP_GPU", "KREP_GPU_DISABLE", "KREP_GPU_ASSUME_AVAILABLE", "KREP_GPU_INJECT_FAILURE", "KREP_GPU_NO_FALLBACK_HOOK",
          "KREP_GPU_MIN_BYTES", "KREP_GPU_DEVICE", "KREP_GPU_NUM", "KREP_GPU_COST_MODEL", "KREP_GPU_COST")


def run(args, **env):
    e = {k: v for k, v in os.environ.items() if k not in _CLEAN}
    e.update({k: str(v) for k, v in env.items()})
    r = subprocess.run([CLI] + args, env=e, capture_output=True, timeout=300)
    return r.returncode, r.stdout, r.stderr


@pytest.fixture(scope="module")
def files(tmp_path_factory):
    import krep_amd
    d = tmp_path_factory.mktemp("failover")
    big = krep_amd.load().generate_host(3 * (1 << 20) + 123, 0, 2, 11, b"Sherlock", 5000)
    big[1000:1008] = np.frombuffer(b"sherLOCK", dtype=np.uint8)
    f_big = d / "big.txt"
    f_big.write_bytes(big.tobytes())
    f_small = d / "small.txt"
    f_small.write_bytes(b"The quick brown fox\nSherlock Holmes and sherlock\nnothing here\nfoxSherlock fox\n")
    return f_big, f_small


def invocations(f_big, f_small):
    return [(["-c", "Sherlock"], f_big), (["-c", "-i", "sherlock"], f_big), (["-c", "-o", "Sherlock"], f_big),
            (["-c", "-e", "Sherlock", "-e", "the", "-e", "qz"], f_big), (["-c", "absent-pattern"], f_big),
            (["Sherlock"], f_small), (["-o", "fox"], f_small), (["-w", "fox"], f_small),
            (["-o", "-e", "fox", "-e", "Sherlock", "-e", "ox"], f_small), (["-c", "-m", "7", "Sherlock"], f_big),
            (["-o", "-i", "th"], f_big)]


@needs_cli
def test_cli_without_a_device_is_the_cpu_cli(files):
    """KREP_GPU=1 on a box without a usable device: krep_gpu_available() == 0, the switch stays off, every path is the
    reference's own — byte-identical stdout and exit code, nothing on stderr.  (The round-2 binary printed count 0, rc 1.)"""
    for args, path in invocations(*files):
        a = ["-t", "1", "--color=never"] + args + [str(path)]
        cpu = run(a)
        gpu = run(a, KREP_GPU=1, KREP_GPU_DISABLE=1)
        assert gpu[:2] == cpu[:2], (args, cpu[:2], gpu[:2])
        assert b"krep-gpu" not in gpu[2], gpu[2]
    # and the CPU path keeps its own chunking: same answer with the default thread count
    assert run(["-c", "-o", "Sherlock", str(files[0])], KREP_GPU=1, KREP_GPU_DISABLE=1)[:2] == \
        run(["-c", "-o", "Sherlock", str(files[0])])[:2]


@needs_cli
@pytest.mark.parametrize("hook", [True, False], ids=["registered_cpu_fallback", "rerun_in_search_chunk_thread"])
def test_cli_operator_failure_falls_back_to_the_cpu_function(files, hook):
    """The operator IS handed out (a device is assumed) and fails at run time (device allocation: here for real, on a GPU
    box injected).  hook=True: the backend calls the CLI's registered CPU selector itself (krep_gpu_set_cpu_fallback);
    hook=False: it returns status KREP_GPU_FAILED and the patched search_chunk_thread()/search_string() re-run the chunk
    with the CPU pointer (krep.c:1944-1948).  Either way: the CPU CLI's bytes and exit code."""
    env = dict(KREP_GPU=1, KREP_GPU_ASSUME_AVAILABLE=1, KREP_GPU_INJECT_FAILURE=1, KREP_GPU_MIN_BYTES=0, KREP_GPU_COST_MODEL=0)
    if not hook:
        env["KREP_GPU_NO_FALLBACK_HOOK"] = 1
    for args, path in invocations(*files):
        a = ["-t", "1", "--color=never"] + args + [str(path)]
        cpu = run(a)
        gpu = run(a, **env)
        assert gpu[:2] == cpu[:2], (args, cpu[:2], gpu[:2], gpu[2])
        assert b"krep-gpu:" in gpu[2]  # the failure is reported, not hidden
    s = ["-s", "--color=never", "Sherlock", "xx Sherlock yy Sherlock"]
    assert run(s, **env)[:2] == run(s)[:2]


@needs_cli
def test_cli_small_files_stay_on_the_cpu_function(files):
    """Size threshold (krep_gpu_worthwhile, default 1 MiB): a small file never reaches the backend — with a device that
    would fail every call, nothing fails because nothing is called."""
    env = dict(KREP_GPU=1, KREP_GPU_ASSUME_AVAILABLE=1, KREP_GPU_INJECT_FAILURE=1)
    a = ["-t", "1", "--color=never", "-o", "fox", str(files[1])]
    got = run(a, **env)
    assert got[:2] == run(a)[:2] and b"krep-gpu" not in got[2]


# ---- library level: the registered CPU selector, status codes, nothing half-written ----------------------------------
SELECT_T = C.CFUNCTYPE(C.c_void_p, C.POINTER(abi.SearchParams))


@pytest.fixture()
def failing_engine(monkeypatch):
    import krep_amd
    e = krep_amd.load()
    monkeypatch.setenv("KREP_GPU_ASSUME_AVAILABLE", "1")
    monkeypatch.delenv("KREP_GPU_DISABLE", raising=False)
    e.inject_failure(1)
    yield e
    e.inject_failure(0)
    e.set_cpu_fallback(None)
    e.set_reference_simd(abi.REF_AVX2)


def test_operator_failure_without_a_fallback_is_status_failed(failing_engine):
    import krep_amd
    e = failing_engine
    p = abi.Params([b"needle"])
    assert e.select(p) is not None  # a device is assumed, so the operator is handed out
    res = e.lib.krep_gpu_match_result_init(16)
    res.contents.count = 3  # the caller's list already holds three records
    text = b"hay needle hay needle"
    ret = e.lib.krep_gpu_literal_search(p.ref, C.c_char_p(text), len(text), res)
    assert ret == 0 and e.last_status() == abi.STATUS_FAILED and e.last_error()
    assert res.contents.count == 3  # nothing appended, nothing lost
    res.contents.count = 0
    e.lib.krep_gpu_match_result_free(res)
    with pytest.raises(krep_amd.KrepGpuError):
        e.search(p, text)
    rc, n, _ = e.search_buffer(p, text)
    assert rc == 2 and e.last_status() == abi.STATUS_FAILED


def test_registered_cpu_selector_answers_a_failed_operator(failing_engine):
    """The host registers its selector of CPU functions; here: the reference's own select_search_algorithm() from the
    compiled oracle/_ref build (or the restatement's functions where _ref is absent)."""
    e = failing_engine
    r = ol.ref(abi.REF_AVX2)
    o = ol.oracle()
    calls = []

    def select(pp):
        calls.append(1)
        if r is not None:
            return r.lib.select_search_algorithm(pp)
        par = pp.contents
        algo = abi.RA_AHO_CORASICK if par.num_patterns > 1 else o.lib.ko_select(pp, abi.REF_AVX2)
        return C.cast(getattr(o.lib, o.fn[algo]), C.c_void_p).value

    cb = SELECT_T(select)
    e.set_cpu_fallback(C.cast(cb, C.c_void_p).value)
    e.set_reference_simd(abi.REF_AVX2)
    text = e.generate_host(200_000, 0, 2, 5, b"Sherlock", 3000)
    chk = r if r is not None else o
    for pats, kw in (([b"Sherlock"], {}), ([b"e"], dict(count_lines=True)), ([b"the"], dict(case_sensitive=False)),
                     ([b"lock"], dict(max_count=5))):
        p = abi.Params(pats, **kw)
        algo = e.mirror_select(p, text.size)
        want = chk.call(algo, abi.Params(pats, **kw), text)
        got = e.search(p, text)
        assert e.last_status() == abi.STATUS_FELL_BACK
        assert got[0] == want[0] and np.array_equal(got[1], want[1]), (pats, kw)
    # several patterns: the CPU function needs the caller's real trie (krep.c:2528), and with result_order set the records
    # come back in the formatter's (start, end) order whoever produced them
    pats = [b"Sherlock", b"lock", b"he", b"her"]
    p = abi.Params(pats)
    trie = chk._acb(p.ref)
    p.s.ac_trie = trie
    res = e.lib.krep_gpu_match_result_init(16)
    e.set_result_order(True)
    try:
        ret = e.lib.krep_gpu_aho_corasick_search(p.ref, C.c_void_p(text.ctypes.data), text.size, res)
        got = abi.result_positions(res)
    finally:
        e.set_result_order(False)
        e.lib.krep_gpu_match_result_free(res)
        chk._acf(trie)
    want = chk.call(abi.RA_AHO_CORASICK, abi.Params(pats), text)
    assert e.last_status() == abi.STATUS_FELL_BACK and ret == want[0]
    order = np.lexsort((want[1][:, 1], want[1][:, 0]))
    assert np.array_equal(got, want[1][order])
    assert calls


def test_selector_returns_null_without_a_device(monkeypatch):
    import krep_amd
    e = krep_amd.load()
    monkeypatch.setenv("KREP_GPU_DISABLE", "1")
    p = abi.Params([b"Sherlock"])
    assert not e.available() and "KREP_GPU_DISABLE" in e.unavailable_reason()
    assert e.select(p) is None and not e.can_accelerate(p) and not e.worthwhile(p, 1 << 30)
    q = abi.Params([b"a", b"b"])
    assert e.select(q) is None
