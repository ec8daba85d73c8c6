"""Failure -> CPU fallback on a box WITH a device: every point at which a host-buffer operator can fail at run time
(device allocation, host->device copy, kernel launch, device->host copy of the records) is forced through the
library's injection hook, with and without a registered CPU selector, through the C-ABI and through the patched
reference CLI.  A failed call appends nothing, the next call works again.  (CPU-box twin: tests/test_failover.py.)"""
import ctypes as C
import os

import numpy as np
import pytest

import oracle_lib as ol
from krep_amd import abi
from test_failover import CLI, SELECT_T, invocations, needs_cli, run

pytestmark = pytest.mark.gpu

KINDS = {1: "device allocation", 2: "host->device copy", 3: "kernel launch", 4: "device->host copy"}


@pytest.fixture()
def gpu():
    import krep_amd
    e = krep_amd.load()
    assert e.device_count() >= 1 and e.available(), e.unavailable_reason()
    e.set_reference_simd(abi.REF_AVX2)
    yield e
    e.inject_failure(0)
    e.set_cpu_fallback(None)
    e.release_device_resources()


def _cpu_selector():
    r = ol.ref(abi.REF_AVX2)
    o = ol.oracle()

    def select(pp):
        if r is not None:
            return r.lib.select_search_algorithm(pp)
        algo = abi.RA_AHO_CORASICK if pp.contents.num_patterns > 1 else o.lib.ko_select(pp, abi.REF_AVX2)
        return C.cast(getattr(o.lib, o.fn[algo]), C.c_void_p).value

    return SELECT_T(select), (r if r is not None else o)


@pytest.mark.parametrize("kind", sorted(KINDS))
def test_every_failure_point_without_fallback_is_status_failed_and_recovers(gpu, kind):
    import krep_amd
    text = gpu.generate_host(6 * (1 << 20) + 5, 0, 2, 3, b"Sherlock", 4000)
    p = abi.Params([b"Sherlock"])
    good = gpu.search(p, text)
    assert gpu.last_status() == abi.STATUS_OK and good[0] > 1000
    gpu.release_device_resources()  # so that allocation and plan creation happen again under the injection
    gpu.inject_failure(kind)
    res = gpu.lib.krep_gpu_match_result_init(16)
    ret = gpu.lib.krep_gpu_literal_search(p.ref, C.c_void_p(text.ctypes.data), text.size, res)
    assert ret == 0 and gpu.last_status() == abi.STATUS_FAILED and res.contents.count == 0, KINDS[kind]
    assert "injected" in gpu.last_error() or "failed" in gpu.last_error()
    gpu.lib.krep_gpu_match_result_free(res)
    with pytest.raises(krep_amd.KrepGpuError):
        gpu.search(p, text)
    rc, n, _ = gpu.search_buffer(p, text)
    assert rc == 2
    # streamed pieces take the same exits
    gpu.set_stream_chunk(1 << 20)
    try:
        rc, n, _ = gpu.search_buffer(p, text)
        assert rc == 2 and gpu.last_status() == abi.STATUS_FAILED
    finally:
        gpu.set_stream_chunk(0)
    gpu.inject_failure(0)
    again = gpu.search(p, text)
    assert gpu.last_status() == abi.STATUS_OK and again[0] == good[0] and np.array_equal(again[1], good[1])


@pytest.mark.parametrize("kind", sorted(KINDS))
def test_every_failure_point_with_registered_cpu_selector(gpu, kind):
    cb, chk = _cpu_selector()
    gpu.set_cpu_fallback(C.cast(cb, C.c_void_p).value)
    text = gpu.generate_host(3 * (1 << 20) + 77, 0, 2, 9, b"Sherlock", 3000)
    gpu.release_device_resources()
    gpu.inject_failure(kind)
    for pats, kw in (([b"Sherlock"], {}), ([b"the"], dict(case_sensitive=False)), ([b"e"], dict(count_lines=True))):
        if kind == 4 and kw.get("count_lines"):
            continue  # -c copies no records back: nothing to fail
        p = abi.Params(pats, **kw)
        want = chk.call(gpu.mirror_select(p, text.size), abi.Params(pats, **kw), text)
        got = gpu.search(p, text)
        assert gpu.last_status() == abi.STATUS_FELL_BACK, (KINDS[kind], pats)
        assert got[0] == want[0] and np.array_equal(got[1], want[1]), (KINDS[kind], pats, kw)
    gpu.inject_failure(0)
    p = abi.Params([b"Sherlock"])
    got = gpu.search(p, text)
    assert gpu.last_status() == abi.STATUS_OK
    want = chk.call(gpu.mirror_select(p, text.size), abi.Params([b"Sherlock"]), text)
    assert got[0] == want[0] and np.array_equal(got[1], want[1])


def test_device_out_of_range_is_not_available(gpu):
    cfg = gpu.default_config()
    cfg.device = 99
    gpu.set_thread_config(cfg)
    try:
        assert not gpu.available() and "out of range" in gpu.unavailable_reason()
        assert gpu.select(abi.Params([b"Sherlock"])) is None
    finally:
        gpu.set_thread_config(None)
    assert gpu.available() and gpu.select(abi.Params([b"Sherlock"])) is not None


@needs_cli
def test_cli_with_an_unusable_device_or_failing_operator(tmp_path):
    import krep_amd
    big = krep_amd.load().generate_host(3 * (1 << 20) + 123, 0, 2, 11, b"Sherlock", 5000)
    f_big = tmp_path / "big.txt"
    f_big.write_bytes(big.tobytes())
    f_small = tmp_path / "small.txt"
    f_small.write_bytes(b"The quick brown fox\nSherlock Holmes and sherlock\nnothing here\nfoxSherlock fox\n")
    for args, path in invocations(f_big, f_small):
        a = ["-t", "1", "--color=never"] + args + [str(path)]
        cpu = run(a)
        ok = run(a, KREP_GPU=1, KREP_GPU_MIN_BYTES=0, KREP_GPU_COST_MODEL=0)
        assert ok[:2] == cpu[:2] and b"krep-gpu" not in ok[2], (args, ok)
        bad_dev = run(a, KREP_GPU=1, KREP_GPU_DEVICE=99, KREP_GPU_MIN_BYTES=0, KREP_GPU_COST_MODEL=0)
        assert bad_dev[:2] == cpu[:2] and b"krep-gpu" not in bad_dev[2], (args, bad_dev)
        for kind in KINDS:
            for extra in ({}, {"KREP_GPU_NO_FALLBACK_HOOK": 1}):
                got = run(a, KREP_GPU=1, KREP_GPU_INJECT_FAILURE=kind, KREP_GPU_MIN_BYTES=0, KREP_GPU_COST_MODEL=0, **extra)
                assert got[:2] == cpu[:2], (args, KINDS[kind], extra, cpu[:2], got)
