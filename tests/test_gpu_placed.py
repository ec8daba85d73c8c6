"""krep_gpu_alloc_placed() (include/krep_gpu.h, krep_amd/csrc/kg_place.hip): one block for a text and its records, its physical placement
drawn for by timing BASELINE config 3's single-byte workload on up to k candidates.  The allocator has no counterpart in the reference
(krep maps files, krep.c:3325); what is checked is its contract — layout, what `info` reports, that the block scans like any other buffer
(same records as a torch allocation of the same bytes), that nothing is timed for small blocks, and that a failed allocation is an error."""
import numpy as np
import pytest

from krep_amd import abi
from krep_amd.engine import KrepGpuError

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    import krep_amd
    e = krep_amd.load()
    assert e.device_count() >= 1
    return e


def _scan(gpu, d_text, n, d_pos, cap, pat=b"#"):
    plan = gpu.plan(abi.Params([pat]))
    out = plan.scan(d_text, n, 0, n, 0, d_pos, cap)
    plan.close()
    return out


def test_small_block_is_one_plain_allocation(gpu):
    import torch
    n = (3 << 20) + 17
    d_text, d_rec, info = gpu.alloc_placed(n, 1 << 20, tries=4)
    try:
        assert info.tries == 1 and info.kept == 0 and info.records_ms[0] == 0.0
        assert d_rec == d_text + ((n + 64 + 255) & ~255) and d_rec % 256 == 0
        gpu.generate(d_text, n, 0, 3, 7, b"#", 0)
        host = gpu.generate_host(n, 0, 3, 7, b"#", 0)
        want = np.flatnonzero(host == ord("#"))
        out = _scan(gpu, d_text, n, d_rec, (1 << 20) // 16)
        assert out.count == len(want) and out.stored == len(want)

        class _Raw:
            __cuda_array_interface__ = {"shape": (2 * len(want),), "typestr": "<i8", "data": (d_rec, False), "version": 2}

        rec = torch.as_tensor(_Raw(), device="cuda").view(-1, 2).cpu().numpy()
        assert np.array_equal(rec[:, 0], want) and np.array_equal(rec[:, 1], want + 1)
    finally:
        gpu.free_placed(d_text)
    t2, r2, _ = gpu.alloc_placed(4096, 0, tries=1)
    assert r2 == 0
    gpu.free_placed(t2)


def test_large_block_is_drawn_for_and_scans_like_any_buffer(gpu):
    import torch
    n = (2 << 30) + 4096
    rec_bytes = n // 4
    d_text, d_rec, info = gpu.alloc_placed(n, rec_bytes, tries=3)
    try:
        assert 1 <= info.tries <= 3 and info.kept < info.tries
        for i in range(info.tries):
            assert 0.05 < info.count_only_ms[i] < 50 and info.count_only_ms[i] <= info.records_ms[i] < 100, (i, list(info.records_ms), list(info.count_only_ms))
        if info.tries > 1 and info.kept != info.tries - 1:  # (not the early accept: the fastest of what was drawn)
            assert info.records_ms[info.kept] == min(info.records_ms[: info.tries])
        # the caller's own bytes in the block: the same records as in a torch allocation
        gpu.generate(d_text, n, 0, 2, 11, b"Sherlock", 10000)
        other = torch.empty(n + 64, dtype=torch.uint8, device="cuda")
        gpu.generate(other.data_ptr(), n, 0, 2, 11, b"Sherlock", 10000)
        cap = rec_bytes // 16
        pos = torch.empty(2 * cap, dtype=torch.int64, device="cuda")
        a = _scan(gpu, d_text, n, d_rec, cap, b"Sherlock")
        b = _scan(gpu, other.data_ptr(), n, pos.data_ptr(), cap, b"Sherlock")
        assert a.count == b.count > 200000 and a.stored == b.stored

        class _Raw:
            __cuda_array_interface__ = {"shape": (2 * a.stored,), "typestr": "<i8", "data": (d_rec, False), "version": 2}

        assert torch.equal(torch.as_tensor(_Raw(), device="cuda"), pos[: 2 * b.stored])
    finally:
        gpu.free_placed(d_text)


def test_failed_allocation_is_an_error(gpu):
    gpu.inject_failure(1)
    try:
        with pytest.raises(KrepGpuError):
            gpu.alloc_placed(1 << 20, 1 << 16, tries=2)
    finally:
        gpu.inject_failure(0)
    with pytest.raises(KrepGpuError):
        gpu.alloc_placed(0, 0)
