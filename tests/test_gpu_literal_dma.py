"""GPU parity of the LDS-DMA literal kernel (krep_amd/csrc/kg_literal_dma.hip: 2..8-byte patterns with a rare first byte, 32-KiB units,
no -c) against the compiled reference — the functions the mirror selector names (simd_sse42_search krep.c:4702, boyer_moore_search
:1260, memchr_short_search :4371) on the same inputs.  The kernel is taken from ~24 GiB on (tickets of 8 units); here small texts
reach it through the test switches: krep_gpu_debug_force_rounds(4) (32-KiB units), $KREP_GPU_LIT_UPT (ticket size; 0 = the static
deal) and $KREP_GPU_LIT_DMA_ALL.  Cases sit on its seams: the positions whose window crosses a round (deferred to the next round's
first bytes), a ticket's end (the 256-byte DMA piece), the text's ragged end (guarded loads), ownership windows, -w, -i, max_count,
overflowing staging slots (the emit-mode re-scan runs in the register kernel on this kernel's info words)."""
import os

import numpy as np
import pytest

import cases
from krep_amd import abi

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    import krep_amd
    e = krep_amd.load()
    assert e.device_count() >= 1
    return e


@pytest.fixture(params=["0", "1", "4", "8"])
def dma(gpu, request):
    os.environ["KREP_GPU_LIT_DMA_ALL"] = "1"
    os.environ["KREP_GPU_LIT_UPT"] = request.param
    gpu.force_rounds(4)
    yield request.param
    gpu.force_rounds(0)
    os.environ.pop("KREP_GPU_LIT_DMA_ALL", None)
    os.environ.pop("KREP_GPU_LIT_UPT", None)


def _check(gpu, o, text, pat, kw, level=abi.REF_AVX2):
    gpu.set_reference_simd(level)
    try:
        p = abi.Params([pat], **kw)
        algo = gpu.mirror_select(p, len(text))
        want = o.call(algo, abi.Params([pat], **kw), text)
        got = gpu.search(p, text)
    finally:
        gpu.set_reference_simd(abi.REF_AVX2)
    assert got[0] == want[0], (pat, kw, len(text), got[0], want[0])
    assert np.array_equal(got[1], want[1]), (pat, kw, got[1][:6], want[1][:6])


def test_plants_on_every_seam(gpu, oracle_engine, dma):
    rng = np.random.RandomState(1 + int(dma))
    before = gpu.literal_dma_launches()
    for n in (4 * 8192 + 5, 8 * 32768, 8 * 32768 + 8191, 3 * 8 * 32768 + 12345, 40 * 32768 - 1):
        for pat in (b"Sherlock", b"Qx", b"Zeb", b"WXYZ", b"Kappa", b"Jacket7", b"#include"[:8]):
            m = len(pat)
            text = cases.rand_text(rng, n, bytes(range(97, 123)) + b"  \n")
            p = np.frombuffer(pat, dtype=np.uint8)
            spots = [0, 1, 8192 - m, 8192 - m + 1, 8192 - 1, 8192, 16384 - 3, 32768 - 1, 32768 - m + 2, 4 * 32768 - 2, 8 * 32768 - 1, 8 * 32768 - m,
                     8 * 32768 - m + 1, 8 * 32768, n - m, n - m - 1, n - 8192 - 2, n // 2]
            for s in spots:
                if 0 <= s <= n - m:
                    text[s:s + m] = p
            for kw in (dict(), dict(count_lines=True, only_match=True), dict(whole_word=True), dict(max_count=5)):
                _check(gpu, oracle_engine, text, pat, kw)
    assert gpu.literal_dma_launches() > before


def test_case_insensitive_and_scalar_family(gpu, oracle_engine, dma):
    rng = np.random.RandomState(50 + int(dma))
    n = 9 * 32768 + 777
    before = gpu.literal_dma_launches()
    for pat in (b"qUiZ", b"Jump", b"xylo", b"Zq", b"VwXyZ12"):
        text = cases.rand_text(rng, n, b"abcdefghijklmnopqrstuvwxyzQJXZ  \n")
        m = len(pat)
        for s in (5, 8192 - 2, 32768 - 1, n - m, 100000):
            v = np.frombuffer(pat, dtype=np.uint8).copy()
            letter = ((v | 0x20) >= 97) & ((v | 0x20) <= 122)
            v = np.where((rng.rand(m) < 0.5) & letter, v ^ 0x20, v).astype(np.uint8)  # some letters in the other case
            text[s:s + m] = v
        for level in (abi.REF_AVX2, abi.REF_SCALAR):
            _check(gpu, oracle_engine, text, pat, dict(case_sensitive=False), level)
            _check(gpu, oracle_engine, text, pat, dict(case_sensitive=False, whole_word=True), level)
            _check(gpu, oracle_engine, text, pat, dict(), level)
    assert gpu.literal_dma_launches() > before


def test_windows_and_overflowing_slots(gpu, oracle_engine, dma):
    """Device windows (start ownership) cut inside rounds, at round and ticket ends; dense plants overflow the 16-entry slots."""
    import torch
    rng = np.random.RandomState(7 + int(dma))
    n = 20 * 32768 + 4321
    pat = b"Xy7"
    text = cases.rand_text(rng, n, b"abcXy7 \n")
    host = np.ascontiguousarray(text)
    want = oracle_engine.call(abi.RA_BMH, abi.Params([pat]), host)[1].astype(np.int64)
    buf = torch.from_numpy(host).cuda()
    cap = len(want) + 4096
    pos = torch.empty(2 * cap, dtype=torch.int64, device="cuda")
    gpu.set_algo_override(abi.ALGO_BM)
    try:
        plan = gpu.plan(abi.Params([pat]))
        base = (3 << 32) + 17
        for lo, hi in ((0, n), (8191, 8193), (8192, 3 * 8192), (32768 - 1, 8 * 32768 + 1), (100, n - 100), (8 * 32768 - 2, 8 * 32768 + 2), (n - 9000, n)):
            out = plan.scan(buf.data_ptr(), n, lo, hi, base, pos.data_ptr(), cap, global_len=base + n)
            sel = want[(want[:, 0] >= lo) & (want[:, 0] < hi)] + base
            assert out.count == len(sel) and np.array_equal(pos[: 2 * out.stored].view(-1, 2).cpu().numpy(), sel), (lo, hi)
        plan.close()
    finally:
        gpu.set_algo_override(abi.ALGO_AUTO)


def test_kernel_choice_follows_the_first_byte_in_the_text(gpu, oracle_engine, dma):
    """The kernel is chosen by what the TEXT holds (kg_scan.hip lit_pass): a sample before the plan's first eligible launch, and the
    kernel's own count of the 1-KiB cells its prefilter let through.  Text A does not hold the pattern's first byte outside the plants;
    in text B it is 1 byte in 100 (every cell holds it).  Same records as the compiled reference whichever kernel runs; a plan barred
    by B looks again when it is handed A."""
    import torch
    rng = np.random.RandomState(90 + int(dma))
    n = (72 << 20) + 4321
    pat = b"Qwerty"
    m = len(pat)
    A = cases.rand_text(rng, n, bytes(range(97, 123)) + b"  \n")
    for s in list(rng.randint(0, n - m, 300)) + [0, n - m, 8 * 32768 - 3]:
        A[s:s + m] = np.frombuffer(pat, dtype=np.uint8)
    B = A.copy()
    B[rng.rand(n) < 0.01] = ord("Q")
    algo = gpu.mirror_select(abi.Params([pat]), n)
    want = {id(t): oracle_engine.call(algo, abi.Params([pat]), t)[1].astype(np.int64) for t in (A, B)}
    dA, dB = torch.from_numpy(A).cuda(), torch.from_numpy(B).cuda()
    cap = 4096
    pos = torch.empty(2 * cap, dtype=torch.int64, device="cuda")

    def scan(plan, d, host):
        before = gpu.literal_dma_launches()
        out = plan.scan(d.data_ptr(), n, 0, n, 0, pos.data_ptr(), cap)
        assert np.array_equal(pos[: 2 * out.stored].view(-1, 2).cpu().numpy(), want[id(host)])
        return gpu.literal_dma_launches() - before

    plan = gpu.plan(abi.Params([pat]))
    assert scan(plan, dA, A) == 1
    looked, barred, rate = plan.literal_dma_state()
    assert looked and not barred and rate < 0.01, (looked, barred, rate)
    assert scan(plan, dB, B) == 1          # not barred yet: this launch is the one that counts B's cells
    looked, barred, rate = plan.literal_dma_state()
    assert barred and rate > 0.9, (barred, rate)
    assert scan(plan, dB, B) == 0          # the register kernel
    assert scan(plan, dA, A) == 1          # another text: sampled again, the bar is lifted
    assert not plan.literal_dma_state()[1]
    plan.close()
    plan = gpu.plan(abi.Params([pat]))     # a fresh plan on B: the sample bars the kernel before its first launch
    assert scan(plan, dB, B) == 0
    assert plan.literal_dma_state()[:2] == (True, True)
    os.environ["KREP_GPU_LIT_DMA_KEEP"] = "1"
    try:
        assert scan(plan, dB, B) == 1      # (the measurement aid: no sample, no bar; same records)
    finally:
        os.environ.pop("KREP_GPU_LIT_DMA_KEEP", None)
    plan.close()
