"""GPU parity for the greedy non-overlapping family (simd_sse42_search / kmp_search) with BORDERED
patterns, where the all-occurrence set and the reference's set differ (test/test_krep.c:444-477)."""
import numpy as np
import pytest

import cases
import oracle_lib as ol
from krep_amd import abi

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    import krep_amd
    e = krep_amd.load()
    assert e.device_count() >= 1
    return e


def _check(gpu, o, text, pat, kw, level):
    gpu.set_reference_simd(level)
    p = abi.Params([pat], **kw)
    algo = gpu.mirror_select(p, text.size)
    want = o.call(algo, abi.Params([pat], **kw), text)
    got = gpu.search(p, text)
    assert got[0] == want[0], (abi.RA_NAMES[algo], pat, kw, text.size, got[0], want[0])
    assert np.array_equal(got[1], want[1]), (abi.RA_NAMES[algo], pat, kw, text.size, got[1][:8], want[1][:8])
    return algo


def test_reference_overlap_vectors(gpu, oracle_engine):
    # 'aba' in 'abababa': BM 3, KMP/SSE4.2 2; 'aa' in 'aaaaa': BM 4, KMP/SSE4.2 2 (test_krep.c:444-477)
    for text, pat, n_o, n_n in ((b"abababa", b"aba", 3, 2), (b"aaaaa", b"aa", 4, 2)):
        t = np.frombuffer(text, dtype=np.uint8)
        gpu.set_reference_simd(abi.REF_AVX2)
        assert gpu.search(abi.Params([pat]), t)[0] == n_n          # SSE4.2 path of an AVX2 build
        gpu.set_reference_simd(abi.REF_SCALAR)
        gpu.set_algo_override(abi.ALGO_BM)
        assert gpu.search(abi.Params([pat]), t)[0] == n_o
        gpu.set_algo_override(abi.ALGO_KMP)
        assert gpu.search(abi.Params([pat]), t)[0] == n_n
        gpu.set_algo_override(abi.ALGO_AUTO)


@pytest.mark.parametrize("seed", range(4))
def test_bordered_patterns(gpu, oracle_engine, seed):
    rng = np.random.RandomState(300 + seed)
    pats = [b"aa", b"aba", b"abab", b"aaaa", b"abaab", b"a-a", b"ab\nab", b"aabaa", b"abcabcab", b"a" * 16, b"ab" * 8]
    seen = set()
    for i in range(60):
        alpha = [b"ab", b"ab\n", b"ab-\n ", b"abc"][i % 4]
        n = [5, 40, 1000, 8192, 8200, 33000, 70000][rng.randint(0, 7)]
        text = cases.rand_text(rng, n, alpha)
        pat = pats[rng.randint(0, len(pats))]
        kw = dict(whole_word=bool(rng.rand() < 0.3), max_count=[abi.SIZE_MAX, abi.SIZE_MAX, 1, 3, 50][rng.randint(0, 5)])
        mode = ["pos", "lines", "count"][rng.randint(0, 3)]
        if mode == "lines":
            if b"\n" in pat:
                continue
            kw.update(count_lines=True)
        elif mode == "count":
            kw.update(count_lines=True, only_match=True)
        level = [abi.REF_AVX2, abi.REF_SSE42, abi.REF_AVX512][i % 3]
        seen.add(_check(gpu, oracle_engine, text, pat, kw, level))
    assert abi.RA_SSE42 in seen


def test_kmp_selection_and_extra_record(gpu, oracle_engine):
    """Scalar build: repetitive 4..7-byte patterns go to kmp_search (krep.c:1860-1865), which also stores
    the (max_count+1)-th match before breaking (krep.c:1717-1724)."""
    rng = np.random.RandomState(11)
    text = cases.rand_text(rng, 50000, b"ab")
    for pat in (b"abab", b"aaaa", b"ababab", b"aabaab"):
        for kw in (dict(), dict(max_count=5), dict(count_lines=True, only_match=True), dict(whole_word=True),
                   dict(case_sensitive=False, max_count=2)):
            algo = _check(gpu, oracle_engine, text, pat, kw, abi.REF_SCALAR)
            assert algo == abi.RA_KMP


def test_giant_cluster(gpu, oracle_engine):
    text = np.full(300_000, ord("a"), dtype=np.uint8)
    for pat in (b"aa", b"aaa", b"aaaaaaa"):
        _check(gpu, oracle_engine, text, pat, dict(), abi.REF_AVX2)
        _check(gpu, oracle_engine, text, pat, dict(max_count=1000), abi.REF_AVX2)


def test_pointer_jumping_form_of_the_walks(gpu, oracle_engine, monkeypatch):
    """A cluster longer than 4096 elements hands the pass to its parallel form (pointer jumping over nxt[i] = first element at
    or behind start_i + consume_i).  Forced here for ordinary inputs as well: greedy SSE4.2/KMP selection, BMH under -o and
    memchr_short_search's -o candidate walk must come out identical."""
    monkeypatch.setenv("KREP_GPU_FORCE_POINTER_JUMPING", "1")
    rng = np.random.RandomState(77)
    for n in (50, 4000, 70001):
        for alpha in (b"ab", b"ab\n", b"aab_ "):
            text = cases.rand_text(rng, n, alpha)
            for pat, kw, level in ((b"aa", dict(), abi.REF_AVX2), (b"abab", dict(whole_word=True), abi.REF_AVX2),
                                   (b"aba", dict(max_count=7), abi.REF_SSE42), (b"abab", dict(count_lines=True, whole_word=True), abi.REF_AVX2),
                                   (b"aaaa", dict(), abi.REF_SCALAR)):
                _check(gpu, oracle_engine, text, pat, kw, level)
            gpu.set_only_matching(True)
            oracle_engine.set_only_matching(True)
            try:
                for pat, kw, level in ((b"ab", dict(case_sensitive=False), abi.REF_AVX2), (b"aab", dict(case_sensitive=False, whole_word=True), abi.REF_AVX2),
                                       (b"aa", dict(), abi.REF_SCALAR), (b"abab", dict(case_sensitive=False), abi.REF_AVX2)):
                    _check(gpu, oracle_engine, text, pat, kw, level)
            finally:
                gpu.set_only_matching(False)
                oracle_engine.set_only_matching(False)
    monkeypatch.delenv("KREP_GPU_FORCE_POINTER_JUMPING")
    # and the natural trigger: one 300 000-element cluster, all three consumption rules
    text = np.full(300_000, ord("a"), dtype=np.uint8)
    text[123_456] = ord("b")
    _check(gpu, oracle_engine, text, b"aa", dict(), abi.REF_AVX2)
    gpu.set_only_matching(True)
    oracle_engine.set_only_matching(True)
    try:
        _check(gpu, oracle_engine, text, b"ab", dict(case_sensitive=False), abi.REF_AVX2)   # every candidate fails: m skipped
        _check(gpu, oracle_engine, text, b"aa", dict(case_sensitive=False, whole_word=True), abi.REF_AVX2)
    finally:
        gpu.set_only_matching(False)
        oracle_engine.set_only_matching(False)


def _count_on_device(gpu, o, text, pat, kw, level, om=False):
    """count-only through the device API (no record buffer: what kg_runs.hip takes) against the reference function's return value"""
    import torch
    gpu.set_reference_simd(level)
    p = abi.Params([pat], **kw)
    algo = gpu.mirror_select(p, text.size)
    want = o.call(algo, abi.Params([pat], **kw), text)[0]
    buf = torch.from_numpy(np.ascontiguousarray(text)).cuda()
    plan = gpu.plan(p, only_matching=om)
    got = plan.scan(buf.data_ptr(), text.size).count
    plan.close()
    assert got == want, (abi.RA_NAMES[algo], pat, kw, text.size, got, want)


def test_one_repeated_byte_counted_without_a_list(gpu, oracle_engine):
    """kg_runs.hip (round 6): a pattern of m copies of one byte, count-only, through the greedy families — floor(R / m) kept matches per
    maximal run, found from run lengths carried across lanes, cells and units (no list of all occurrences).  Against simd_sse42_search,
    kmp_search and boyer_moore_search under -o of the compiled reference / the restatement: runs on every seam (16-byte lane, 1-KiB
    cell, 8-KiB round, 32-KiB unit), a run longer than the look-back (the two-level form takes over), a text of nothing but the byte,
    windows chained with krep_gpu_scan_device_seq (the boundary record is where the reference's scan stands), -i, patterns longer than
    a lane."""
    import torch
    rng = np.random.RandomState(2026)
    kw = dict(count_lines=True, only_match=True)
    launches = gpu.runs_launches()
    for n in (7, 1000, 16 * 1024 + 3, 3 * 32768 + 77, 300_001):
        for alpha, pats in ((b"a b", [b"aa", b"aaa", b"  ", b"aaaaaaa"]), (b"-=x\n", [b"--", b"==", b"----", b"=" * 16]), (b"a", [b"aa", b"aaaaa"])):
            text = cases.rand_text(rng, n, alpha)
            for s, ln in ((0, 40), (16 - 3, 9), (1024 - 5, 30), (8192 - 7, 20), (32768 - 9, 33), (2 * 32768 - 1, 3), (n - 25, 25), (40000, 70000)):
                if 0 <= s and s + ln <= n:
                    text[s:s + ln] = pats[0][0]
            for pat in pats:
                for level in (abi.REF_AVX2, abi.REF_SCALAR):
                    if level == abi.REF_SCALAR:
                        gpu.set_algo_override(abi.ALGO_KMP)
                    try:
                        _count_on_device(gpu, oracle_engine, text, pat, kw, level)
                        _count_on_device(gpu, oracle_engine, text, pat, dict(max_count=3, **kw), level)
                        _check(gpu, oracle_engine, text, pat, kw, level)  # (the host operator with records: the list road)
                    finally:
                        gpu.set_algo_override(abi.ALGO_AUTO)
    # patterns longer than a 16-byte lane (kmp_search: any length), -i: a letter in either case is the same byte of the run
    text = cases.rand_text(rng, 120_000, b"aAb ")
    text[5000:5400] = ord("a")
    gpu.set_algo_override(abi.ALGO_KMP)
    try:
        for pat, kk in ((b"aaa", dict(case_sensitive=False)), (b"aA", dict(case_sensitive=False)), (b"a" * 40, dict(case_sensitive=False)), (b"a" * 17, dict())):
            _count_on_device(gpu, oracle_engine, text, pat, dict(**kk, **kw), abi.REF_SCALAR)
    finally:
        gpu.set_algo_override(abi.ALGO_AUTO)
    # boyer_moore_search under -o (greedy as well, krep.c:1371)
    gpu.set_only_matching(True)
    oracle_engine.set_only_matching(True)
    gpu.set_algo_override(abi.ALGO_BM)
    try:
        text = cases.rand_text(rng, 100_000, b"a b")
        _count_on_device(gpu, oracle_engine, text, b"aa", kw, abi.REF_SCALAR, om=True)
        _count_on_device(gpu, oracle_engine, text, b"   ", kw, abi.REF_SCALAR, om=True)
    finally:
        gpu.set_algo_override(abi.ALGO_AUTO)
        gpu.set_only_matching(False)
        oracle_engine.set_only_matching(False)
    # pieces in text order: the greedy phase continues across a cut through the boundary record
    gpu.set_reference_simd(abi.REF_AVX2)
    n = 200_000
    text = cases.rand_text(rng, n, b"a b")
    text[50_000 - 30:50_000 + 45] = ord("a")
    want = oracle_engine.call(abi.RA_SSE42, abi.Params([b"aaa"], **kw), text)[0]
    buf = torch.from_numpy(np.ascontiguousarray(text)).cuda()
    plan = gpu.plan(abi.Params([b"aaa"], **kw))
    assert plan.scan(buf.data_ptr(), n).count == want
    for cuts in ((0, 50_000, n), (0, 49_999, 50_001, 50_016, 123_457, n), (0, 16, 32768, n)):
        carry, total = None, 0
        for lo, hi in zip(cuts[:-1], cuts[1:]):
            o, carry = plan.scan_seq(buf.data_ptr(), n, lo, hi, carry_in=carry)
            total += o.total_matches
        assert total == want, (cuts, total, want)
    plan.close()
    assert gpu.runs_launches() > launches + 100  # the run-length kernel is what answered
