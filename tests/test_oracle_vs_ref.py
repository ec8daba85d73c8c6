"""Differential pin of the oracle restatement against the UNMODIFIED reference compiled into
oracle/_ref (count, result->count and every offset, in order).  Skipped when _ref is absent
(it is built from /root/reference by oracle/Makefile and travels to the GPU box prebuilt)."""
import random

import numpy as np
import pytest

import oracle_lib as ol
from krep_amd import abi

LEVELS = [abi.REF_SCALAR, abi.REF_SSE42, abi.REF_AVX2, abi.REF_AVX512, abi.REF_NEON]
ALPHAS = [b"ab", b"ab\n", b"abAB \n", b"abc_ \n-", bytes(range(97, 123)) + b" \n"]
SIZES = [0, 1, 2, 3, 5, 8, 15, 16, 17, 31, 32, 33, 40, 63, 64, 65, 70, 100, 127, 128, 129, 200, 300, 1000]


def _have_any():
    return any(ol.ref_available(l) for l in LEVELS)


pytestmark = pytest.mark.skipif(not _have_any(), reason="oracle/_ref not built")


def _case(rng):
    alpha = rng.choice(ALPHAS)
    n = rng.choice(SIZES)
    text = bytes(rng.choice(alpha) for _ in range(n))
    algo = rng.choice([abi.RA_BMH, abi.RA_KMP, abi.RA_MEMCHR, abi.RA_MEMCHR_SHORT, abi.RA_SSE42,
                       abi.RA_AVX2, abi.RA_AVX512, abi.RA_NEON, abi.RA_AHO_CORASICK])
    m = {abi.RA_MEMCHR: [1], abi.RA_MEMCHR_SHORT: [2, 3], abi.RA_AVX2: [2, 5, 16, 17, 18, 20, 32],
         abi.RA_AVX512: [3, 17, 33, 34, 40, 64], abi.RA_NEON: [1, 2, 3, 4, 5, 8, 9, 15, 16, 17, 20]}.get(algo, [1, 2, 3, 4, 5, 8, 9, 16])
    m = rng.choice(m)

    def mk(k):
        if n >= k and rng.random() < 0.7:
            s = rng.randrange(0, n - k + 1)
            return text[s:s + k]
        return bytes(rng.choice(alpha) for _ in range(k))

    kw = dict(case_sensitive=rng.random() < 0.6, whole_word=rng.random() < 0.3,
              max_count=rng.choice([abi.SIZE_MAX] * 3 + [0, 1, 2, 3, 5]))
    mode = rng.choice(["pos", "lines", "count"])
    if mode == "lines":
        kw.update(count_lines=True)
    elif mode == "count":
        kw.update(count_lines=True, only_match=True)
    if algo == abi.RA_AHO_CORASICK:
        pats = [mk(rng.choice([1, 2, 3, 4, 6])) for _ in range(rng.choice([2, 3, 5, 8]))]
    else:
        pats = [mk(m)]
    return algo, pats, kw, text


@pytest.mark.parametrize("seed", range(8))
def test_oracle_equals_reference(seed):
    rng = random.Random(1000 + seed)
    o = ol.oracle()
    refs = [ol.ref(l) for l in LEVELS if ol.ref_available(l)]
    checked = 0
    for _ in range(500):
        algo, pats, kw, text = _case(rng)
        for r in refs:
            if not r.has(algo):
                continue
            a = r.call(algo, abi.Params(pats, **kw), text)
            b = o.call(algo, abi.Params(pats, **kw), text)
            assert a[0] == b[0] and np.array_equal(a[1], b[1]), (r.name, abi.RA_NAMES[algo], pats, kw, text)
            checked += 1
    assert checked > 500


def test_memchr_batch_boundary_quirk():
    """max_count == 4096 with more matches: the reference stores the 4097th match FIRST
    (krep.c:3976-3991 + :4026-4038).  The restatement must reproduce the same list."""
    r = ol.ref(abi.REF_SCALAR) or ol.ref(abi.REF_AVX2)
    o = ol.oracle()
    text = (b"x#" * 5000)
    for mc in (4095, 4096, 4097, 8192, 100):
        a = r.call(abi.RA_MEMCHR, abi.Params([b"#"], max_count=mc), text)
        b = o.call(abi.RA_MEMCHR, abi.Params([b"#"], max_count=mc), text)
        assert a[0] == b[0] and np.array_equal(a[1], b[1]), mc
    a = r.call(abi.RA_MEMCHR, abi.Params([b"#"], max_count=4096), text)
    assert a[1][0, 0] == 2 * 4096 + 1  # the out-of-order record


def test_select_mirror_matches_reference():
    rng = random.Random(7)
    o = ol.oracle()
    pats = [b"a", b"ab", b"abc", b"abab", b"aaaa", b"abcabc", b"Sherlock", b"x" * 16, b"y" * 17,
            b"z" * 32, b"q" * 33, b"w" * 64, b"e" * 65, b"abcdefg", b"aabaab", b"ababab"]
    for lvl in LEVELS:
        r = ol.ref(lvl)
        if r is None:
            continue
        for pat in pats:
            for cs in (True, False):
                p = abi.Params([pat], case_sensitive=cs)
                assert r.select(p) == o.select(p, lvl), (lvl, pat, cs)
        p = abi.Params([b"ab", b"cd"])
        assert r.select(p) == o.select(p, lvl) == abi.RA_AHO_CORASICK


def test_larger_random_texts():
    rng = np.random.RandomState(5)
    o = ol.oracle()
    text = rng.choice(np.frombuffer(b"abc \n", dtype=np.uint8), size=300_000).astype(np.uint8)
    for lvl in LEVELS:
        r = ol.ref(lvl)
        if r is None:
            continue
        for algo, pat in ((abi.RA_BMH, b"abc a"), (abi.RA_MEMCHR, b"\n"), (abi.RA_MEMCHR_SHORT, b"ab"),
                          (abi.RA_KMP, b"aab"), (abi.RA_SSE42, b"abab"), (abi.RA_AVX2, b"abc abc abc abc abc"),
                          (abi.RA_AHO_CORASICK, None)):
            if not r.has(algo):
                continue
            pats = [pat] if pat else [b"ab", b"abc", b"c a", b"bcab", b"a"]
            for kw in (dict(), dict(count_lines=True), dict(whole_word=True), dict(case_sensitive=False, max_count=1000)):
                a = r.call(algo, abi.Params(pats, **kw), text)
                b = o.call(algo, abi.Params(pats, **kw), text)
                assert a[0] == b[0] and np.array_equal(a[1], b[1]), (r.name, abi.RA_NAMES[algo], kw)
