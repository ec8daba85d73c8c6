"""Seeded parity cases shared by the GPU tests and the golden-vector generator."""
from __future__ import annotations

import random

import numpy as np

from krep_amd import abi

ALPHAS = {
    "ab": b"ab",
    "abn": b"ab\n",
    "mixed": b"abAB \n",
    "word": b"abc_ \n-",
    "text": bytes(range(97, 123)) + b"  \n",
}
# lengths around every geometry edge of the kernel: 16 B lane, 1 KiB cell, 8 KiB wave segment,
# 32 KiB tile, and a few tiles
EDGE_SIZES = [0, 1, 2, 7, 8, 9, 15, 16, 17, 23, 24, 31, 32, 33, 63, 64, 65, 127, 128, 1000, 1023, 1024, 1025,
              1031, 2048, 8191, 8192, 8193, 8200, 16384, 32767, 32768, 32769, 32776, 40000, 65536, 65537, 98304 + 5,
              200003]


def rand_text(rng: np.random.RandomState, n: int, alpha: bytes) -> np.ndarray:
    a = np.frombuffer(alpha, dtype=np.uint8)
    return a[rng.randint(0, len(a), size=n)].astype(np.uint8) if n else np.zeros(0, dtype=np.uint8)


def pick_pattern(rng: np.random.RandomState, text: np.ndarray, m: int, alpha: bytes) -> bytes:
    n = text.size
    if n >= m and rng.rand() < 0.75:
        s = rng.randint(0, n - m + 1)
        return text[s:s + m].tobytes()
    return rand_text(rng, m, alpha).tobytes()


def literal_cases(seed: int, count: int, sizes=None, lens=None, big=False):
    """Yields (text ndarray, pattern bytes, kwargs for abi.Params)."""
    rng = np.random.RandomState(seed)
    pyr = random.Random(seed)
    sizes = sizes or EDGE_SIZES
    lens = lens or [1, 2, 3, 4, 5, 7, 8, 9, 12, 16, 17, 31, 32, 33, 64, 65, 200]
    for _ in range(count):
        alpha = ALPHAS[pyr.choice(list(ALPHAS))]
        n = pyr.choice(sizes)
        m = pyr.choice(lens)
        text = rand_text(rng, n, alpha)
        if m > 8 and n > m and rng.rand() < 0.6:
            # long patterns: plant a few copies so that verification is exercised
            pat = rand_text(rng, m, alpha).tobytes()
            for _k in range(pyr.choice([1, 2, 5])):
                s = rng.randint(0, n - m + 1)
                text[s:s + m] = np.frombuffer(pat, dtype=np.uint8)
        else:
            pat = pick_pattern(rng, text, m, alpha)
        kw = dict(case_sensitive=pyr.random() < 0.6, whole_word=pyr.random() < 0.25,
                  max_count=pyr.choice([abi.SIZE_MAX] * 4 + [0, 1, 2, 3, 7, 100]))
        mode = pyr.choice(["pos", "pos", "lines", "count"])
        if mode == "lines":
            kw.update(count_lines=True)
        elif mode == "count":
            kw.update(count_lines=True, only_match=True)
        yield text, pat, kw


def has_border(p: bytes) -> bool:
    return any(p[:len(p) - k] == p[k:] for k in range(1, len(p)))


def golden_cases():
    """The fixed case list behind tests/golden/ref_vectors.json: (id, text, patterns, kwargs, algo, simd level).
    Inputs are regenerated from seeds (numpy RandomState is a frozen stream); the OUTPUTS stored in the JSON were
    produced by the unmodified reference (oracle/_ref) via tests/golden/gen_ref_vectors.py."""
    out = []
    rng = np.random.RandomState(424242)
    pyr = random.Random(424242)
    single = [(abi.RA_BMH, abi.REF_SCALAR, [1, 2, 4, 8, 9, 17, 40, 100]), (abi.RA_KMP, abi.REF_SCALAR, [2, 4, 6]),
              (abi.RA_MEMCHR, abi.REF_SCALAR, [1]), (abi.RA_MEMCHR_SHORT, abi.REF_SCALAR, [2, 3]),
              (abi.RA_SSE42, abi.REF_SSE42, [2, 4, 8, 16]), (abi.RA_AVX2, abi.REF_AVX2, [17, 24, 32]),
              (abi.RA_AVX512, abi.REF_AVX512, [33, 48, 64])]
    cid = 0
    for algo, level, lens in single:
        for rep in range(14):
            alpha = ALPHAS[pyr.choice(list(ALPHAS))]
            m = pyr.choice(lens)
            n = max(pyr.choice([64, 777, 8192, 8200, 33000, 70001]), 2 * m)
            text = rand_text(rng, n, alpha)
            pat = pick_pattern(rng, text, m, alpha)
            if m > 8:
                for _k in range(3):
                    s = rng.randint(0, n - m + 1)
                    text[s:s + m] = np.frombuffer(pat, dtype=np.uint8)
            kw = dict(case_sensitive=True if algo in (abi.RA_SSE42, abi.RA_AVX2, abi.RA_AVX512) else pyr.random() < 0.6,
                      whole_word=pyr.random() < 0.2, max_count=pyr.choice([abi.SIZE_MAX] * 3 + [2, 9]))
            mode = pyr.choice(["pos", "pos", "lines", "count"])
            if mode == "lines":
                kw.update(count_lines=True)
            elif mode == "count":
                kw.update(count_lines=True, only_match=True)
            out.append((cid, text, [pat], kw, algo, level))
            cid += 1
    for rep in range(24):
        alpha = [b"ab", b"abc\n", b"abAB -\n", bytes(range(97, 105)) + b" \n"][rep % 4]
        n = pyr.choice([100, 8195, 40000, 90000])
        text = rand_text(rng, n, alpha)
        pats = [pick_pattern(rng, text, pyr.choice([1, 2, 3, 4, 6, 9]), alpha) for _ in range(pyr.choice([2, 4, 9]))]
        kw = dict(case_sensitive=pyr.random() < 0.6, whole_word=pyr.random() < 0.2,
                  max_count=pyr.choice([abi.SIZE_MAX] * 3 + [5]))
        if pyr.random() < 0.3 and not any(b"\n" in p for p in pats):
            kw.update(count_lines=True)
        out.append((cid, text, pats, kw, abi.RA_AHO_CORASICK, abi.REF_SCALAR))
        cid += 1
    # round 2: neon_search (the reference's arm64 path, built against oracle/neon_shim) and the -c paths through the
    # block-structured bodies; appended so that the vectors above keep their inputs
    rng = np.random.RandomState(535353)
    pyr = random.Random(535353)
    for algo, level, lens in ((abi.RA_NEON, abi.REF_NEON, [2, 3, 4, 8, 13, 16]), (abi.RA_AVX2, abi.REF_AVX2, [17, 20, 32]),
                              (abi.RA_AVX512, abi.REF_AVX512, [33, 40, 64])):
        for rep in range(16):
            alpha = [b"ab\n", b"ab \n", b"abc_ -\n", bytes(range(97, 123)) + b"  \n"][rep % 4]
            m = pyr.choice(lens)
            n = max(pyr.choice([90, 300, 777, 8200, 33000, 70001]), 2 * m)
            text = rand_text(rng, n, alpha)
            pat = rand_text(rng, m, alpha.replace(b"\n", b"")).tobytes()
            for _k in range(pyr.choice([1, 3, 6])):
                s = rng.randint(max(0, n - 150 - m), n - m + 1) if pyr.random() < 0.5 else rng.randint(0, n - m + 1)
                text[s:s + m] = np.frombuffer(pat, dtype=np.uint8)
            if rep % 5 == 4:
                text[text == 10] = ord("a")  # no newline at all
                text[n // 3] = 10
            kw = dict(case_sensitive=True, whole_word=pyr.random() < 0.5, max_count=pyr.choice([abi.SIZE_MAX] * 3 + [1, 4]))
            if rep % 4 != 3:
                kw.update(count_lines=True)
            out.append((cid, text, [pat], kw, algo, level))
            cid += 1
    return out
