"""CPU pin of the end-of-text replay (krep_amd/csrc/kg_replay.h) and of the window decomposition around it.

-c through simd_avx2_search (-w), simd_avx512_search and neon_search is computed by the product as
    canonical distinct-line count over the starts in [0, n-256)   (scan kernel, GPU)
  + replay_lines() from the block-loop position `cur` the reference has when it enters the last 256 bytes.
The replay is __host__ __device__; here the SAME function (exported as krep_gpu_debug_replay_host, no GPU needed) is
combined with a numpy/bytes restatement of the decomposition and compared with the oracle on thousands of texts.  The
GPU tests (tests/test_gpu_replay.py) then check the device side (last accepted occurrence, newline searches)."""
import ctypes as C
import random

import pytest

import oracle_lib as ol
from krep_amd import abi, build

W = 256
BLOCK = {abi.RA_AVX2: 32, abi.RA_AVX512: 64, abi.RA_NEON: 16}


@pytest.fixture(scope="module")
def lib():
    L = C.CDLL(build.build())
    L.krep_gpu_debug_replay_host.restype = C.c_uint64
    L.krep_gpu_debug_replay_host.argtypes = [C.c_int, C.c_char_p, C.c_size_t, C.c_char_p, C.c_uint32, C.c_int, C.c_size_t, C.c_int]
    L.krep_gpu_debug_fold_carry.restype = None
    L.krep_gpu_debug_fold_carry.argtypes = [C.POINTER(abi.SeqCarry)] * 3
    return L


def wordc(b):
    return chr(b).isalnum() and b < 128 or b == 95


def accepted(text, pat, ww):
    n, m = len(text), len(pat)
    out = []
    i = text.find(pat)
    while i != -1:
        ok = True
        if ww:
            if i > 0 and wordc(text[i - 1]):
                ok = False
            elif i + m < n and wordc(text[i + m]):
                ok = False
        if ok:
            out.append(i)
        i = text.find(pat, i + 1)
    return out


def decompose(text, pat, ww, algo):
    """(K, cur, open): what the device side hands to the replay (kg_scan.hip: replay_entry)."""
    n, B = len(text), BLOCK[algo]
    X = n - W if n > W else 0
    if X == 0:
        return 0, 0, 0
    before = [i for i in accepted(text, pat, ww) if i < X]
    K = len({text.rfind(b"\n", 0, i) + 1 for i in before})
    if not before:
        return K, (X // B) * B, 0
    q = before[-1]
    nl = text.find(b"\n", q)
    if nl != -1:
        nls = nl + 1
        return K, (nls + ((X - nls) // B) * B if nls <= X else nls), 0
    if algo != abi.RA_NEON:
        return K, n, 0
    ls = text.rfind(b"\n", 0, q) + 1
    b2 = [i for i in before if i < ls]
    grid0 = text.find(b"\n", b2[-1]) + 1 if b2 else 0
    return K, grid0 + ((X - grid0) // B) * B, 1


def make_case(rng, algo):
    m = rng.choice({abi.RA_AVX2: [17, 18, 24, 31, 32], abi.RA_AVX512: [33, 34, 40, 63, 64],
                    abi.RA_NEON: [2, 3, 4, 5, 8, 15, 16]}[algo])
    alpha = rng.choice([b"ab", b"ab\n", b"ab \n", b"abc_ -\n", b"a\n", b"ab" * 8 + b"\n"])
    n = rng.choice([0, 5, 40, 100, 255, 256, 257, 300, 320, 400, 511, 512, 700, 1000, 1500, 2500])
    text = bytearray(rng.choice(alpha) for _ in range(n))
    pat = bytes(rng.choice(alpha.replace(b"\n", b"") or b"a") for _ in range(m))
    if rng.random() < 0.15:
        pat = bytes(rng.choice(alpha) for _ in range(m))  # may contain '\n'
    for _ in range(rng.choice([0, 1, 2, 4, 8])):
        if n >= m:
            near_end = rng.random() < 0.6
            s = rng.randrange(max(0, n - 200 - m), n - m + 1) if near_end else rng.randrange(0, n - m + 1)
            text[s:s + m] = pat
    if rng.random() < 0.3 and n:
        # long unterminated last line
        cut = rng.randrange(0, n)
        for i in range(cut, n):
            if text[i] == 10:
                text[i] = ord("a")
    return bytes(text), pat


@pytest.mark.parametrize("algo", [abi.RA_AVX2, abi.RA_AVX512, abi.RA_NEON])
@pytest.mark.parametrize("seed", range(4))
def test_replay_plus_canonical_prefix_equals_reference_function(lib, algo, seed):
    rng = random.Random(9000 + 10 * algo + seed)
    o = ol.oracle()
    checked = 0
    for _ in range(700):
        text, pat = make_case(rng, algo)
        if len(text) < len(pat) or (algo != abi.RA_NEON and pat[-1] == 0):
            continue
        for ww in ((True,) if algo == abi.RA_AVX2 else (False, True)):
            maxc = rng.choice([abi.SIZE_MAX, abi.SIZE_MAX, 1, 2, 5])
            want, _ = o.call(algo, abi.Params([pat], count_lines=True, whole_word=ww, max_count=maxc), text)
            K, cur, opn = decompose(text, pat, ww, algo)
            extra = lib.krep_gpu_debug_replay_host(algo, text, len(text), pat, len(pat), int(ww), cur, opn) if cur < len(text) else 0
            got = min(K + extra, maxc)
            assert got == want, (abi.RA_NAMES[algo], pat, ww, maxc, len(text), K, cur, opn, extra, want, text)
            checked += 1
    assert checked > 400


# ---- the same in PIECES (round 3): every piece contributes its own part of the line-skip history, the product's fold
# (kg::fold_carry, exported as krep_gpu_debug_fold_carry) combines them in text order — whatever order they were scanned in —
# and only the piece that ends the text replays.  The per-piece quantities below restate what the device side derives
# (kg_scan.hip, the replay branch of scan_literal for a window inside the text).
def piece_record(text, acc, lo, hi, X, final, algo):
    n = len(text)
    lim, nl_end = (X, n) if final else (hi, hi)
    pc = abi.SeqCarry()
    mine = [i for i in acc if lo <= i < lim]
    first_nl = text.find(b"\n", lo, nl_end)
    pc.local_first_nl1 = first_nl + 1 if first_nl != -1 else 0
    if not mine:
        return pc
    q = mine[-1]
    nl = text.find(b"\n", q, nl_end)
    pc.local_q1 = q + 1
    pc.local_nl1 = nl + 1 if nl != -1 else 0
    if algo == abi.RA_NEON:
        p = text.rfind(b"\n", 0, q)
        if p < lo:  # q's line started in front of this piece (a newline in the halo belongs to the piece before)
            pc.local_g0_kind = 3
        else:
            ls = p + 1
            b2 = [i for i in mine if i < ls]
            if not b2:
                pc.local_g0_kind = 2
            else:
                pc.local_g0_kind = 1
                pc.local_g0 = text.find(b"\n", b2[-1]) + 1
    return pc


@pytest.mark.parametrize("algo", [abi.RA_AVX2, abi.RA_AVX512, abi.RA_NEON])
@pytest.mark.parametrize("seed", range(3))
def test_chained_pieces_fold_equals_reference_function(lib, algo, seed):
    rng = random.Random(7000 + 10 * algo + seed)
    o = ol.checker()
    B = BLOCK[algo]
    checked = 0
    for _ in range(500):
        text, pat = make_case(rng, algo)
        n = len(text)
        X = n - W if n > W else 0
        if n < len(pat) or X < B + 2 or (algo != abi.RA_NEON and pat[-1] == 0):
            continue
        for ww in ((True,) if algo == abi.RA_AVX2 else (False, True)):
            want, _ = o.call(algo, abi.Params([pat], count_lines=True, whole_word=ww), text)
            acc = accepted(text, pat, ww)
            cuts = sorted({0, n} | {rng.randrange(1, X - B + 1) for _ in range(rng.choice([1, 2, 3, 6]))})
            pieces = list(zip(cuts[:-1], cuts[1:]))
            recs = [piece_record(text, acc, lo, hi, X, hi == n, algo) for lo, hi in pieces]
            tc = abi.SeqCarry()
            recs[-1].local_lines = 1 + 4242  # (the end piece's canonical count rides on its record: krep_gpu_replay_tail starts from it)
            for pc in recs:  # the left fold, with the product's own function
                out = abi.SeqCarry()
                lib.krep_gpu_debug_fold_carry(C.byref(tc), C.byref(pc), C.byref(out))
                tc = out
            assert tc.local_lines == 1 + 4242
            if not tc.q1:
                cur, opn = (X // B) * B, 0
            elif tc.nl1:
                cur, opn = (tc.nl1 + ((X - tc.nl1) // B) * B if tc.nl1 <= X else tc.nl1), 0
            elif algo == abi.RA_NEON:
                cur, opn = tc.g0 + ((X - tc.g0) // B) * B, 1
            else:
                cur, opn = n, 0
            K = len({text.rfind(b"\n", 0, i) + 1 for i in acc if i < X})
            extra = lib.krep_gpu_debug_replay_host(algo, text, n, pat, len(pat), int(ww), cur, opn) if cur < n else 0
            assert K + extra == want, (abi.RA_NAMES[algo], pat, ww, n, cuts, K, cur, opn, extra, want, text)
            # ... and the folded record is the one the whole-text derivation arrives at
            wK, wcur, wopn = decompose(text, pat, ww, algo)
            assert (cur, opn) == (wcur, wopn), (abi.RA_NAMES[algo], pat, ww, n, cuts, cur, opn, wcur, wopn, text)
            checked += 1
    assert checked > 150
