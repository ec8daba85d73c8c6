"""The reference's OWN known answers through the HIP path (missing #2): every vector of
tests/golden/reference_kat.json — harvested from /root/reference/test/test_krep.c:326-372, :394-402, :444-477, :532-558,
:897-975 and test_multiple_patterns.c by tests/golden/make_reference_kat.py — is replayed through the C-ABI operator for
EVERY function the vector names.  A function is reached the way the reference's CLI reaches it: a (SIMD build level,
--algo override) pair for which select_search_algorithm() (krep.c:1771-1870, mirrored by krep_gpu_plan_ref_algo) returns that
function; the operator then has to reproduce that function's count (and number of stored records) on the GPU.  No oracle
is involved: the expected values are the reference's assertions themselves."""
import json
import os

import numpy as np
import pytest

from krep_amd import abi

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
KAT = json.load(open(os.path.join(HERE, "golden", "reference_kat.json")))
ALGO = {"bm": abi.RA_BMH, "kmp": abi.RA_KMP, "sse42": abi.RA_SSE42, "memchr": abi.RA_MEMCHR,
        "memchr_short": abi.RA_MEMCHR_SHORT, "ac": abi.RA_AHO_CORASICK}
LEVELS = (abi.REF_SCALAR, abi.REF_SSE42, abi.REF_AVX2, abi.REF_AVX512, abi.REF_NEON)
OVERRIDES = (abi.ALGO_AUTO, abi.ALGO_BM, abi.ALGO_KMP)


@pytest.fixture(scope="module")
def gpu():
    import krep_amd
    e = krep_amd.load()
    assert e.device_count() >= 1 and e.available(), e.unavailable_reason()
    yield e
    e.set_reference_simd(abi.REF_AVX2)
    e.set_algo_override(abi.ALGO_AUTO)


def _params(v):
    kw = dict(case_sensitive=v["case_sensitive"], count_lines=v["count_lines"], only_match=v["only_match"],
              whole_word=v["whole_word"], max_count=abi.SIZE_MAX if v["max_count"] is None else v["max_count"])
    if v.get("track_positions") is not None:
        kw["track_positions"] = v["track_positions"]
    p = abi.Params([s.encode("latin-1") for s in v["patterns"]], **kw)
    if v.get("count_matches_mode"):
        p.s.count_matches_mode = True
    return p


def _routes(gpu, v, algo):
    """(level, override) pairs under which the selector hands out `algo` for this vector's parameters"""
    out = []
    for lvl in LEVELS:
        for ov in OVERRIDES:
            gpu.set_reference_simd(lvl)
            gpu.set_algo_override(ov)
            p = _params(v)
            if not gpu.can_accelerate(p):
                continue
            plan = gpu.plan(p)
            top = plan.ref_algo
            plan.close()
            if top == algo:
                out.append((lvl, ov))
    return out


REACHED = {}


@pytest.mark.parametrize("i", range(len(KAT)))
def test_reference_known_answers_on_the_gpu(gpu, i):
    v = KAT[i]
    text = v["text"].encode("latin-1")
    if v["text_len"] is not None:
        text = text[: v["text_len"]]
    t = np.frombuffer(text, dtype=np.uint8)
    ran = 0
    for a in v["algos"]:
        routes = _routes(gpu, v, ALGO[a])
        REACHED[(i, a)] = len(routes)
        for lvl, ov in routes:
            gpu.set_reference_simd(lvl)
            gpu.set_algo_override(ov)
            ret, pos = gpu.search(_params(v), t)
            assert ret == v["expect"], (v["src"], a, "level", lvl, "override", ov, ret, v["expect"])
            if v["expect_result_count"] is not None:
                assert len(pos) == v["expect_result_count"], (v["src"], a, lvl, ov)
            ran += 1
    gpu.set_reference_simd(abi.REF_AVX2)
    gpu.set_algo_override(abi.ALGO_AUTO)
    assert ran >= 1, ("no selector route reaches any function this vector names", v["src"], v["algos"], v["patterns"])


def test_every_named_function_was_reached_somewhere(gpu):
    """run after the replay: each of the six functions the vectors name went through the HIP path, and the pairs the selector
    cannot produce for a vector's parameters (e.g. simd_sse42_search for a 20-byte pattern) are few and listed"""
    if len(REACHED) < len(KAT):
        pytest.skip("replay did not run in this session")
    per_algo = {}
    for (i, a), n in REACHED.items():
        per_algo.setdefault(a, [0, 0])
        per_algo[a][0 if n else 1] += 1
    for a in ALGO:
        assert per_algo[a][0] >= 1, (a, per_algo)
    unreachable = [(KAT[i]["src"], a, KAT[i]["patterns"]) for (i, a), n in REACHED.items() if n == 0]
    print("KAT replay: (vector, function) pairs on the GPU:", sum(1 for n in REACHED.values() if n), "unreachable:", unreachable)
    assert len(unreachable) <= len(REACHED) // 5, unreachable
