"""Pins the oracle (oracle/krep_oracle.c) — and, when present, the compiled reference in oracle/_ref —
to the known-answer vectors asserted by the reference's own tests (tests/golden/reference_kat.json)."""
import json
import os

import pytest

import oracle_lib as ol
from krep_amd import abi

HERE = os.path.dirname(os.path.abspath(__file__))
KAT = json.load(open(os.path.join(HERE, "golden", "reference_kat.json")))
ALGO = {"bm": abi.RA_BMH, "kmp": abi.RA_KMP, "sse42": abi.RA_SSE42, "memchr": abi.RA_MEMCHR,
        "memchr_short": abi.RA_MEMCHR_SHORT, "ac": abi.RA_AHO_CORASICK}


def _params(v):
    kw = dict(case_sensitive=v["case_sensitive"], count_lines=v["count_lines"],
              only_match=v["only_match"], whole_word=v["whole_word"],
              max_count=abi.SIZE_MAX if v["max_count"] is None else v["max_count"])
    if v.get("track_positions") is not None:
        kw["track_positions"] = v["track_positions"]
    p = abi.Params([s.encode("latin-1") for s in v["patterns"]], **kw)
    if v.get("count_matches_mode"):
        p.s.count_matches_mode = True
    return p


def _engines():
    out = [ol.oracle()]
    for lvl in (abi.REF_SCALAR, abi.REF_AVX2):
        r = ol.ref(lvl)
        if r is not None:
            out.append(r)
    return out


@pytest.mark.parametrize("i", range(len(KAT)))
def test_reference_known_answers(i):
    v = KAT[i]
    text = v["text"].encode("latin-1")
    if v["text_len"] is not None:
        text = text[: v["text_len"]]
    for eng in _engines():
        for a in v["algos"]:
            algo = ALGO[a]
            if not eng.has(algo):
                continue
            ret, pos = eng.call(algo, _params(v), text)
            assert ret == v["expect"], (eng.name, a, v["src"])
            if v["expect_result_count"] is not None:
                assert len(pos) == v["expect_result_count"], (eng.name, a, v["src"])
