"""Reference-produced golden vectors (tests/golden/ref_vectors.json, generated from the compiled, unmodified
reference) against (a) the oracle restatement on CPU and (b) the HIP path on the GPU."""
import hashlib
import json
import os

import numpy as np
import pytest

import cases
import oracle_lib as ol
from krep_amd import abi

HERE = os.path.dirname(os.path.abspath(__file__))
VEC = json.load(open(os.path.join(HERE, "golden", "ref_vectors.json")))
CASES = cases.golden_cases()
assert len(VEC) == len(CASES)


def _same(v, ret, pos):
    return (ret == v["ret"] and len(pos) == v["count"]
            and hashlib.sha256(np.ascontiguousarray(pos, dtype=np.uint64).tobytes()).hexdigest() == v["sha256"])


def test_oracle_reproduces_reference_vectors():
    o = ol.oracle()
    for v, (cid, text, pats, kw, algo, level) in zip(VEC, CASES):
        assert v["id"] == cid and v["n"] == text.size
        ret, pos = o.call(algo, abi.Params(pats, **kw), text)
        assert _same(v, ret, pos), (cid, v["algo"], pats, kw)


@pytest.mark.gpu
def test_gpu_reproduces_reference_vectors():
    import krep_amd
    e = krep_amd.load()
    checked = 0
    for v, (cid, text, pats, kw, algo, level) in zip(VEC, CASES):
        e.set_reference_simd(level)
        p = abi.Params(pats, **kw)
        if len(pats) == 1:
            # force the reference function of the vector where the selector would choose another one
            e.set_algo_override(abi.ALGO_BM if algo == abi.RA_BMH else abi.ALGO_KMP if algo == abi.RA_KMP else abi.ALGO_AUTO)
            eff = e.mirror_select(p, text.size)
            if eff != algo:
                e.set_algo_override(abi.ALGO_AUTO)
                continue  # the vector names a function the selector would not execute for this input
            if not e.can_accelerate(p):
                assert e.select(p) is None
                e.set_algo_override(abi.ALGO_AUTO)
                continue  # left to the CPU by contract (krep_gpu_can_accelerate): memchr_short -c under -o
        ret, pos = e.search(p, text)
        e.set_algo_override(abi.ALGO_AUTO)
        assert _same(v, ret, pos), (cid, v["algo"], pats, kw, ret, v["ret"])
        checked += 1
    e.set_reference_simd(abi.REF_AVX2)
    assert checked > 80
