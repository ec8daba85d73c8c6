#!/usr/bin/env python3
"""Generates tests/golden/ref_vectors.json by RUNNING THE UNMODIFIED REFERENCE (oracle/_ref/*.so, built
from /root/reference by oracle/Makefile) on the seeded cases of tests/cases.py:golden_cases().  Stored per case:
the function's return value, result->count, a SHA-256 of the (start,end) records and the first/last records.
Run in the build container (where /root/reference exists); the JSON is committed so that the GPU box — which has
no /root/reference — still checks against reference-produced outputs."""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import cases  # noqa: E402
import oracle_lib as ol  # noqa: E402
from krep_amd import abi  # noqa: E402

out = []
for cid, text, pats, kw, algo, level in cases.golden_cases():
    ref = ol.ref(level)
    assert ref is not None and ref.has(algo), (level, algo)
    ret, pos = ref.call(algo, abi.Params(pats, **kw), text)
    out.append(dict(id=cid, algo=abi.RA_NAMES[algo], level=level, n=int(text.size), npat=len(pats), ret=ret,
                    count=int(len(pos)), sha256=hashlib.sha256(pos.tobytes()).hexdigest(),
                    first=pos[:2].tolist(), last=pos[-1:].tolist()))
json.dump(out, open(os.path.join(HERE, "ref_vectors.json"), "w"), indent=0)
print(len(out), "vectors;", sum(1 for o in out if o["count"]), "with positions")
