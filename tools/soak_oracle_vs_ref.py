"""Differential soak of the oracle restatement against oracle/_ref (all SIMD levels, newline-rich alphabets).
usage: python tools/soak_oracle_vs_ref.py [seeds] [cases_per_seed]"""
import os, random, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import oracle_lib as ol  # noqa: E402
from krep_amd import abi  # noqa: E402
import test_oracle_vs_ref as T  # noqa: E402

T.ALPHAS = T.ALPHAS + [b"a\n", b"\n", b"ab\n\n", b"aA\n_"]
seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
per = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
o = ol.oracle()
refs = [ol.ref(l) for l in T.LEVELS if ol.ref_available(l)]
bad = n = 0
for seed in range(seeds):
    rng = random.Random(50_000 + seed)
    for _ in range(per):
        algo, pats, kw, text = T._case(rng)
        for om in (False, True):
            for r in refs:
                if not r.has(algo):
                    continue
                r.lib_only = getattr(r.lib, "only_matching", None)
                a = r.call(algo, abi.Params(pats, **kw), text)
                b = o.call(algo, abi.Params(pats, **kw), text)
                n += 1
                if not (a[0] == b[0] and np.array_equal(a[1], b[1])):
                    bad += 1
                    if bad <= 5:
                        print("MISMATCH", r.name, abi.RA_NAMES[algo], pats, kw, text, a[0], b[0], a[1][:6].tolist(), b[1][:6].tolist())
            break  # only_matching is a file-static in the reference: not settable through _ref
print(f"{n} comparisons, {bad} mismatches")
sys.exit(1 if bad else 0)
