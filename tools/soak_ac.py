"""Randomised soak of the multi-pattern scan against the CPU checker (more seeds and shapes than tests/test_gpu_ac.py)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import numpy as np
import krep_amd, oracle_lib as ol, cases
from krep_amd import abi
import test_gpu_ac as T
gpu = krep_amd.load(); o = ol.oracle()
bad = 0
for seed in range(2000, 2000 + (int(sys.argv[1]) if len(sys.argv) > 1 else 40)):
    rng = np.random.RandomState(seed)
    for it in range(40):
        alpha = [b"ab", b"abc\n", b"abAB -\n", bytes(range(97, 105)) + b" \n", bytes(range(256))][it % 5]
        n = [0, 3, 17, 500, 8192, 8195, 16384, 16385, 40000, 140000][rng.randint(0, 10)]
        text = cases.rand_text(rng, n, alpha)
        k = [2, 3, 5, 9, 40, 200][rng.randint(0, 6)]
        lens = [[1, 2, 3], [2, 3, 4, 6], [4, 5, 8, 16], [1, 4, 9, 30], [3, 3, 3], [4, 4, 4, 5], [5, 6, 7, 20]][rng.randint(0, 7)]
        pats = [cases.pick_pattern(rng, text, lens[rng.randint(0, len(lens))], alpha) for _ in range(k)]
        if rng.rand() < 0.3: pats.append(pats[0])
        kw = dict(case_sensitive=bool(rng.rand() < 0.6), whole_word=bool(rng.rand() < 0.25),
                  max_count=[abi.SIZE_MAX, abi.SIZE_MAX, abi.SIZE_MAX, 0, 1, 4, 77][rng.randint(0, 7)])
        mode = ["pos", "pos", "lines", "count"][rng.randint(0, 4)]
        if mode == "lines":
            if any(b"\n" in p for p in pats): continue
            kw.update(count_lines=True)
        elif mode == "count":
            kw.update(count_lines=True, only_match=True)
        try:
            T._check(gpu, o, text, pats, kw)
        except AssertionError as e:
            bad += 1; print("FAIL seed", seed, it, str(e)[:300], flush=True)
            if bad > 5: sys.exit(1)
print("soak done, failures:", bad)
