"""Randomised soak of the single-literal scan against the CPU checker (more seeds than tests/test_gpu_literal.py)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import numpy as np
import krep_amd, oracle_lib as ol, cases
from krep_amd import abi
import test_gpu_literal as T

gpu = krep_amd.load(); o = ol.checker()  # the compiled reference, function by function (tests/oracle_lib.py)
bad = n = 0
for rounds in (1, 4):
    gpu.force_rounds(rounds)
    for seed in range(3000, 3000 + (int(sys.argv[1]) if len(sys.argv) > 1 else 12)):
        level = [abi.REF_SCALAR, abi.REF_SSE42, abi.REF_AVX2, abi.REF_AVX512][seed % 4]
        for text, pat, kw in cases.literal_cases(seed, 150):
            gpu.set_reference_simd(level)
            try:
                T._check(gpu, o, text, pat, kw, level)
                n += 1
            except Exception as e:
                bad += 1
                print("FAIL", seed, str(e)[:300], flush=True)
                if bad > 5:
                    sys.exit(1)
gpu.force_rounds(0)
print("soak done:", n, "cases, failures:", bad)
