"""Randomised soak of the ANCHORED multi-pattern scan (kg_ac_anchor.hip: anchors by rarity, four- / five-class index, exact dictionary) against
the compiled reference's aho_corasick_search: texts of 1-3 MiB (the anchor decision is taken on texts >= 1 MiB), i.i.d. and word-like, word
dictionaries with near misses, every switch combination.   usage: python tools/soak_anchor.py [seeds]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import numpy as np
import krep_amd, oracle_lib as ol, cases, wordlist
from krep_amd import abi

gpu = krep_amd.load(); o = ol.checker()
W = wordlist.word_list(); blob = wordlist.pack(W)
SW = ("KREP_GPU_AC_ANCHOR", "KREP_GPU_AC_ANCHOR5", "KREP_GPU_AC_NO_ANCHOR5", "KREP_GPU_AC_NO_EXACT", "KREP_GPU_AC_NO_ANCHOR", "KREP_GPU_AC_UPT", "KREP_GPU_AC_NODEFER")
COMBOS = ({}, {"KREP_GPU_AC_ANCHOR": "1", "KREP_GPU_AC_NO_ANCHOR5": "1"}, {"KREP_GPU_AC_ANCHOR": "1", "KREP_GPU_AC_ANCHOR5": "1"},
          {"KREP_GPU_AC_ANCHOR": "1", "KREP_GPU_AC_ANCHOR5": "1", "KREP_GPU_AC_NO_EXACT": "1"}, {"KREP_GPU_AC_NO_ANCHOR5": "1"}, {"KREP_GPU_AC_NO_EXACT": "1"})
bad = n = 0
a0 = gpu.anchored_launches()
for seed in range(7000, 7000 + (int(sys.argv[1]) if len(sys.argv) > 1 else 20)):
    rng = np.random.RandomState(seed)
    for it in range(6):
        size = (1 << 20) + int(rng.randint(0, 2 << 20))
        wordy = it % 2 == 0
        if wordy:
            text = gpu.generate_host(size, int(rng.randint(0, 1 << 20)) * 80, 5, 20260930 + seed, blob, 80)
            k = [5, 50, 300, 1000][rng.randint(0, 4)]
            pool = [wordlist.dictionary(W, kind, n=k, seed=seed + it, min_len=ml, max_len=16) for kind, ml in (("rare", 4), ("uniform", 6), ("common", 4))][rng.randint(0, 3)]
            pats = list(pool)
            for j in range(min(20, len(pats))):  # near misses: a dictionary word with one letter changed, a suffix, an affix
                w = bytearray(pats[j]); w[rng.randint(0, len(w))] = 97 + rng.randint(0, 26); pats.append(bytes(w))
            pats += [pats[0][1:] if len(pats[0]) > 4 else pats[0], b"tion", b"ness", b"ation"][: rng.randint(0, 5)]
            pats = [p for p in pats if len(p) >= 4]
            if rng.rand() < 0.35:  # a few 1..3-byte words beside them: the plan scans the two parts on their own and merges the lists (kg_scan.hip scan_ac_split)
                shorts = [w for w in W if len(w) <= 3]
                pats += [shorts[rng.randint(0, len(shorts))] for _ in range(rng.randint(1, 4))] + [b"of", b"a"][: rng.randint(0, 3)]
        else:
            alpha = [b"ab", b"abcdefgh \n", bytes(range(97, 123)) + b"  \n", b"abAB -\n"][rng.randint(0, 4)]
            text = cases.rand_text(rng, size, alpha)
            k = [3, 9, 40, 300][rng.randint(0, 4)]
            lens = [[4, 5, 6], [4, 5, 8, 16], [5, 9, 13, 16], [6, 7, 30, 64], [4, 4, 4, 12], [7, 8, 9, 16]][rng.randint(0, 6)]
            pats = [cases.pick_pattern(rng, text, lens[rng.randint(0, len(lens))], alpha) for _ in range(k)]
        if rng.rand() < 0.3:
            pats.append(pats[0])
        for s in (0, 1, 9, 16383, 16384, 16385, 32767, size - 20):  # matches on the seams and inside the first 16 bytes
            p = np.frombuffer(pats[rng.randint(0, len(pats))], dtype=np.uint8)
            if 0 <= s and s + p.size <= size:
                text[s:s + p.size] = p
        kw = dict(case_sensitive=bool(rng.rand() < 0.7), whole_word=bool(rng.rand() < 0.2), max_count=[abi.SIZE_MAX] * 3 + [1, 500])
        kw["max_count"] = kw["max_count"][rng.randint(0, 5)]
        if rng.rand() < 0.25:
            kw.update(count_lines=True, only_match=True)
        env = dict(COMBOS[rng.randint(0, len(COMBOS))])
        if rng.rand() < 0.6:  # tickets of several units (what large texts get): the deferred verify stage, batches that span units
            env["KREP_GPU_AC_UPT"] = str([2, 3, 5, 8][rng.randint(0, 4)])
        if rng.rand() < 0.15:
            env["KREP_GPU_AC_NODEFER"] = "1"
        os.environ.update(env)
        try:
            want = o.call(abi.RA_AHO_CORASICK, abi.Params(pats, **kw), text)
            got = gpu.search(abi.Params(pats, **kw), text)
            n += 1
            if got[0] != want[0] or not np.array_equal(got[1], want[1]):
                bad += 1
                print("FAIL seed", seed, it, "wordy" if wordy else "iid", env, kw, len(pats), got[0], want[0], flush=True)
                if bad > 5:
                    sys.exit(1)
        finally:
            for kk in SW:
                os.environ.pop(kk, None)
print("soak done:", n, "cases,", gpu.anchored_launches() - a0, "anchored launches, failures:", bad)
sys.exit(1 if bad else 0)
