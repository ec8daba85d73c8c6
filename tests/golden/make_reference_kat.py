#!/usr/bin/env python3
"""Writes tests/golden/reference_kat.json: the known-answer vectors the reference's OWN tests assert
for the literal-scan hot path, transcribed (inputs + expected counts) with the file:line they come
from.  Counts only — the reference never asserts literal match offsets; offsets are
pinned by ref_vectors.json (generated from the compiled reference by gen_ref_vectors.py).

algos: bm kmp sse42 memchr memchr_short ac   (sse42 vectors are the reference's "latent" SIMD
assertions, enabled with -DKREP_USE_SSE42=1; oracle/_ref/krep_test runs them: 196/196).
"""
import json, os

K = []
def lit(src, algos, text, pat, ret, *, cs=True, lines=False, only=False, ww=False, maxc=None,
        track=None, text_len=None, res_count=None):
    K.append(dict(src=src, algos=algos, text=text, patterns=[pat], case_sensitive=cs,
                  count_lines=lines, only_match=only, whole_word=ww, max_count=maxc,
                  track_positions=track, text_len=text_len, expect=ret, expect_result_count=res_count))
def ac(src, text, pats, ret, *, cs=True, maxc=None, track=False, res_count=None, lines=False):
    K.append(dict(src=src, algos=["ac"], text=text, patterns=pats, case_sensitive=cs,
                  count_lines=lines, only_match=False, whole_word=False, max_count=maxc,
                  track_positions=track, count_matches_mode=not track, text_len=None, expect=ret,
                  expect_result_count=res_count))

fox = "The quick brown fox jumps over the lazy dog"
for pat, n in (("quick", 1), ("fox", 1), ("cat", 0)):
    lit("test/test_krep.c:326-372", ["bm", "kmp", "sse42"], fox, pat, n)
a17 = "a" * 17
lit("test/test_krep.c:394-402", ["kmp", "bm", "sse42"], a17, "a", 17)
lit("test/test_krep.c:406-411", ["bm", "kmp"], a17, "", 0)
lit("test/test_krep.c:413-418", ["bm", "kmp"], "", "test", 0)
lit("test/test_krep.c:421-430", ["kmp", "bm", "sse42"], "abcdef", "abc", 1)
lit("test/test_krep.c:432-441", ["kmp", "bm", "sse42"], "abcdef", "def", 1)
lit("test/test_krep.c:444-459", ["bm"], "abababa", "aba", 3)
lit("test/test_krep.c:444-459", ["kmp", "sse42"], "abababa", "aba", 2)
lit("test/test_krep.c:462-477", ["bm"], "aaaaa", "aa", 4)
lit("test/test_krep.c:462-477", ["kmp", "sse42"], "aaaaa", "aa", 2)
Fox = "The Quick Brown Fox Jumps Over The Lazy Dog"
lit("test/test_krep.c:491-495", ["bm", "sse42"], Fox, "quick", 0)
lit("test/test_krep.c:497-500", ["bm", "sse42"], Fox, "quick", 1, cs=False)
lit("test/test_krep.c:502-505", ["kmp"], Fox, "FOX", 0)
lit("test/test_krep.c:507-510", ["kmp"], Fox, "FOX", 1, cs=False)
cats = "cat scatter catalog cat catapult cat"
lit("test/test_krep.c:537-542", ["bm"], cats, "cat", 6)
lit("test/test_krep.c:545-558", ["bm", "kmp"], cats, "cat", 3, ww=True)
mm = "match match match\nno hits here\nmatch match\n"
lit("test/test_krep.c:572-580", ["bm", "kmp"], mm, "match", 2, lines=True)
lit("test/test_krep.c:582-587", ["memchr_short"], "abxxab\nxxab\n", "ab", 2, lines=True)
lit("test/test_krep.c:589-594", ["memchr"], "AaA a", "a", 4, cs=False)
abc4 = "abc---abc---abc---abc"
for tl, n in ((None, 4), (18, 3), (12, 2), (6, 1), (0, 0)):
    lit("test/test_krep.c:836-879", ["bm", "kmp", "sse42"], abc4, "abc", n, text_len=tl)
six = "line1: match\nline2: no\nline3: match\nline4: match\nline5: no\nline6: match"
for mc, n in ((2, 2), (4, 4), (5, 4), (1, 1), (0, 0)):
    lit("test/test_krep.c:897-932", ["bm"], six, "match", n, maxc=mc, res_count=n)
for mc, n in ((2, 2), (4, 4), (5, 4), (1, 1), (0, 0)):
    lit("test/test_krep.c:938-953", ["bm"], six, "match", n, lines=True, maxc=mc)
for mc, n in ((2, 2), (4, 4)):
    lit("test/test_krep.c:959-973", ["bm"], six, "match", n, only=True, maxc=mc, res_count=n)
fruit = "apple banana apple orange apple banana orange apple orange"
for mc in (3, 5, 6):
    ac("test/test_krep.c:1011-1065", fruit, ["apple", "orange"], mc, maxc=mc, track=True, res_count=mc)
lit("test/test_krep.c:1120-1129", ["bm"], "IP addresses: 192.168.1.1 and 10.0.0.1, ports: 8080 and 443",
    "192.168.1.1", 1)
lit("test/test_krep.c:1120-1129", ["bm"], "IP addresses: 192.168.1.1 and 10.0.0.1, ports: 8080 and 443",
    "8080", 1)
ac("test/test_krep.c:1167-1198", "foo bar baz foo qux bar", ["foo", "bar"], 4)
lit("test/test_krep.c:1203-1207", ["bm"], "\x00\x01\x02\x03\x04\x05\x06\x07", "abc", 0)
lit("test/test_krep.c:1211-1217", ["bm"], "xxxxPATTERNyyyy", "PATTERN", 0, text_len=8)
lit("test/test_krep.c:1211-1217", ["bm"], "PATTERNyyyy", "PATTERN", 1)
lit("test/test_krep.c:1221-1230", ["bm"], "aaaa", "aa", 3, only=True, track=True)
lit("test/test_krep.c:1233-1243", ["bm"], "word anotherword word", "word", 2, ww=True, track=True)
rep = " ".join(["match"] * 10)
for mc, n in ((0, 0), (1, 1), (3, 3), (5, 5), (10, 10), (None, 10)):
    lit("test/test_krep.c:1414-1450", ["bm"], rep, "match", n, maxc=mc, track=True, res_count=n)
lit("test/test_krep.c:1466-1473", ["bm", "kmp"], "abc", "abcdef", 0)
lit("test/test_krep.c:1307-1318", ["bm", "kmp"], "Test text to search within", "", 0)
ac("test/test_multiple_patterns.c:61-100", "ushers", ["he", "she", "his", "hers"], 3)
ac("test/test_multiple_patterns.c:61-100", "xyz", ["he", "she", "his", "hers"], 0)
ac("test/test_multiple_patterns.c:113-176", "UsHeRs", ["he", "she", "his", "hers"], 3, cs=False)
ac("test/test_multiple_patterns.c:113-176", "UsHeRs", ["HE", "SHE", "HIS", "HERS"], 3, cs=False)
ac("test/test_multiple_patterns.c:189-221", "abc", ["a", "b", "c", "ab", "bc", "abc"], 6)
ac("test/test_multiple_patterns.c:189-221", "", ["a", "b", "c", "ab", "bc", "abc"], 0)
ac("test/test_multiple_patterns.c:254-279", "abc", ["abcd", "abcde"], 0)
ac("test/test_multiple_patterns.c:288-340", "apple banana cherry", ["apple", "banana", "cherry"], 3,
   track=True, res_count=3)

out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_kat.json")
with open(out, "w") as f:
    json.dump(K, f, indent=1)
print(len(K), "vectors ->", out)
