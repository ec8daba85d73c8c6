"""Pins the oracle's -o (only_matching) behaviour against the reference CLI itself.

`only_matching` is a file-static of krep.c (krep.c:117) that no shared-object build can set, so the differential test
against oracle/_ref/*.so cannot reach it.  The stock CLI can: `krep -o` prints one `file:line:match` per match and
`krep -c -o` the number of matches.  This covers memchr_short_search's skip after a FAILED candidate (krep.c:4495),
the greedy inversion of boyer_moore_search (krep.c:1371) and the overlapping mode of simd_sse42_search (krep.c:4842)."""
import os
import random
import subprocess

import pytest

import oracle_lib as ol
from krep_amd import abi

CLI = ol.ref_cli()
pytestmark = pytest.mark.skipif(CLI is None, reason="oracle/_ref/krep not built (needs /root/reference)")


def run_cli(args, path):
    r = subprocess.run([CLI, "-t", "1", "--color=never"] + args + [str(path)], capture_output=True, timeout=60)
    return r.returncode, r.stdout


def expected_o_output(path, text, pos):
    out = []
    for s, e in pos.tolist():
        line = text.count(b"\n", 0, s) + 1
        out.append(b"%s:%d:%s\n" % (str(path).encode(), line, text[s:e]))
    return b"".join(out)


@pytest.mark.parametrize("seed", range(3))
def test_only_matching_families_match_the_cli(tmp_path, seed):
    rng = random.Random(4200 + seed)
    o = ol.oracle()
    o.set_only_matching(True)
    try:
        checked = 0
        for case in range(60):
            alpha = rng.choice([b"ab", b"abA\n", b"aAbB \n", b"abc_ \n", b"aB\n"])
            n = rng.choice([1, 7, 40, 200, 1000, 5000])
            # ends with a newline: the CLI's -o formatter mis-numbers matches behind the last '\n' once it has built its
            # newline index (krep.c:619-650, the search leaves a stale index) — a formatter matter, outside the scan path
            text = bytes(rng.choice(alpha) for _ in range(n)) + b"\n"
            n += 1
            path = tmp_path / f"t{case}.txt"
            path.write_bytes(text)
            letters = bytes(c for c in alpha if c not in b"\n")
            for m, cs in ((2, False), (3, False), (2, True), (3, True), (4, False), (5, True), (8, False), (40, True)):
                pat = bytes(rng.choice(letters) for _ in range(m))
                if n >= m and rng.random() < 0.6:
                    s = rng.randrange(0, n - m + 1)
                    if b"\n" not in text[s:s + m]:
                        pat = text[s:s + m]
                if pat.startswith(b"-") or pat.strip() != pat:
                    continue
                for ww in (False, True):
                    args = (["-i"] if not cs else []) + (["-w"] if ww else [])
                    p = abi.Params([pat], case_sensitive=cs, whole_word=ww)
                    algo = o.select(p, abi.REF_AVX2)  # oracle/_ref/krep is the AVX2 build (oracle/Makefile)
                    ret, pos = o.call(algo, p, text)
                    rc, out = run_cli(["-o"] + args + [pat.decode()], path)
                    assert out == expected_o_output(path, text, pos), (abi.RA_NAMES[algo], pat, cs, ww, text)
                    assert rc == (0 if len(pos) else 1)
                    pc = abi.Params([pat], case_sensitive=cs, whole_word=ww, count_lines=True, only_match=True)
                    retc, _ = o.call(o.select(pc, abi.REF_AVX2), pc, text)
                    rc2, out2 = run_cli(["-c", "-o"] + args + [pat.decode()], path)
                    assert out2.strip().split(b":")[-1] == str(retc).encode(), (abi.RA_NAMES[algo], pat, cs, ww, out2, retc)
                    checked += 1
        assert checked > 300
    finally:
        o.set_only_matching(False)
