"""CPU-side checks of the drop-in boundary: libkrep_gpu.so builds for gfx950, loads, exports every entry
point include/krep_gpu.h declares, keeps the reference's struct layouts, and fails LOUDLY without a GPU
(no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from krep_amd import abi, build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    return C.CDLL(build.build())


def declared_functions():
    src = open(os.path.join(ROOT, "include", "krep_gpu.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = re.findall(r"\b([a-z_][a-z0-9_]*)\s*\([^;{]*\)\s*;", src)
    return sorted({n for n in names if n.startswith("krep_gpu_") or n.startswith("search_buffer")})


def test_every_declared_symbol_is_exported(lib):
    fns = declared_functions()
    assert len(fns) >= 30 and "search_buffer" in fns and "search_buffer_ex" in fns and "krep_gpu_literal_search" in fns
    for n in fns:
        assert hasattr(lib, n), n


def test_struct_layouts_match_reference_header():
    # krep.h:49-94 on LP64: match_position_t 16 B, match_result_t 24 B, search_params_t 72 B
    assert C.sizeof(abi.MatchPosition) == 16 and C.sizeof(abi.MatchResult) == 24 and C.sizeof(abi.SearchParams) == 72
    assert abi.SearchParams.case_sensitive.offset == 40 and abi.SearchParams.whole_word.offset == 45
    assert abi.SearchParams.compiled_regex.offset == 48 and abi.SearchParams.max_count.offset == 64


def test_mirror_of_select_search_algorithm_matches_oracle(lib):
    import krep_amd
    import oracle_lib as ol
    e = krep_amd.load()
    o = ol.oracle()
    pats = [b"a", b"ab", b"abc", b"abab", b"aaaa", b"abcabc", b"Sherlock", b"x" * 16, b"y" * 17, b"z" * 32, b"q" * 33,
            b"w" * 64, b"e" * 65, b"aabaab"]
    top = {abi.RA_AVX512: abi.RA_AVX512, abi.RA_AVX2: abi.RA_AVX2}
    for lvl in (abi.REF_SCALAR, abi.REF_SSE42, abi.REF_AVX2, abi.REF_AVX512):
        e.set_reference_simd(lvl)
        for pat in pats:
            for cs in (True, False):
                p = abi.Params([pat], case_sensitive=cs)
                sel = o.select(p, lvl)
                eff = e.mirror_select(p, 1000)
                # the effective algorithm is the selected one after the reference's internal delegation
                m = len(pat)
                if sel == abi.RA_AVX512:
                    want = abi.RA_AVX512 if m > 32 else (abi.RA_AVX2 if m > 16 else abi.RA_SSE42)
                elif sel == abi.RA_AVX2:
                    want = abi.RA_BMH if not cs else (abi.RA_AVX2 if m > 16 else abi.RA_SSE42)
                else:
                    want = sel
                assert eff == want, (lvl, pat, cs, sel, eff)
    e.set_reference_simd(abi.REF_AVX2)


def test_fails_loudly_without_gpu():
    import krep_amd
    e = krep_amd.load()
    if e.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(krep_amd.KrepGpuError):
        e.search(abi.Params([b"x"]), b"xxx")
    rc, n, _ = e.search_buffer(abi.Params([b"x"]), b"xxx")
    assert rc == 2  # error, like search_file()/search_string() (krep.h:159,168)


def test_host_generator_is_deterministic_and_sliceable():
    import krep_amd
    e = krep_amd.load()
    a = e.generate_host(5000, 0, 2, 42, b"Sherlock", 1000)
    b = e.generate_host(3000, 1500, 2, 42, b"Sherlock", 1000)
    assert np.array_equal(a[1500:4500], b)
    assert bytes(a).count(b"Sherlock") == 5
    c = e.generate_host(200_000, 0, 3, 7, b"#", 0)
    assert 1500 < int((c == ord("#")).sum()) < 2500


def test_header_is_plain_c_and_python_mirrors_its_structs(tmp_path):
    """include/krep_gpu.h is the boundary a C host (krep.c) compiles against: C11, -pedantic clean, and the ctypes mirrors of
    the configuration / boundary-record structs have the C layout (a silent mismatch would corrupt the caller's stack)."""
    import subprocess
    src = tmp_path / "abi_check.c"
    src.write_text('#include "krep_gpu.h"\n#include <stdio.h>\n#include <stddef.h>\n'
                   'int main(void) { printf("%zu %zu %zu %zu %zu %zu %zu\\n", sizeof(search_params_t), sizeof(match_result_t), '
                   'sizeof(krep_gpu_config_t), sizeof(krep_gpu_seq_carry_t), sizeof(krep_gpu_scan_out_t), '
                   'offsetof(krep_gpu_config_t, min_text_bytes), offsetof(krep_gpu_seq_carry_t, local_g0_kind)); return 0; }\n')
    exe = tmp_path / "abi_check"
    r = subprocess.run(["gcc", "-std=c11", "-Wall", "-Wextra", "-pedantic", "-Werror", f"-I{os.path.join(ROOT, 'include')}",
                        str(src), "-o", str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    sizes = [int(x) for x in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()]
    assert sizes == [C.sizeof(abi.SearchParams), C.sizeof(abi.MatchResult), C.sizeof(abi.Config), C.sizeof(abi.SeqCarry),
                     C.sizeof(abi.ScanOut), abi.Config.min_text_bytes.offset, abi.SeqCarry.local_g0_kind.offset], sizes


def test_split_mode_table():
    """Host logic, no GPU: which families may be cut how (krep_gpu_split_mode).  Independent pieces for the
    all-occurrence functions, chained pieces (one boundary record) for the sequential ones, one window for the three classes
    DESIGN.md §7 names."""
    import krep_amd
    e = krep_amd.load()
    n = 10_000_000
    W, P, CH = abi.SPLIT_WHOLE, abi.SPLIT_PIECES, abi.SPLIT_CHAIN
    table = [
        # (reference build, only_matching, patterns, params kwargs, expected)
        (abi.REF_AVX2, False, [b"Sherlock"], {}, P),                                   # border-free: greedy == all occurrences
        (abi.REF_AVX2, False, [b"Sherlock"], dict(count_lines=True), P),
        (abi.REF_AVX2, False, [b"abab"], {}, CH),                                      # simd_sse42_search, bordered: greedy walk
        (abi.REF_AVX2, False, [b"abab"], dict(count_lines=True), P),                   # plain -c needs no selection
        (abi.REF_AVX2, False, [b"abab"], dict(count_lines=True, whole_word=True), CH),
        (abi.REF_SCALAR, False, [b"abab"], {}, CH),                                    # kmp_search
        (abi.REF_SCALAR, False, [b"aXbY"], {}, P),                                     # boyer_moore_search
        (abi.REF_AVX2, True, [b"ab"], dict(case_sensitive=False), CH),                 # memchr_short_search under -o
        (abi.REF_AVX2, True, [b"abab"], dict(case_sensitive=False), CH),               # boyer_moore_search under -o
        (abi.REF_AVX2, True, [b"abba"], {}, P),                                        # SSE4.2 under -o: all occurrences
        (abi.REF_AVX2, False, [b"x" * 20], dict(count_lines=True), P),                 # AVX2 body without -w: canonical
        (abi.REF_AVX2, False, [b"x" * 20], dict(count_lines=True, whole_word=True), CH),  # ... with -w: end-of-text replay
        (abi.REF_AVX512, False, [b"x" * 40], dict(count_lines=True), CH),
        (abi.REF_AVX512, False, [b"x" * 40], {}, P),
        (abi.REF_NEON, False, [b"xyz"], dict(count_lines=True), CH),
        (abi.REF_NEON, False, [b"xyz"], dict(max_count=0, track_positions=False), W),  # neon_search's max_count == 0 corner
        (abi.REF_AVX2, False, [b"a\nb"], dict(count_lines=True), W),                   # the newline-pattern -c walk
        (abi.REF_AVX2, False, [b"a\nb"], {}, P),
        (abi.REF_AVX2, False, [b"ab", b"cd"], {}, P),
        (abi.REF_AVX2, False, [b"ab", b"cd"], dict(count_lines=True), P),
        (abi.REF_AVX2, False, [b"a\nb", b"cd"], dict(count_lines=True), W),            # multi-pattern -c, newline inside a pattern
        (abi.REF_AVX2, False, [b"e"], {}, P),
    ]
    try:
        for level, om, pats, kw, want in table:
            cfg = e.default_config()
            cfg.reference_simd, cfg.only_matching = level, int(om)
            e.set_thread_config(cfg)
            got = e.split_mode(abi.Params(pats, **kw), n)
            assert got == want, (level, om, pats, kw, got, want)
    finally:
        e.set_thread_config(None)


def test_size_policy_of_worthwhile(monkeypatch):
    """krep_gpu_worthwhile(): size first (no device is touched for a small text), then the input class, then the device."""
    import krep_amd
    e = krep_amd.load()
    p = abi.Params([b"Sherlock"])
    monkeypatch.setenv("KREP_GPU_ASSUME_AVAILABLE", "1")
    monkeypatch.delenv("KREP_GPU_DISABLE", raising=False)
    assert e.default_config().min_text_bytes == 1 << 20
    assert e.worthwhile(p, 1 << 20) and not e.worthwhile(p, (1 << 20) - 1)
    e.lib.krep_gpu_set_min_text_bytes(4096)
    try:
        assert e.worthwhile(p, 4096) and not e.worthwhile(p, 4095)
        r = abi.Params([b"a.*b"])
        r.s.use_regex = True
        assert not e.worthwhile(r, 1 << 30)
    finally:
        e.lib.krep_gpu_set_min_text_bytes((1 << 64) - 1)  # back to "not set": $KREP_GPU_MIN_BYTES, else 1 MiB
    monkeypatch.setenv("KREP_GPU_DISABLE", "1")
    assert not e.worthwhile(p, 1 << 30)
