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
    # every header under include/: the drop-in boundary (krep_gpu.h) and the test hooks (krep_gpu_debug.h, moved out in round 6)
    src = "".join(open(os.path.join(ROOT, "include", f)).read() for f in sorted(os.listdir(os.path.join(ROOT, "include"))) if f.endswith(".h"))
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = re.findall(r"\b([a-z_][a-z0-9_]*)\s*\([^;{]*\)\s*;", src)
    return sorted({n for n in names if n.startswith("krep_gpu_") or n.startswith("search_buffer")})


def test_every_declared_symbol_is_exported(lib):
    fns = declared_functions()
    assert len(fns) >= 30 and "search_buffer" in fns and "search_buffer_ex" in fns and "krep_gpu_literal_search" in fns
    for n in fns:
        assert hasattr(lib, n), n
    # the boundary header itself declares no test hook any more
    boundary = open(os.path.join(ROOT, "include", "krep_gpu.h")).read()
    assert "krep_gpu_debug_" not in boundary
    assert any(n.startswith("krep_gpu_debug_") for n in fns)


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


def test_repetitive_pattern_predicate_exhaustively_against_the_reference_selector(lib):
    """KMP-or-BMH on a scalar build hangs on is_repetitive_pattern() (krep.c:1860-1865, :1873-1914).  The mirror states it as a
    property (a run longer than m/2, or a period in [2, m/2], read off the border chain — kg_mirror.hip); here it is compared
    with the compiled reference's own select_search_algorithm() for EVERY pattern of 3..9 bytes over {a, b, c} (and the
    restatement where oracle/_ref is absent)."""
    import itertools
    import krep_amd
    import oracle_lib as ol
    e = krep_amd.load()
    r = ol.ref(abi.REF_SCALAR)
    o = ol.oracle()
    e.set_reference_simd(abi.REF_SCALAR)
    try:
        n = 0
        for m in range(3, 10):
            for tup in itertools.product(b"abc", repeat=m):
                pat = bytes(tup)
                p = abi.Params([pat])
                want = r.select(p) if r is not None else o.select(p, abi.REF_SCALAR)
                assert e.mirror_select(p, 1000) == want, pat
                n += 1
        assert n == sum(3 ** m for m in range(3, 10))
    finally:
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
    all-occurrence functions, chained pieces (one boundary record) for the sequential ones, one window for the one class
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
        (abi.REF_AVX2, False, [b"a\nb"], dict(count_lines=True), CH),                  # the newline-pattern -c walk (r05: chained)
        (abi.REF_AVX2, False, [b"a\nb"], {}, P),
        (abi.REF_AVX2, False, [b"ab", b"cd"], {}, P),
        (abi.REF_AVX2, False, [b"ab", b"cd"], dict(count_lines=True), P),
        (abi.REF_AVX2, False, [b"a\nb", b"cd"], dict(count_lines=True), CH),           # multi-pattern -c, newline inside a pattern (r05: chained)
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
    """krep_gpu_worthwhile(): size first (no device is touched for a small text), then the input class, then the COST MODEL
    (kg_cost.hip: t_gpu = init + launch + bytes / host-path rate against t_cpu = bytes / min(threads x rate, cap) of the function
    the reference would run), then the device.  VERDICT r03 item 4: for a host-resident single literal on a many-core box the
    CPU pointer is the right answer, for a 1000-pattern dictionary the GPU."""
    import krep_amd
    e = krep_amd.load()
    p = abi.Params([b"Sherlock"])
    monkeypatch.setenv("KREP_GPU_ASSUME_AVAILABLE", "1")
    monkeypatch.delenv("KREP_GPU_DISABLE", raising=False)
    monkeypatch.delenv("KREP_GPU_COST", raising=False)
    monkeypatch.delenv("KREP_GPU_COST_MODEL", raising=False)
    e.set_cost_rates(None)
    assert e.default_config().min_text_bytes == 1 << 20
    rates = e.cost_rates()
    assert rates.enabled == 1 and rates.gpu_host_path_gbps > 0 and rates.cpu_simd_cap_gbps > rates.gpu_host_path_gbps
    try:
        # ---- the size rule alone (cost model off): exactly the threshold
        off = e.cost_rates()
        off.enabled = 0
        e.set_cost_rates(off)
        assert e.worthwhile(p, 1 << 20) and not e.worthwhile(p, (1 << 20) - 1)
        e.lib.krep_gpu_set_min_text_bytes(4096)
        assert e.worthwhile(p, 4096) and not e.worthwhile(p, 4095)
        r = abi.Params([b"a.*b"])
        r.s.use_regex = True
        assert not e.worthwhile(r, 1 << 30)
        e.lib.krep_gpu_set_min_text_bytes((1 << 64) - 1)  # back to "not set": $KREP_GPU_MIN_BYTES, else 1 MiB
        # ---- the cost model with its default rates
        e.set_cost_rates(None)
        dic = abi.Params([bytes([97 + (i * 7 + j) % 26 for j in range(4 + i % 13)]) for i in range(1000)])
        one_thread, many = 1, 256
        # a single SIMD literal: 256 threads read host memory faster than PCIe delivers it -> the CPU function, at any size
        for n in (8 << 20, 1 << 30, 32 << 30):
            c = e.cost_estimate(p, n, many)
            assert c.cpu_algo in (abi.RA_SSE42, abi.RA_AVX2) and c.cpu_threads == many and c.cpu_seconds < c.gpu_seconds
            assert not e.worthwhile_ex(p, n, many)
        # ... on ONE thread (krep -t 1) the GPU wins once the device start is paid for (6 GB/s against 50)
        assert not e.worthwhile_ex(p, 64 << 20, one_thread)      # 0.01 s of CPU work < 0.45 s of device start
        assert e.worthwhile_ex(p, 8 << 30, one_thread)           # 1.4 s against 0.6 s
        # 1000 patterns (aho_corasick_search: 8 605 states x 2 KiB do not fit any cache): the GPU, from tens of MiB on
        c = e.cost_estimate(dic, 1 << 30, many)
        assert c.cpu_algo == abi.RA_AHO_CORASICK and c.gpu_seconds < c.cpu_seconds
        assert e.worthwhile_ex(dic, 1 << 30, many) and e.worthwhile(dic, 32 << 30)
        assert not e.worthwhile_ex(dic, 2 << 20, many)           # 2 MiB: the device start alone costs more
        # the default thread count is search_file()'s: min(cores, size / 4 MiB), at least 1 (krep.c:2748-2759)
        assert e.cost_estimate(p, 3 << 20).cpu_threads == 1
        assert e.cost_estimate(p, 64 << 20).cpu_threads == min(os.cpu_count() or 1, 16)
        # overriding a rate moves the verdict: a host whose memchr path is slow
        slow = e.cost_rates()
        slow.cpu_simd_gbps, slow.cpu_simd_cap_gbps, slow.gpu_init_ms = 0.1, 1.0, 0.0
        e.set_cost_rates(slow)
        assert e.worthwhile_ex(p, 64 << 20, many)
        e.set_cost_rates(None)
        monkeypatch.setenv("KREP_GPU_COST", "simd=0.1:1,init=0")
        assert e.worthwhile_ex(p, 64 << 20, many)
        monkeypatch.delenv("KREP_GPU_COST")
        e.set_cost_rates(None)
        monkeypatch.setenv("KREP_GPU_COST_MODEL", "0")
        assert e.worthwhile_ex(p, 64 << 20, many)                # the size rule again
        monkeypatch.delenv("KREP_GPU_COST_MODEL")
    finally:
        e.lib.krep_gpu_set_min_text_bytes((1 << 64) - 1)
        e.set_cost_rates(None)
    monkeypatch.setenv("KREP_GPU_DISABLE", "1")
    off = e.cost_rates()
    off.enabled = 0
    e.set_cost_rates(off)
    try:
        assert not e.worthwhile(p, 1 << 30)
    finally:
        e.set_cost_rates(None)
