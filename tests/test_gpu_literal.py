"""GPU parity: the HIP literal scan, called through the C-ABI, against the COMPILED REFERENCE (oracle/_ref: the function
select_search_algorithm() would have executed, called directly; the restatement oracle/krep_oracle.c only under the file-static
only_matching) — bit-exact count + offsets."""
import numpy as np
import pytest

import cases
import oracle_lib as ol
from krep_amd import abi

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=[1, 4], ids=["tile32k", "tile128k"])
def gpu(request):
    """Every test runs with both kernel tile shapes (1 round = 32 KiB tiles, 4 rounds = 128 KiB tiles)."""
    import krep_amd
    e = krep_amd.load()
    assert e.device_count() >= 1, "no MI355X visible"
    e.force_rounds(request.param)
    yield e
    e.force_rounds(0)


def _check(gpu, o, text, pat, kw, level):
    import krep_amd
    gpu.set_reference_simd(level)
    p = abi.Params([pat], **kw)
    algo = gpu.mirror_select(p, text.size)
    if not gpu.can_accelerate(p):
        # the input class the backend leaves to the CPU (include/krep_gpu.h: krep_gpu_can_accelerate): the selector
        # hands back NULL and the operator refuses loudly — never a silent approximation
        assert gpu.select(p) is None
        with pytest.raises(krep_amd.KrepGpuError):
            gpu.search(p, text)
        return
    want = o.call(algo, abi.Params([pat], **kw), text)
    got = gpu.search(p, text)
    assert got[0] == want[0], (abi.RA_NAMES[algo], pat, kw, text.size, got[0], want[0])
    assert np.array_equal(got[1], want[1]), (abi.RA_NAMES[algo], pat, kw, text.size, got[1][:8], want[1][:8])


@pytest.mark.parametrize("seed", range(6))
def test_all_occurrence_family(gpu, oracle_engine, seed):
    """Scalar reference build: BMH / memchr / memchr_short (family O) and KMP for repetitive 4..7 B."""
    n = 0
    for text, pat, kw in cases.literal_cases(100 + seed, 120):
        gpu.set_reference_simd(abi.REF_SCALAR)
        p = abi.Params([pat], **kw)
        algo = gpu.mirror_select(p, text.size)
        if algo == abi.RA_KMP and cases.has_border(pat if kw["case_sensitive"] else pat.lower()):
            continue  # greedy family with a bordered pattern: covered in test_gpu_greedy.py
        _check(gpu, oracle_engine, text, pat, kw, abi.REF_SCALAR)
        n += 1
    assert n > 60


@pytest.mark.parametrize("seed", range(8))
def test_simd_builds_border_free(gpu, oracle_engine, seed):
    """SSE4.2 / AVX2 / AVX-512 / NEON reference builds: SSE4.2 (<=16 B), AVX2 (17..32), AVX-512 (33..64) and NEON
    (2..16 B) bodies, every mode — including -c through the block-structured bodies (end-of-text replay)."""
    n = 0
    for text, pat, kw in cases.literal_cases(200 + seed, 120):
        level = [abi.REF_SSE42, abi.REF_AVX2, abi.REF_AVX512, abi.REF_NEON][seed % 4]
        gpu.set_reference_simd(level)
        p = abi.Params([pat], **kw)
        algo = gpu.mirror_select(p, text.size)
        folded = pat if kw["case_sensitive"] else pat.lower()
        if algo in (abi.RA_SSE42, abi.RA_KMP) and cases.has_border(folded):
            continue
        _check(gpu, oracle_engine, text, pat, kw, level)
        n += 1
    assert n > 50


def test_headline_literal_small(gpu, oracle_engine):
    """'Sherlock' planted in background text, every alignment of a match against 16 B / 1 KiB / 32 KiB edges."""
    rng = np.random.RandomState(3)
    text = cases.rand_text(rng, 3 * 32768 + 100, cases.ALPHAS["text"])
    pat = b"Sherlock"
    spots = [0, 9, 1017, 2044, 8185, 16380, 32761, 32769, 65530, 65539, text.size - 8]
    for s in spots:
        text[s:s + 8] = np.frombuffer(pat, dtype=np.uint8)
    for level in (abi.REF_SCALAR, abi.REF_AVX2):
        for kw in (dict(), dict(count_lines=True), dict(whole_word=True), dict(case_sensitive=False),
                   dict(max_count=5), dict(count_lines=True, only_match=True)):
            _check(gpu, oracle_engine, text, pat, kw, level)
    got = gpu.search(abi.Params([pat]), text)
    assert got[0] == len(spots) and got[1][:, 0].tolist() == sorted(spots)


def test_dense_single_byte(gpu, oracle_engine):
    rng = np.random.RandomState(4)
    text = cases.rand_text(rng, 300_000, b"#abcdefghij \n")
    for kw in (dict(), dict(count_lines=True), dict(max_count=4096), dict(max_count=8192), dict(max_count=4097),
               dict(whole_word=True), dict(case_sensitive=False)):
        _check(gpu, oracle_engine, text, b"#", kw, abi.REF_AVX2)
    _check(gpu, oracle_engine, text, b"A", dict(case_sensitive=False), abi.REF_AVX2)


def test_line_counting_at_density(gpu, oracle_engine):
    """-c where nearly every 1-KiB cell holds matches (round 5: the line bookkeeping is a carry chain over the 64 lanes of a cell,
    kg_literal.hip line_cell; a single byte runs it on the 0x80 flag words of the lane's 16 bytes): single bytes at 0.5-30 %
    hits with lines of every shape — none, only newlines, runs of blank lines, a newline every 15-17 bytes (one per lane, at the
    lane's first / last byte), lines longer than a cell and longer than a unit — the pattern '\n' itself, -i, -w, two- and
    four-byte patterns at density, whole text and ownership windows folded with krep_gpu_combine_line_counts, against
    memchr_search / memchr_short_search / boyer_moore_search of the compiled reference."""
    import torch
    rng = np.random.RandomState(5150)
    n = (3 << 20) + 4321
    texts = []
    for alpha in (b"ae\n", b"aaaaaaaaaaaaaaaaaaaaaaaaaaaaaae\n", b"abcdefghijklmnopqrstuvwxyz     \n", b"ab e", b"\n", b"e\n\n\n", b"Ee \n"):
        texts.append(cases.rand_text(rng, n, alpha))
    t = cases.rand_text(rng, n, b"abcde ")           # one newline per lane-sized stretch, drifting over the 16-byte grid
    t[np.arange(0, n, 17)] = 10
    texts.append(t)
    t = cases.rand_text(rng, n, b"abcde ")
    t[15::16] = 10                                    # the lane's last byte
    texts.append(t)
    t = cases.rand_text(rng, n, b"abcde ")
    t[0::16] = 10                                     # the lane's first byte
    texts.append(t)
    t = cases.rand_text(rng, n, b"abcde ")           # lines longer than a 1-KiB cell / than a 32-KiB unit
    t[rng.randint(0, n, 40)] = 10
    t[np.arange(5000, n, 70001)] = 10
    texts.append(t)
    t = cases.rand_text(rng, n, b"xyz\n")            # sparse hits between many lines, some lines with several
    t[rng.randint(0, n, 3000)] = ord("e")
    texts.append(t)
    pats = [(b"e", dict()), (b"e", dict(case_sensitive=False)), (b"E", dict(case_sensitive=False)), (b"e", dict(whole_word=True)),
            (b"\n", dict()), (b" ", dict()), (b"ab", dict()), (b"e\n", dict()), (b"abcd", dict()), (b"a", dict(max_count=1000))]
    for ti, text in enumerate(texts):
        d = torch.from_numpy(text).cuda()
        for pat, kw in pats:
            kwc = dict(count_lines=True, **kw)
            p = abi.Params([pat], **kwc)
            gpu.set_reference_simd(abi.REF_SCALAR)
            algo = gpu.mirror_select(p, text.size)
            if b"\n" in pat and len(pat) > 1 and algo not in (abi.RA_BMH, abi.RA_MEMCHR_SHORT):
                continue
            want = oracle_engine.call(algo, abi.Params([pat], **kwc), text)[0]
            got = gpu.search(p, text, want_result=False)[0]
            assert got == want, (ti, pat, kw, abi.RA_NAMES[algo], got, want)
            if "max_count" in kw or algo == abi.RA_KMP:
                continue
            # ownership windows: cuts on and off the 16-byte grid, inside a cell, at a unit boundary
            plan = gpu.plan(p)
            cuts = [0, 7, 1024 + 16, 32768, 32768 + 5, (1 << 20) + 1000, (2 << 20) + 15, n]
            outs = [plan.scan(d.data_ptr(), n, lo, hi) for lo, hi in zip(cuts[:-1], cuts[1:])]
            arr = (abi.ScanOut * len(outs))(*outs)
            assert gpu.lib.krep_gpu_combine_line_counts(arr, len(outs)) == want, (ti, pat, kw, [o.line_count for o in outs])
            plan.close()
        del d
    gpu.set_reference_simd(abi.REF_AVX2)


def test_staging_overflow_takes_emit_mode(gpu, oracle_engine):
    """Units with more hits than their staging slot are re-scanned in emit mode: force tiny slots."""
    rng = np.random.RandomState(9)
    text = cases.rand_text(rng, 150_000, b"ab#\n")
    try:
        gpu.set_algo_override(abi.ALGO_BM)  # all-occurrence family for every pattern (--algo=bm, krep.c:1788)
        for cap in (1, 3):
            gpu.force_stage_cap(cap)
            for pat, kw in ((b"#", dict()), (b"ab", dict()), (b"aba", dict(case_sensitive=False)),
                            (b"a#ba#b", dict()), (b"ab", dict(max_count=1000)), (b"#", dict(whole_word=True))):
                _check(gpu, oracle_engine, text, pat, kw, abi.REF_SCALAR)
    finally:
        gpu.force_stage_cap(0)
        gpu.set_algo_override(abi.ALGO_AUTO)


def test_avx512_unexamined_block_is_reproduced(gpu, oracle_engine):
    """simd_avx512_search steps over its last full 64-byte block when fewer than (m-1)+64 bytes remain
    (krep.c:5171): matches starting there are lost by the reference, hence by its drop-in."""
    rng = np.random.RandomState(12)
    pat = b"0123456789abcdefghijklmnopqrstuvwxyzABCDEFGH"  # 44 bytes -> AVX-512 body on an AVX-512 build
    for n in (64 * 5 + 10, 64 * 5 + 42, 64 * 5 + 43, 64 * 5 + 63, 64 * 200 + 3):
        text = cases.rand_text(rng, n, b"xyz \n")
        b = n - n % 64
        for s in (b - 64, b - 50, b - 130, 3):
            if 0 <= s and s + len(pat) <= n:
                text[s:s + len(pat)] = np.frombuffer(pat, dtype=np.uint8)
        for kw in (dict(), dict(count_lines=True, only_match=True), dict(max_count=1)):
            _check(gpu, oracle_engine, text, pat, kw, abi.REF_AVX512)


def test_maximum_pattern_length_and_dense_long_patterns(gpu, oracle_engine):
    """MAX_PATTERN_LENGTH = 1024 (krep.c:77); and a long pattern whose every alignment is a candidate."""
    rng = np.random.RandomState(21)
    text = cases.rand_text(rng, 70_000, b"ab")
    for m in (1023, 1024, 513):
        pat = text[5000:5000 + m].tobytes()
        text[40_000:40_000 + m] = np.frombuffer(pat, dtype=np.uint8)
        for kw in (dict(), dict(count_lines=True, only_match=True), dict(case_sensitive=False)):
            _check(gpu, oracle_engine, text, pat, kw, abi.REF_AVX2)
    aaa = np.full(20_000, ord("a"), dtype=np.uint8)
    aaa[7000] = ord("\n")
    for m in (9, 100, 1024):
        for kw in (dict(), dict(count_lines=True), dict(max_count=17)):
            _check(gpu, oracle_engine, aaa, b"a" * m, kw, abi.REF_SCALAR if m < 33 else abi.REF_AVX2)


def test_binary_haystack_all_byte_values(gpu, oracle_engine):
    """Arbitrary bytes (NULs, 0x80-0xFF): nothing may be treated as a terminator, and -i folds ASCII A-Z only
    (lower_table is built in the C locale, krep.c:125-134)."""
    rng = np.random.RandomState(33)
    text = rng.randint(0, 256, size=120_000).astype(np.uint8)
    text[rng.randint(0, text.size, 3000)] = 0
    for m in (1, 2, 3, 4, 8, 13, 16):
        for _ in range(3):
            s = rng.randint(0, text.size - m)
            pat = text[s:s + m].tobytes()
            for t in rng.randint(0, text.size - m, 4):
                text[t:t + m] = np.frombuffer(pat, dtype=np.uint8)
            for kw in (dict(), dict(case_sensitive=False), dict(whole_word=True), dict(count_lines=True)):
                _check(gpu, oracle_engine, text, pat, kw, abi.REF_AVX2)
    hi = np.frombuffer(bytes([0xC1, 0xE1, ord("A"), ord("a"), 0x41 + 0x80]) * 2000, dtype=np.uint8)
    _check(gpu, oracle_engine, hi, bytes([0xE1, ord("a")]), dict(case_sensitive=False), abi.REF_SCALAR)
    _check(gpu, oracle_engine, hi, b"A", dict(case_sensitive=False), abi.REF_SCALAR)


@pytest.mark.parametrize("m", [9, 11, 12, 15, 16, 17, 23, 32, 33, 47, 63, 64, 65])
def test_nine_to_sixteen_byte_verify_in_registers(gpu, oracle_engine, m):
    """m = 9..16: bytes 8..15 are compared in registers against the lane's and the next lane's data (next cell /
    the 16 bytes behind the round for lane 63).  Occurrences and near misses (same first 8 bytes, one of the later
    bytes changed) straddling every lane, cell, round and unit boundary, with and without -i."""
    rng = np.random.RandomState(1000 + m)
    pat = (b"Sherlock" + b"HolmesXY" + b"and-the-Hound_of_the-Baskervilles+0123456789abcdef")[:m]  # > 16: chunked loads
    n = 3 * 32768 + 777
    text = cases.rand_text(rng, n, b"abcdefgh \n")
    spots = []
    for edge in (16, 1024, 8192, 32768, 65536, 98304, n):
        for d in range(-m - 3, 3):
            s = edge + d
            if 0 <= s and s + m <= n:
                spots.append(s)
    spots = sorted(set(spots))
    keep, last = [], -10**9
    for s in spots:                      # non-overlapping plants, alternating hit / near miss
        if s >= last + m:
            keep.append(s)
            last = s
    for i, s in enumerate(keep):
        p = bytearray(pat)
        if i % 3 == 1:
            p[8 + (i // 3) % (m - 8)] ^= 0x01        # near miss in the verified tail
        if i % 3 == 2:
            p = bytearray(bytes(p).swapcase())        # case variant: a hit only with -i
        text[s:s + m] = np.frombuffer(bytes(p), dtype=np.uint8)
    for cs in (True, False):
        for kw in (dict(), dict(count_lines=True), dict(whole_word=True)):
            _check(gpu, oracle_engine, text, pat, dict(case_sensitive=cs, **kw), abi.REF_SCALAR)


def test_the_parity_checker_is_the_compiled_reference(gpu, oracle_engine):
    """the -m gpu tests compared the HIP path with the RESTATEMENT.  They now ask the unmodified reference
    compiled into oracle/_ref, function by function (tests/oracle_lib.py: Checker); the restatement only answers under -o
    and where a build cannot run on this host."""
    text = cases.rand_text(np.random.RandomState(5), 60_000, b"abcd \n")
    jobs = [(abi.REF_SCALAR, b"cab", {}), (abi.REF_SCALAR, b"abab", {}), (abi.REF_SSE42, b"d", {}),
            (abi.REF_SSE42, b"ab", dict(case_sensitive=False)), (abi.REF_SSE42, b"abcd", {}),
            (abi.REF_AVX2, b"abcd abcd abcd abcd ", {}), (abi.REF_AVX512, b"abcd " * 9, {}), (abi.REF_NEON, b"dab", {})]
    before = oracle_engine.direct_calls
    for level, pat, kw in jobs:
        text[1000:1000 + len(pat)] = np.frombuffer(pat, dtype=np.uint8)
        _check(gpu, oracle_engine, text, pat, kw, level)
    want = {f"ref:{ol._REF_FILES[lv]}" for lv in ol._REF_FILES if ol.ref_available(lv)}
    assert oracle_engine.direct_calls - before >= len([lv for lv, _, _ in jobs if ol.ref_available(lv)])
    assert want <= oracle_engine.used, (want, oracle_engine.used)


@pytest.mark.timeout(240)
def test_single_byte_one_pass_with_a_starved_grid(gpu):
    """the one-pass single-byte kernel must make progress when only a few of its waves really run (a
    shared or partitioned device).  Grids of 1, 2 and 3 workgroups over a 1 GiB text (8 192 tickets of 128 KiB; the round-3
    kernel waited circularly here until its multi-second safety net fired and the plan fell back to the two-pass kernels):
    the exact record list, no hand-over to the two-pass kernels, and in seconds."""
    import time
    import torch
    n = 1 << 30
    buf = torch.empty(n + 64, dtype=torch.uint8, device="cuda")
    gpu.generate(buf.data_ptr(), n, 0, 3, 20260925, b"#", 0)
    want = torch.nonzero(buf[:n] == ord("#")).flatten()
    cap = int(want.numel()) + 4096
    pos = torch.empty(2 * cap, dtype=torch.int64, device="cuda")
    before = gpu.single_failovers()
    try:
        for blocks in (1, 2, 3, 0):
            gpu.force_single_grid(blocks)
            pos.zero_()
            plan = gpu.plan(abi.Params([b"#"]))
            t0 = time.time()
            out = plan.scan(buf.data_ptr(), n, 0, n, 0, pos.data_ptr(), cap)
            dt = time.time() - t0
            assert out.count == out.stored == int(want.numel()) and not out.overflow, blocks
            assert torch.equal(pos[: 2 * out.stored].view(-1, 2)[:, 0], want), blocks
            assert gpu.single_failovers() == before, f"grid of {blocks} blocks handed over to the two-pass kernels"
            assert dt < 30, (blocks, dt)
            plan.close()
    finally:
        gpu.force_single_grid(0)


def test_single_byte_one_pass_dense_shapes(gpu, oracle_engine):
    """The one-pass single-byte kernel beyond ~1.5 % hits (kg_single.hip: 64- and 32-KiB tickets with 16-KiB rings, chosen from
    the density the first scan counted): 2.5 %, 8 %, 14 % and — too dense for any ring, the two-pass kernels — 33 % of the bytes;
    a new plan and a re-used one, case-insensitive too.  memchr_search, /root/reference/krep.c:3891-4041."""
    import torch
    rng = np.random.RandomState(77)
    n = 5 * (1 << 20) + 333
    gpu.force_rounds(4)  # (the large-text tile shape on a small text)
    try:
        for alpha, expect_two_pass in ((b"e" + bytes(range(65, 65 + 39)), False), (b"e" + b"abcdfghijkl", False), (b"e" + b"abcdfg", False),
                                       (b"eab", None)):  # (None: whether a 5-MiB text at 33 % overflows a ring depends on how the tickets fall)
            text = cases.rand_text(rng, n, alpha)
            text[::4001] = ord("E")
            for kw in (dict(), dict(case_sensitive=False)):
                p = abi.Params([b"e"], **kw)
                want = oracle_engine.call(gpu.mirror_select(p, n), abi.Params([b"e"], **kw), text)
                before = gpu.single_failovers()
                got = gpu.search(p, text)
                assert got[0] == want[0] and np.array_equal(got[1], want[1]), (alpha[:4], kw, got[0], want[0])
                if expect_two_pass is not None:
                    assert (gpu.single_failovers() > before) == expect_two_pass, (alpha[:4], kw)
            d = torch.from_numpy(text).cuda()
            pw = abi.Params([b"e"])
            want = oracle_engine.call(gpu.mirror_select(pw, n), abi.Params([b"e"]), text)
            plan = gpu.plan(abi.Params([b"e"]))
            cap = int(want[0]) + 7
            pos = torch.zeros(2 * cap, dtype=torch.int64, device="cuda")
            for rep in range(3):  # the second and third scan start in the shape the first one chose
                pos.zero_()
                out = plan.scan(d.data_ptr(), n, 0, n, 0, pos.data_ptr(), cap)
                assert out.count == want[0]
                assert np.array_equal(pos[:2 * out.count].cpu().numpy().astype(np.uint64).reshape(-1, 2), want[1]), (alpha[:4], rep)
            # ... and on a starved grid (the shapes with 32-KiB tickets draw once per WORKGROUP behind a barrier; the resolver's own
            # workgroup draws per wave: a single resident workgroup must still get through)
            for blocks in (1, 2, 3):
                gpu.force_single_grid(blocks)
                try:
                    pos.zero_()
                    out = plan.scan(d.data_ptr(), n, 0, n, 0, pos.data_ptr(), cap)
                finally:
                    gpu.force_single_grid(0)
                assert out.count == want[0], (alpha[:4], blocks)
                assert np.array_equal(pos[:2 * out.count].cpu().numpy().astype(np.uint64).reshape(-1, 2), want[1]), (alpha[:4], blocks)
            plan.close()
    finally:
        gpu.force_rounds(0)


def test_one_plan_across_texts_of_changing_density(gpu, oracle_engine):
    """What a plan has learnt about density is re-evaluated by every scan (ring shape of the one-pass kernel, the
    two-pass fallback, the byte-set dictionary, the dense road of a tiny dictionary were one-way): ONE plan scans a sparse text,
    a text too dense for any ring, a medium one and the sparse one again — every list exact each time, single byte and byte set."""
    import torch
    rng = np.random.RandomState(4242)
    n = 5 * (1 << 20) + 77
    gpu.force_rounds(4)
    try:
        texts = [cases.rand_text(rng, n, alpha) for alpha in (b"e" + bytes(range(65, 91)) * 30, b"eta", b"et" + b"abcdfghijk" * 2,
                                                            b"e" + bytes(range(65, 91)) * 30)]
        for pats in ([b"e"], [b"e", b"t"], [b"he", b"e"]):
            algo = abi.RA_AHO_CORASICK if len(pats) > 1 else gpu.mirror_select(abi.Params(pats), n)
            plan = gpu.plan(abi.Params(pats))
            for rep in range(2):
                for ti, text in enumerate(texts):
                    want = oracle_engine.call(algo, abi.Params(pats), text)
                    d = torch.from_numpy(text).cuda()
                    cap = int(want[0]) + 5
                    pos = torch.zeros(2 * cap, dtype=torch.int64, device="cuda")
                    out = plan.scan(d.data_ptr(), n, 0, n, 0, pos.data_ptr(), cap)
                    assert out.count == want[0] and not out.overflow, (pats, rep, ti, out.count, want[0])
                    got = pos[:2 * out.count].cpu().numpy().astype(np.uint64).reshape(-1, 2)
                    assert np.array_equal(got, want[1]), (pats, rep, ti)
                    del d, pos
            plan.close()
    finally:
        gpu.force_rounds(0)


def test_dense_short_literals_take_the_one_pass_record_writer(gpu, oracle_engine):
    """A 2..8-byte literal with records on a DENSE text (more than ~24 hits per 32-KiB unit: the staging slots of the sparse kinds
    overflow there): the first scan of a plan counts the density through the two-pass kernels, the following ones write their
    records in one pass (kg_single.hip, MULTI) — every list exact, case-insensitive too, with hits in the last bytes of the
    text and straddling rounds and cells; a later sparse text sends the plan back to the staging road.
    The all-occurrence functions, /root/reference/krep.c:3421-3530 (BMH), :3891-4041 (memchr_short for m <= 3)."""
    import torch
    rng = np.random.RandomState(909)
    n = 5 * (1 << 20) + 1234
    gpu.force_rounds(4)
    try:
        for pat in (b"ab", b"abc", b"xy z", b"hello", b"abcabd", b"Sherloc", b"a1b2c3d4"):
            # dense: the pattern, its prefixes and fillers as tokens in random order — one token in ~30 is the pattern
            toks = [pat] + [pat[:-1], pat[1:], b" ", b"q", pat[:1] * 2] * 5
            dense = np.frombuffer(b"".join(toks[i] for i in rng.randint(0, len(toks), n))[:n], dtype=np.uint8).copy()
            assert len(dense) == n
            alpha = pat
            # occurrences planted across every kind of boundary: cells (1 KiB), rounds (8 KiB), units (32 KiB), the end of the text
            for at in (1024 - 1, 8192 - 3, 32768 - 2, 131072 - 1, 3 * 131072 - 4, n - len(pat), n - len(pat) - 1):
                dense[at:at + len(pat)] = np.frombuffer(pat, dtype=np.uint8)
            sparse = cases.rand_text(rng, n, bytes(range(65, 91)) * 4 + alpha)
            for kw in (dict(), dict(case_sensitive=False)):
                if kw:
                    up = np.frombuffer(pat.upper(), dtype=np.uint8)
                    dense[40000:40000 + len(pat)] = up
                p = abi.Params([pat], **kw)
                algo = gpu.mirror_select(p, n)
                fam_all = oracle_engine.call(algo, abi.Params([pat], **kw), dense)
                plan = gpu.plan(abi.Params([pat], **kw))
                launches = []
                for ti, text in enumerate((dense, dense, dense, sparse, dense, dense)):
                    want = fam_all if text is dense else oracle_engine.call(algo, abi.Params([pat], **kw), text)
                    d = torch.from_numpy(text).cuda()
                    cap = int(want[0]) + 3
                    pos = torch.zeros(2 * cap, dtype=torch.int64, device="cuda")
                    before = gpu.single_launches()
                    out = plan.scan(d.data_ptr(), n, 0, n, 0, pos.data_ptr(), cap)
                    launches.append(gpu.single_launches() - before)
                    assert out.count == want[0] and not out.overflow, (pat, kw, ti, out.count, want[0])
                    got = pos[:2 * out.count].cpu().numpy().astype(np.uint64).reshape(-1, 2)
                    assert np.array_equal(got, want[1]), (pat, kw, ti)
                    del d, pos
                if plan_is_all_occurrence(gpu, p, n):
                    # ownership windows with a global base on the plan that now writes in one pass: a window owns the matches
                    # that START in it, the windows' lists concatenate to the whole list
                    d = torch.from_numpy(dense).cuda()
                    cap = int(fam_all[0]) + 3
                    pos = torch.zeros(2 * cap, dtype=torch.int64, device="cuda")
                    parts, before = [], gpu.single_launches()
                    cuts = [0, 5, (1 << 20) + 17, (3 << 20) + 16383, n - 1, n]
                    for lo, hi in zip(cuts[:-1], cuts[1:]):
                        out = plan.scan(d.data_ptr(), n, lo, hi, 777, pos.data_ptr(), cap)
                        parts.append(pos[:2 * out.stored].cpu().numpy().astype(np.uint64).reshape(-1, 2) - 777)
                    assert np.array_equal(np.concatenate(parts), fam_all[1]), (pat, kw, "windows")
                    assert gpu.single_launches() > before, (pat, kw, "windows on the two-pass road")
                    del d, pos
                plan.close()
                if plan_is_all_occurrence(gpu, p, n):
                    # scan 0 learns, 1 and 2 take the one-pass kernel; the sparse text (3) takes it once more and switches it off,
                    # so 4 learns again and 5 is one pass
                    assert launches[0] == 0 and launches[1] >= 1 and launches[2] >= 1 and launches[5] >= 1, (pat, kw, launches)
    finally:
        gpu.force_rounds(0)


def plan_is_all_occurrence(gpu, p, n):
    """the literal kernels with a RECORDS sink serve the all-occurrence functions directly; greedy families go through the walk"""
    return gpu.mirror_select(p, n) in (abi.RA_BMH, abi.RA_MEMCHR_SHORT, abi.RA_MEMCHR)


def test_dense_short_literals_whole_word_in_one_pass(gpu, oracle_engine):
    """-w on the one-pass record writer and in the two-pass kernels (round 6): the neighbours of a start position are taken from the
    lane's registers — the 24-byte window, the lane below's last dword — not from memory.  Texts made of the pattern as a word, the
    pattern inside words, and every neighbour class is_whole_word_match (krep.h:312-319) tells apart (letters, digits, '_', blank,
    punctuation, bytes >= 0x80), with occurrences on every seam: lane 0 of a cell (the byte in front comes from the lane above the
    seam or, for lane 0 of a cell, from memory), cells, rounds, units, the first and the last bytes of the text."""
    import torch
    rng = np.random.RandomState(4242)
    n = 5 * (1 << 20) + 777
    gpu.force_rounds(4)
    try:
        for pat in (b"the", b"ab", b"word", b"hello", b"Sherloc", b"a1b2c3d4"):
            m = len(pat)
            toks = [pat, pat, pat + b"s", b"x" + pat, b"_" + pat, pat + b"9", b"\xc3" + pat, pat + b"\xa9", b" ", b" ", b"\n", b",", b".", b"-", b"(", b")",
                    b"qq", b" "] * 2
            dense = np.frombuffer(b"".join(toks[i] for i in rng.randint(0, len(toks), n))[:n], dtype=np.uint8).copy()
            assert len(dense) == n
            for at in (0, 16, 1024, 1024 - m, 8192, 8192 - 1, 32768, 32768 - m + 1, 131072, 3 * 131072 - 2, n - m, n - m - 1):
                dense[at:at + m] = np.frombuffer(pat, dtype=np.uint8)
                if at:
                    dense[at - 1] = rng.choice(np.frombuffer(b" a_7.\xe9", dtype=np.uint8))
                if at + m < n:
                    dense[at + m] = rng.choice(np.frombuffer(b" z_0,\x80", dtype=np.uint8))
            sparse = cases.rand_text(rng, n, bytes(range(65, 91)) * 4 + b"  ")
            for kw in (dict(whole_word=True), dict(whole_word=True, case_sensitive=False)):
                p = abi.Params([pat], **kw)
                algo = gpu.mirror_select(p, n)
                want_dense = oracle_engine.call(algo, abi.Params([pat], **kw), dense)
                assert want_dense[0] > n // 400, (pat, kw, want_dense[0])
                plan = gpu.plan(abi.Params([pat], **kw))
                launches = []
                for ti, text in enumerate((dense, dense, dense, sparse, dense, dense)):
                    want = want_dense if text is dense else oracle_engine.call(algo, abi.Params([pat], **kw), text)
                    d = torch.from_numpy(text).cuda()
                    cap = int(want[0]) + 3
                    pos = torch.zeros(2 * cap, dtype=torch.int64, device="cuda")
                    before = gpu.single_launches()
                    out = plan.scan(d.data_ptr(), n, 0, n, 0, pos.data_ptr(), cap)
                    launches.append(gpu.single_launches() - before)
                    assert out.count == want[0] and not out.overflow, (pat, kw, ti, out.count, want[0])
                    got = pos[:2 * out.count].cpu().numpy().astype(np.uint64).reshape(-1, 2)
                    assert np.array_equal(got, want[1]), (pat, kw, ti)
                    # counting only (the two-pass kernels without their stores) agrees
                    assert plan.scan(d.data_ptr(), n).count == want[0]
                    del d, pos
                plan.close()
                if plan_is_all_occurrence(gpu, p, n):
                    assert launches[0] == 0 and launches[1] >= 1 and launches[5] >= 1, (pat, kw, launches)
    finally:
        gpu.force_rounds(0)
