"""GPU parity for TINY dictionaries (every pattern 1..4 bytes: krep_amd/csrc/kg_ac_tiny.hip) against the reference's
aho_corasick_search: count, every (start, end) record and the emission order (end ascending, longest first —
/root/reference/aho_corasick.c:383-437), -i, -c lines, -c -o, max_count, windows, staging overflow, dense hits."""
import numpy as np
import pytest

import cases
from krep_amd import abi

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    import krep_amd
    e = krep_amd.load()
    assert e.device_count() >= 1
    return e


def _check(gpu, o, text, pats, kw, tiny=True):
    before = gpu.tiny_launches()
    want = o.call(abi.RA_AHO_CORASICK, abi.Params(pats, **kw), text)
    got = gpu.search(abi.Params(pats, **kw), text)
    assert got[0] == want[0], (pats, kw, len(text), got[0], want[0])
    assert np.array_equal(got[1], want[1]), (pats, kw, got[1][:8], want[1][:8])
    if tiny and max(len(p) for p in pats) == 1:
        tiny = None  # (single bytes: with records the one-pass byte-set scan of kg_single.hip, else the register compare)
    if tiny is not None and len(text) and kw.get("max_count", 1) != 0:
        assert (gpu.tiny_launches() > before) == tiny, (pats, kw, "tiny kernel expected" if tiny else "general kernel expected")


def _distinct(rng, text, alpha, lens, k):
    pats = []
    for _ in range(50):
        if len(pats) == k:
            break
        p = cases.pick_pattern(rng, text, lens[rng.randint(0, len(lens))], alpha)
        if p not in pats and sum(len(q) == len(p) for q in pats) < 4:
            pats.append(p)
    return pats


def test_the_textbook_dictionary(gpu, oracle_engine):
    text = np.frombuffer(b"ushers and she said: hers, not his; he hesitated.\nSHE heard HERS\n" * 3000, dtype=np.uint8)
    pats = [b"he", b"she", b"hers", b"his"]
    for kw in (dict(), dict(case_sensitive=False), dict(count_lines=True), dict(count_lines=True, only_match=True),
               dict(max_count=5), dict(case_sensitive=False, count_lines=True)):
        _check(gpu, oracle_engine, text, pats, kw)
    _check(gpu, oracle_engine, text, pats, dict(whole_word=True), tiny=False)  # -w stays on the general kernel


@pytest.mark.parametrize("seed", range(6))
def test_random_tiny_dictionaries(gpu, oracle_engine, seed):
    rng = np.random.RandomState(900 + seed)
    for it in range(30):
        alpha = [b"ab", b"abc\n", b"abAB -\n", bytes(range(97, 105)) + b" \n", b"\x00\x01a\n"][it % 5]
        n = [1, 3, 4, 17, 500, 8191, 8192, 8195, 16384, 16389, 40000, 140000, 300007][rng.randint(0, 13)]
        text = cases.rand_text(rng, n, alpha)
        lens = [[1], [2], [1, 2, 3, 4], [2, 3, 4], [2, 4], [1, 4], [1, 3], [1, 5], [2, 3, 8], [1, 2, 6], [2, 7, 7]][rng.randint(0, 11)]
        pats = _distinct(rng, text, alpha, lens, [2, 2, 3, 5, 8][rng.randint(0, 5)])
        if len(pats) < 2 or min(len(p) for p in pats) > 2:
            continue  # (one pattern is the literal scan's business; 3- and 4-byte patterns only: the general kernel's)
        kw = dict(case_sensitive=bool(rng.rand() < 0.6), max_count=([abi.SIZE_MAX] * 4 + [1, 4, 77])[rng.randint(0, 7)])
        mode = ["pos", "pos", "lines", "count"][rng.randint(0, 4)]
        if not kw["case_sensitive"] and len({p.lower() for p in pats}) != len(pats):
            continue  # duplicates after folding: the general kernel (covered by test_gpu_ac.py)
        if mode == "lines":
            if any(b"\n" in p for p in pats):
                continue
            kw.update(count_lines=True)
        elif mode == "count":
            kw.update(count_lines=True, only_match=True)
        _check(gpu, oracle_engine, text, pats, kw)


def test_dense_hits_overflow_every_staging_slot(gpu, oracle_engine):
    rng = np.random.RandomState(7)
    text = cases.rand_text(rng, 1 << 20, b"etaoin shrdlu\n")
    for pats in ([b"e", b"t"], [b"e", b"th", b"t", b"he"], [b"a", b"ao", b"tao", b"etao"]):  # (0.5 to 0.15 matches per byte)
        _check(gpu, oracle_engine, text, pats, dict())
        _check(gpu, oracle_engine, text, pats, dict(count_lines=True))
        _check(gpu, oracle_engine, text, pats, dict(count_lines=True, only_match=True))
    try:
        gpu.force_stage_cap(4)
        _check(gpu, oracle_engine, text[:200_000], [b"sh", b"rd", b"lu\n"], dict())
        _check(gpu, oracle_engine, text[:200_000], [b"s", b"rd", b"dlu"], dict(case_sensitive=False))
    finally:
        gpu.force_stage_cap(0)


def test_windows_and_shards_own_every_match_once(gpu, oracle_engine):
    """search_buffer over N logical shards (start-offset ownership, /root/reference/krep.c:2729-2770): every match once."""
    rng = np.random.RandomState(11)
    text = cases.rand_text(rng, 200_003, b"abc \n")
    pats = [b"a", b"ab", b"cab", b"abca"]
    p = abi.Params(pats)
    want = oracle_engine.call(abi.RA_AHO_CORASICK, p, text)
    for shards in (2, 3, 7):
        before = gpu.tiny_launches()
        rc, n, pos = gpu.search_buffer(p, text, num_gpus=shards)
        assert rc == 0 and n == want[0], (shards, rc, n, want[0])
        assert np.array_equal(pos, want[1]), shards
        assert gpu.tiny_launches() > before
    pl = abi.Params(pats, count_lines=True)
    wl = oracle_engine.call(abi.RA_AHO_CORASICK, pl, text)
    for shards in (2, 5):
        rc, n, _ = gpu.search_buffer(pl, text, num_gpus=shards)
        assert rc == 0 and n == wl[0], (shards, n, wl[0])


def test_nul_bytes_never_match_in_front_of_the_text(gpu, oracle_engine):
    text = np.frombuffer(b"\x00\x00a\x00\x00\x00b" + b"\x00" * 40 + b"a\x00", dtype=np.uint8)
    for pats in ([b"\x00", b"b"], [b"\x00\x00", b"b\x00"], [b"\x00\x00\x00\x00", b"\x00a"], [b"\x00\x00\x00", b"a\x00"], [b"\x00\x00\x00\x00", b"\x00"]):
        _check(gpu, oracle_engine, text, pats, dict())
        _check(gpu, oracle_engine, text, pats, dict(count_lines=True, only_match=True))


def test_general_kernel_when_the_dictionary_does_not_qualify(gpu, oracle_engine):
    rng = np.random.RandomState(3)
    text = cases.rand_text(rng, 50_000, b"abcd \n")
    _check(gpu, oracle_engine, text, [b"ab", b"abcd", b"abcda"], dict(), tiny=False)              # two lengths beyond 3 bytes
    _check(gpu, oracle_engine, text, [b"ab", b"abcdabcda"], dict(), tiny=False)                   # a 9-byte pattern
    _check(gpu, oracle_engine, text, [b"a", b"b", b"c", b"d", b" "], dict(), tiny=False)          # five of one length
    _check(gpu, oracle_engine, text, [b"ab", b"ab"], dict(), tiny=False)                          # a duplicate
    _check(gpu, oracle_engine, text, [b"abc", b"bcd", b"cdab"], dict(), tiny=False)               # nothing shorter than 3 bytes


def test_a_plan_learns_that_its_dictionary_is_dense(gpu, oracle_engine):
    """The second and later scans of a plan whose units all overflowed their staging slots run as a count pass plus an
    emit-mode pass that writes every record (kg_ac.hip, ac_scan): same count, same records, same order, every time — also
    through ownership windows (/root/reference/krep.c:2729-2770) and with a list capacity below the count."""
    import torch
    rng = np.random.RandomState(21)
    text = cases.rand_text(rng, (1 << 20) + 12345, b"etaoin shrdlu\n")
    d = torch.from_numpy(text).cuda()
    n = text.size
    for pats, kw in (([b"e", b"t"], dict()), ([b"t", b"ao", b"in "], dict(case_sensitive=False))):
        want = oracle_engine.call(abi.RA_AHO_CORASICK, abi.Params(pats, **kw), text)
        plan = gpu.plan(abi.Params(pats, **kw))
        cap = int(want[0]) + 5
        pos = torch.zeros(2 * cap, dtype=torch.int64, device="cuda")
        for rep in range(4):
            pos.zero_()
            out = plan.scan(d.data_ptr(), n, 0, n, 0, pos.data_ptr(), cap)
            assert out.count == want[0], (pats, rep, out.count, want[0])
            got = pos[:2 * out.count].cpu().numpy().astype(np.uint64).reshape(-1, 2)
            assert np.array_equal(got, want[1]), (pats, rep)
        # windows: three pieces owned by start offset concatenate to the whole list
        cuts = [0, 300_001, 700_007, n]
        acc = []
        for lo, hi in zip(cuts[:-1], cuts[1:]):
            pos.zero_()
            out = plan.scan(d.data_ptr(), n, lo, hi, 0, pos.data_ptr(), cap)
            acc.append(pos[:2 * out.count].cpu().numpy().astype(np.uint64).reshape(-1, 2))
        assert np.array_equal(np.concatenate(acc), want[1]), pats
        # a list shorter than the count: the first `small` records, the full count
        small = 1000
        pos.zero_()
        out = plan.scan(d.data_ptr(), n, 0, n, 0, pos.data_ptr(), small)
        assert out.count == want[0] and out.overflow
        assert np.array_equal(pos[:2 * small].cpu().numpy().astype(np.uint64).reshape(-1, 2), want[1][:small]), pats
        plan.close()


def test_a_dictionary_of_single_bytes_is_the_one_pass_byte_scan_with_a_set(gpu, oracle_engine):
    """`-e e -e t` with records runs in kg_single.hip (needle set, records at their final index, rings sized by the counted
    density); without records, under -w, under -c or denser than its rings it stays where it was.  Order and content of the
    list: aho_corasick_search, /root/reference/aho_corasick.c:383-437."""
    import torch
    rng = np.random.RandomState(5)
    n = 6 * (1 << 20) + 1234
    # (alphabet, dictionary, one-pass expected): 7 % / 6 % / 10 % / 13 % of the bytes match; 100 % is beyond the largest rings
    for alpha, pats, one_pass in ((b"etaoin shrdlu\n" * 2 + b"ET", [b"e", b"x"], True), (bytes(range(64, 128)), [b"a", b"B", b"c", b"\x7f"], True),
                                  (b"abcdefghij" * 6 + b"\n", [b"x", b"a"], True), (b"etaoin shrdlu\n" * 2 + b"ET", [b"e", b"t"], True),
                                  (b"ab", [b"a", b"b"], False)):
        text = cases.rand_text(rng, n, alpha)
        for kw in (dict(), dict(case_sensitive=False), dict(max_count=1000)):
            before = gpu.tiny_launches()
            want = oracle_engine.call(abi.RA_AHO_CORASICK, abi.Params(pats, **kw), text)
            got = gpu.search(abi.Params(pats, **kw), text)
            assert got[0] == want[0] and np.array_equal(got[1], want[1]), (pats, kw, got[0], want[0])
            if kw.get("case_sensitive", True):  # (-i doubles the density of the letters: which road it takes is not the point here)
                assert (gpu.tiny_launches() == before) == one_pass, (pats, kw, "one-pass byte-set scan expected" if one_pass else "register compare expected")
        # windows owned by start offset concatenate to the whole list; a re-used plan keeps its shape
        d = torch.from_numpy(text).cuda()
        want = oracle_engine.call(abi.RA_AHO_CORASICK, abi.Params(pats), text)
        plan = gpu.plan(abi.Params(pats))
        cap = int(want[0]) + 3
        pos = torch.zeros(2 * cap, dtype=torch.int64, device="cuda")
        acc = []
        for lo, hi in ((0, 1_000_003), (1_000_003, 4_000_000), (4_000_000, n)):
            pos.zero_()
            out = plan.scan(d.data_ptr(), n, lo, hi, 0, pos.data_ptr(), cap)
            acc.append(pos[:2 * out.count].cpu().numpy().astype(np.uint64).reshape(-1, 2))
        assert np.array_equal(np.concatenate(acc), want[1]), pats
        plan.close()
        _check(gpu, oracle_engine, text[:300_000], pats, dict(count_lines=True, only_match=True), tiny=None)
        _check(gpu, oracle_engine, text[:300_000], pats, dict(count_lines=True), tiny=None if b"\n" not in alpha else True)


def test_one_long_length_beside_short_patterns(gpu, oracle_engine):
    """`-e a -e Sherlock`: ONE length of 5..8 bytes may stand beside 1..3-byte patterns — its last four bytes are compared at the
    END, its first ones a dword earlier (kg_ac_tiny.hip).  All modes, -i, matches across cell / round / unit boundaries and at the
    very start of the text, windows."""
    import torch
    rng = np.random.RandomState(31)
    n = 3 * (1 << 20) + 4321
    text = cases.rand_text(rng, n, b"abcdefghij klmnop\n")
    for pat, spots in ((b"Sherlock", (0, 3, 13, 1017, 1020, 8185, 8190, 16379, 16383, 131070, n - 8)), (b"HoLmes", (40, 1022, 16381, n - 6)),
                       (b"Watso", (77, 8189, n - 5)), (b"Baskerv", (200, 16380, 999_999))):
        for sp in spots:
            text[sp:sp + len(pat)] = np.frombuffer(pat, dtype=np.uint8)
    for pats in ([b"a", b"Sherlock"], [b"e", b"gh", b"Sherlock"], [b"ij", b"HoLmes", b"klmnop"], [b"b", b"cd", b"efg", b"Watso"],
                 [b"a", b"Baskerv", b"abcdefg"]):
        for kw in (dict(), dict(case_sensitive=False), dict(count_lines=True), dict(count_lines=True, only_match=True), dict(max_count=99)):
            _check(gpu, oracle_engine, text, pats, kw)
        _check(gpu, oracle_engine, text[:9], pats, dict())
        _check(gpu, oracle_engine, text[:8], pats, dict())
        want = oracle_engine.call(abi.RA_AHO_CORASICK, abi.Params(pats), text)
        for shards in (2, 5):
            rc, cnt, pos = gpu.search_buffer(abi.Params(pats), text, num_gpus=shards)
            assert rc == 0 and cnt == want[0] and np.array_equal(pos, want[1]), (pats, shards)
        d = torch.from_numpy(text).cuda()
        plan = gpu.plan(abi.Params(pats))
        cap = int(want[0]) + 3
        pos = torch.zeros(2 * cap, dtype=torch.int64, device="cuda")
        for rep in range(3):  # (a dense dictionary: the later scans count first and emit every record)
            pos.zero_()
            out = plan.scan(d.data_ptr(), n, 0, n, 0, pos.data_ptr(), cap)
            assert out.count == want[0]
            assert np.array_equal(pos[:2 * out.count].cpu().numpy().astype(np.uint64).reshape(-1, 2), want[1]), (pats, rep)
        plan.close()


def test_one_pass_records_of_tiny_dictionaries(gpu, oracle_engine, monkeypatch):
    """Round 5: a tiny dictionary's records in ONE pass (kg_ac_tiny.hip FUSED: matches ranked into an LDS ring per 128-KiB ticket,
    the tickets' counts resolved into prefixes by one wave, records written at their final index — kg_tickets.h).  Texts of
    1-6 MiB (the road opens at 1 MiB), the complete list in the reference's order (END ascending, longest first,
    aho_corasick.c:383-437), -i, max_count, ownership windows with a global base, a text too dense for the rings (counted, not
    recorded: the staging road takes it — and the plan goes back to one pass on a sparse text), a starved grid of 1-3 workgroups,
    and the same answers with the road switched off."""
    import torch
    rng = np.random.RandomState(31337)
    n = 6 * (1 << 20) + 1234
    sparse = cases.rand_text(rng, n, bytes(range(97, 123)) + b"    \n")
    sparse[rng.randint(0, n - 8, 4000)] = ord("h")
    dense = cases.rand_text(rng, n, b"hes r\n")
    nested = np.frombuffer((b"ushers and she said hers " * 40 + b"\n") * (n // 1001), dtype=np.uint8).copy()
    for text, tag in ((sparse, "sparse"), (nested[: 3 << 20], "nested"), (dense, "dense"), (sparse[: (1 << 20) + 5], "1 MiB")):
        tn = text.size
        for pats in ([b"he", b"she", b"hers"], [b"h", b"er", b"s s"], [b"xq", b"zj", b"he"], [b"e", b"he", b"she", b"shes"]):
            for kw in (dict(), dict(case_sensitive=False), dict(max_count=777)):
                want = oracle_engine.call(abi.RA_AHO_CORASICK, abi.Params(pats, **kw), text)
                got = gpu.search(abi.Params(pats, **kw), text)
                assert got[0] == want[0] and np.array_equal(got[1], want[1]), (tag, pats, kw, got[0], want[0])
        # device windows with a global base: start-offset ownership, the records of all windows concatenate to the whole list
        pats = [b"he", b"she", b"hers"]
        want = oracle_engine.call(abi.RA_AHO_CORASICK, abi.Params(pats), text)
        d = torch.from_numpy(text).cuda()
        plan = gpu.plan(abi.Params(pats))
        cap = int(want[0]) + 16
        pos = torch.zeros(2 * cap, dtype=torch.int64, device="cuda")
        cuts = [0, 3, (1 << 20) + 17, (2 << 20) + 1, tn] if tn > (3 << 20) else [0, tn]
        parts = []
        for lo, hi in zip(cuts[:-1], cuts[1:]):
            out = plan.scan(d.data_ptr(), tn, lo, hi, 1000, pos.data_ptr(), cap)
            parts.append(pos[: 2 * out.stored].cpu().numpy().astype(np.uint64).reshape(-1, 2) - 1000)
        allp = np.concatenate(parts)
        order = np.lexsort((allp[:, 0], allp[:, 1]))  # windows own by START: merge into (end, start) order
        assert np.array_equal(allp[order], want[1]), (tag, "windows")
        # one plan: dense (falls back) then sparse (one pass again) — exact each time
        plan.close()
        del d, pos
    plan = gpu.plan(abi.Params([b"he", b"she", b"hers"]))
    for text in (dense, sparse, dense[: 2 << 20], sparse):
        want = oracle_engine.call(abi.RA_AHO_CORASICK, abi.Params([b"he", b"she", b"hers"]), text)
        d = torch.from_numpy(text).cuda()
        cap = int(want[0]) + 16
        pos = torch.zeros(2 * cap, dtype=torch.int64, device="cuda")
        out = plan.scan(d.data_ptr(), text.size, 0, text.size, 0, pos.data_ptr(), cap)
        assert out.count == want[0] and np.array_equal(pos[: 2 * out.stored].cpu().numpy().astype(np.uint64).reshape(-1, 2), want[1])
        del d, pos
    plan.close()
    # a starved grid: 1, 2, 3 workgroups over ~48 tickets each (no circular wait whatever part of the grid runs)
    want = oracle_engine.call(abi.RA_AHO_CORASICK, abi.Params([b"he", b"she", b"hers"]), sparse)
    for blocks in (1, 2, 3):
        gpu.force_single_grid(blocks)
        try:
            got = gpu.search(abi.Params([b"he", b"she", b"hers"]), sparse)
        finally:
            gpu.force_single_grid(0)
        assert got[0] == want[0] and np.array_equal(got[1], want[1]), blocks
    # and the road switched off gives the same
    monkeypatch.setenv("KREP_GPU_AC_NO_TINY_FUSED", "1")
    got = gpu.search(abi.Params([b"he", b"she", b"hers"]), sparse)
    assert got[0] == want[0] and np.array_equal(got[1], want[1])


@pytest.mark.parametrize("seed", range(4))
def test_one_pass_records_random_dictionaries(gpu, oracle_engine, seed):
    """Random tiny dictionaries (1-4-byte patterns, nested and overlapping ones included) on 2-3 MiB texts over alphabets of 3 to 30
    symbols: from a match in every lane-cell (the rings overflow, the staging road takes over) down to a handful per ticket — the
    complete list against the compiled reference each time."""
    rng = np.random.RandomState(7700 + seed)
    for it in range(8):
        alpha = [b"abc", b"abcdefgh \n", bytes(range(97, 123)) + b"   \n", b"abAB -\n"][it % 4]
        n = (2 << 20) + int(rng.randint(0, 1 << 20))
        text = cases.rand_text(rng, n, alpha)
        lens = [[1, 2, 3, 4], [2, 3, 4], [2, 4], [1, 4], [2, 3], [3, 4]][rng.randint(0, 6)]
        pats = _distinct(rng, text, alpha, lens, int(rng.randint(2, 9)))
        if len(pats) < 2 or min(len(p) for p in pats) > 2:
            pats.append(text[5:7].tobytes())  # (a tiny dictionary needs a 1- or 2-byte pattern)
            pats = list(dict.fromkeys(pats))
        kw = dict(case_sensitive=bool(rng.rand() < 0.6), max_count=[abi.SIZE_MAX, abi.SIZE_MAX, 5000][rng.randint(0, 3)])
        want = oracle_engine.call(abi.RA_AHO_CORASICK, abi.Params(pats, **kw), text)
        got = gpu.search(abi.Params(pats, **kw), text)
        assert got[0] == want[0] and np.array_equal(got[1], want[1]), (seed, it, pats, kw, got[0], want[0])


def test_four_byte_patterns_beside_a_long_length(gpu, oracle_engine, monkeypatch):
    """Round 5: `-e if -e else -e while` — a 2-byte pattern, 4-byte patterns AND one longer length: the long
    patterns are a FIFTH class of the register-compare kernel, used for the case-sensitive count (every other mode of such a
    dictionary runs in the general kernel, which measured faster for them).  Counting, the complete list (END ascending, longest first — `else`
    inside `elsewhile`, `he` / `here` / `where` ending together), -i, max_count, -c, -w, small and large texts, windows."""
    import torch
    rng = np.random.RandomState(1234)
    n = 3 * (1 << 20) + 99
    text = cases.rand_text(rng, n, b"ehilsw ;\n")
    words = np.frombuffer(b"if else while elsewhile where here he ifelse", dtype=np.uint8)
    for s0 in rng.randint(0, n - 64, 3000):
        text[s0:s0 + words.size] = words
    for pats in ([b"if", b"else", b"while"], [b"he", b"here", b"where", b"e"], [b"se", b"else", b"elsew", b"while", b"ilsew"],
                 [b"e", b"hile", b"elsewhil", b"sewhilee"]):
        for kw in (dict(), dict(case_sensitive=False), dict(max_count=1000), dict(count_lines=True, only_match=True),
                   dict(count_lines=True), dict(whole_word=True)):
            for t in (text, text[:70000], text[: (1 << 20) + 3]):
                want = oracle_engine.call(abi.RA_AHO_CORASICK, abi.Params(pats, **kw), t)
                got = gpu.search(abi.Params(pats, **kw), t)
                assert got[0] == want[0] and np.array_equal(got[1], want[1]), (pats, kw, t.size, got[0], want[0])
        # the register-compare kernel really took the (case-sensitive) count; the list stays in the general kernel (measured faster)
        before = gpu.tiny_launches()
        gpu.search(abi.Params(pats, count_lines=True, only_match=True), text, want_result=False)
        assert gpu.tiny_launches() >= before + 1, pats
        d = torch.from_numpy(text).cuda()
        want = oracle_engine.call(abi.RA_AHO_CORASICK, abi.Params(pats), text)
        plan = gpu.plan(abi.Params(pats))
        cap = int(want[0]) + 16
        pos = torch.zeros(2 * cap, dtype=torch.int64, device="cuda")
        parts = []
        for lo, hi in ((0, 5), (5, (1 << 20) + 1), ((1 << 20) + 1, n)):
            out = plan.scan(d.data_ptr(), n, lo, hi, 0, pos.data_ptr(), cap)
            parts.append(pos[: 2 * out.stored].cpu().numpy().astype(np.uint64).reshape(-1, 2))
        allp = np.concatenate(parts)
        order = np.lexsort((allp[:, 0], allp[:, 1]))
        assert np.array_equal(allp[order], want[1]), (pats, "windows")
        plan.close()
        del d, pos


def test_one_pass_records_on_dense_texts(gpu, oracle_engine):
    """The DENSE flavour of the one-pass record writer (kg_ac_tiny.hip: matches decoded where they are found into 16-bit ring
    entries, tickets of 1..4 units sized by the density a scan counted): dictionaries with and without a long length on texts where
    1-12 % of the bytes end a match.  One plan per dictionary scans a dense text three times (the first scan learns the density —
    through the overflowing item rings or, with a long length, through the staging road — the next ones write in one pass), a
    sparse text (back to the other roads), a very dense one (beyond every ring: the staging road) and the dense one again; then
    ownership windows with a global base, -i, max_count and a starved grid.  Order: aho_corasick.c:383-437."""
    import torch
    rng = np.random.RandomState(4711)
    n = 5 * (1 << 20) + 777
    mid = cases.rand_text(rng, n, b"aSherlock helo\n" + bytes(range(97, 123)))       # ~3 % 'a', words planted below
    for w, k in ((b"Sherlock", 3000), (b"hello", 3000), (b"she", 5000)):
        for at in rng.randint(0, n - 16, k):
            mid[at:at + len(w)] = np.frombuffer(w, dtype=np.uint8)
    for at in (16384 - 3, 16384 * 2 - 1, 65536 - 4, 65536 * 3 - 7, n - 8, n - 5):  # across units, tickets and at the end
        mid[at:at + 5] = np.frombuffer(b"hello", dtype=np.uint8)
    mid[n - 8:] = np.frombuffer(b"Sherlock", dtype=np.uint8)
    sparse = cases.rand_text(rng, n, bytes(range(98, 123)) * 3 + b"   \n")
    thick = cases.rand_text(rng, n, b"ae")                                            # every second byte
    for pats in ([b"a", b"Sherlock"], [b"e", b"hello"], [b"a", b"e"][:1] + [b"he", b"she"], [b"k", b"lo", b"ello", b"c"],
                 [b"a", b"b", b"c", b"Sherloc"]):
        for kw in (dict(), dict(case_sensitive=False)):
            plan = gpu.plan(abi.Params(pats, **kw))
            seen_dense = 0
            for ti, text in enumerate((mid, mid, mid, sparse, thick, mid, mid)):
                want = oracle_engine.call(abi.RA_AHO_CORASICK, abi.Params(pats, **kw), text)
                d = torch.from_numpy(text).cuda()
                cap = int(want[0]) + 9
                pos = torch.zeros(2 * cap, dtype=torch.int64, device="cuda")
                before = gpu.tiny_dense_launches()
                # (six workgroups: every wave scans a dozen tickets, as on a large text — on 5 MiB with the whole chip a wave gets one
                #  ticket, nothing waits in its ring while it scans, and even the item rings hold this text)
                gpu.force_single_grid(6)
                try:
                    out = plan.scan(d.data_ptr(), n, 0, n, 0, pos.data_ptr(), cap)
                finally:
                    gpu.force_single_grid(0)
                seen_dense += gpu.tiny_dense_launches() - before
                assert out.count == want[0] and not out.overflow, (pats, kw, ti, out.count, want[0])
                got = pos[: 2 * out.stored].cpu().numpy().astype(np.uint64).reshape(-1, 2)
                assert np.array_equal(got, want[1]), (pats, kw, ti)
                del d, pos
            assert seen_dense >= 3, (pats, kw, seen_dense)
            # windows with a global base on the plan that now knows the density: start-offset ownership, merged by (end, start)
            want = oracle_engine.call(abi.RA_AHO_CORASICK, abi.Params(pats, **kw), mid)
            d = torch.from_numpy(mid).cuda()
            cap = int(want[0]) + 16
            pos = torch.zeros(2 * cap, dtype=torch.int64, device="cuda")
            parts = []
            cuts = [0, 5, (1 << 20) + 17, (3 << 20) + 16383, n]
            for lo, hi in zip(cuts[:-1], cuts[1:]):
                out = plan.scan(d.data_ptr(), n, lo, hi, 12345, pos.data_ptr(), cap)
                parts.append(pos[: 2 * out.stored].cpu().numpy().astype(np.uint64).reshape(-1, 2) - 12345)
            allp = np.concatenate(parts)
            order = np.lexsort((allp[:, 0], allp[:, 1]))
            assert np.array_equal(allp[order], want[1]), (pats, kw, "windows")
            # a starved grid (no circular wait whatever part of the grid runs)
            for blocks in (1, 3):
                gpu.force_single_grid(blocks)
                try:
                    out = plan.scan(d.data_ptr(), n, 0, n, 0, pos.data_ptr(), cap)
                finally:
                    gpu.force_single_grid(0)
                assert out.count == want[0], (pats, kw, blocks)
                assert np.array_equal(pos[: 2 * out.stored].cpu().numpy().astype(np.uint64).reshape(-1, 2), want[1]), (pats, kw, blocks)
            plan.close()
            del d, pos
    # max_count through the one-shot entry (two scans of one plan are not available there: the dense road is reached by the retry
    # of the overflowing item rings)
    for pats in ([b"a", b"he"], [b"e", b"o", b"lo"]):
        kw = dict(max_count=4321)
        want = oracle_engine.call(abi.RA_AHO_CORASICK, abi.Params(pats, **kw), mid)
        got = gpu.search(abi.Params(pats, **kw), mid)
        assert got[0] == want[0] and np.array_equal(got[1], want[1]), (pats, got[0], want[0])
