"""The SEQUENTIAL match-set families cut into pieces (SURVEY §8e "one boundary record per shard ... one exchange step"):
greedy non-overlapping selection of a bordered pattern (simd_sse42_search krep.c:4839-4848, kmp_search :1741,
boyer_moore_search under -o :1371) and memchr_short_search's -o walk (:4495).  Across a cut the whole coupling is where the
reference's scan stands (krep_gpu_seq_carry_t::resume); shards on different devices start optimistically and the one whose
assumption was wrong is re-scanned.  Every layout must give the single-chunk reference result, offsets and order included.
Also here: the C-level RCCL all-reduce of the shard counters really runs (1-rank self-tests on a 1-GPU box)."""
import numpy as np
import pytest

import cases
import oracle_lib as ol
from krep_amd import abi

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    import krep_amd
    e = krep_amd.load()
    assert e.device_count() >= 1
    yield e
    e.set_stream_chunk(0)
    e.set_reference_simd(abi.REF_AVX2)


# (level, pattern, params kwargs, only_matching) — every family that needs the chain, plus neighbours that do not
JOBS = [
    (abi.REF_AVX2, b"abab", dict(), False),                                # simd_sse42_search, greedy, border 2
    (abi.REF_AVX2, b"aa", dict(), False),                                  # ... border 1: runs of a's are one giant cluster
    (abi.REF_AVX2, b"aba", dict(whole_word=True), False),                  # -w after the selection: a rejected hit consumes
    (abi.REF_AVX2, b"abab", dict(count_lines=True, whole_word=True), False),  # -c -w: distinct lines among the survivors
    (abi.REF_AVX2, b"aa", dict(max_count=700), False),
    (abi.REF_SCALAR, b"abab", dict(), False),                              # kmp_search (repetitive, m < 8)
    (abi.REF_SCALAR, b"aaa", dict(max_count=300), False),                  # ... and its (max_count+1)-th record
    (abi.REF_AVX2, b"ab", dict(case_sensitive=False), True),               # memchr_short_search under -o
    (abi.REF_AVX2, b"aba", dict(case_sensitive=False), True),              # ... m = 3
    (abi.REF_AVX2, b"ab", dict(case_sensitive=False, whole_word=True), True),
    (abi.REF_AVX2, b"abab", dict(case_sensitive=False), True),             # boyer_moore_search under -o: greedy
    (abi.REF_AVX2, b"aabaa", dict(case_sensitive=False, whole_word=True), True),  # ... -w BEFORE the selection
    (abi.REF_AVX2, b"abba", dict(), True),                                 # SSE4.2 under -o: all occurrences (no chain)
]


def _text(rng, n, cuts):
    text = cases.rand_text(rng, n, b"aab _\n")
    text[: n // 3] = cases.rand_text(rng, n // 3, b"ab")  # a third of it is one dense field of overlapping occurrences
    for c in cuts:  # runs and periodic stretches across every cut: clusters that straddle it by every phase
        for k, run in enumerate((b"a" * 61, b"ab" * 33, b"aab" * 21, b"abba" * 9)):
            s = c - 17 - 5 * k + (k % 2) * 40
            if 0 <= s and s + len(run) <= n:
                text[s:s + len(run)] = np.frombuffer(run, dtype=np.uint8)
    return text


def _want(o, gpu, level, pat, kw, om, text):
    gpu.set_reference_simd(level)
    p = abi.Params([pat], **kw)
    algo = gpu.mirror_select(p, text.size)
    chk = o
    o.set_only_matching(om)
    try:
        return algo, chk.call(algo, abi.Params([pat], **kw), text)
    finally:
        o.set_only_matching(False)


def _compare(gpu, level, pat, kw, om, text, want, shards, tag):
    gpu.set_reference_simd(level)
    p = abi.Params([pat], **kw)
    rc, cnt, pos = gpu.search_buffer(p, text, only_matching=om, num_gpus=shards)
    want_ret, want_pos = want
    maxc = kw.get("max_count", abi.SIZE_MAX)
    if kw.get("count_lines") and not om:
        assert cnt == min(want_ret, maxc), (tag, pat, kw, om, shards, cnt, want_ret)
    else:
        keep = min(len(want_pos), maxc)
        assert len(pos) == keep and np.array_equal(pos, want_pos[:keep]), (tag, pat, kw, om, shards, len(pos), keep)


@pytest.mark.parametrize("shards", [2, 3, 8])
def test_sharded_sequential_families_equal_the_whole_text(gpu, oracle_engine, shards):
    rng = np.random.RandomState(500 + shards)
    n = 300_007
    share = (n + shards - 1) // shards
    text = _text(rng, n, [g * share for g in range(1, shards)])
    chains = 0
    for level, pat, kw, om in JOBS:
        algo, want = _want(oracle_engine, gpu, level, pat, kw, om, text)
        cfg = gpu.default_config()
        cfg.reference_simd, cfg.only_matching = level, int(om)
        gpu.set_thread_config(cfg)
        try:
            chains += gpu.split_mode(abi.Params([pat], **kw), n) == abi.SPLIT_CHAIN
        finally:
            gpu.set_thread_config(None)
        _compare(gpu, level, pat, kw, om, text, want, shards, "sharded")
        _compare(gpu, level, pat, kw, om, text, want, 1, "one piece")
    assert chains >= 10  # the families above really take the chained road


def test_streamed_sequential_families_equal_the_whole_text(gpu, oracle_engine):
    """The same through the streamed host path: 1 MiB pieces on one device (every piece takes its predecessor's record), and
    1 MiB pieces of three shards (the first piece of a shard is optimistic)."""
    rng = np.random.RandomState(77)
    n = 5 * (1 << 20) + 12345
    text = _text(rng, n, [k << 20 for k in range(1, 6)] + [(n + 2) // 3, 2 * ((n + 2) // 3)])
    gpu.set_stream_chunk(1 << 20)
    try:
        for level, pat, kw, om in JOBS:
            algo, want = _want(oracle_engine, gpu, level, pat, kw, om, text)
            _compare(gpu, level, pat, kw, om, text, want, 1, "streamed")
            _compare(gpu, level, pat, kw, om, text, want, 3, "streamed x3")
    finally:
        gpu.set_stream_chunk(0)


def test_a_text_that_is_one_cluster(gpu, oracle_engine):
    """'aaaa...' under 'aa': one cluster from the first byte to the last — every optimistic shard start is wrong whenever the
    shard offset is odd, and the correction of one piece changes the record of the next."""
    n = 200_001
    text = np.full(n, ord("a"), dtype=np.uint8)
    for shards in (2, 3, 7):
        for level, pat, kw, om in ((abi.REF_AVX2, b"aa", dict(), False), (abi.REF_AVX2, b"aaa", dict(), False),
                                   (abi.REF_AVX2, b"aa", dict(case_sensitive=False), True)):
            algo, want = _want(oracle_engine, gpu, level, pat, kw, om, text)
            _compare(gpu, level, pat, kw, om, text, want, shards, "one cluster")


def test_device_windows_with_the_boundary_record(gpu, oracle_engine):
    """krep_gpu_scan_device_seq(): windows of a resident text in text order; a window inside the text without a record is
    refused by the plain entry point, and an optimistic scan (carry_in = NULL) differs exactly when a cluster straddles."""
    import torch
    import krep_amd
    gpu.set_reference_simd(abi.REF_AVX2)
    text = np.frombuffer(b"ababababab " * 1000 + b"a" * 999, dtype=np.uint8).copy()
    d = torch.from_numpy(text).cuda()
    n = text.size
    for pat in (b"abab", b"aa"):
        plan = gpu.plan(abi.Params([pat]))
        want = oracle_engine.call(gpu.mirror_select(abi.Params([pat]), n), abi.Params([pat]), text)
        whole = plan.scan(d.data_ptr(), n)
        assert whole.count == want[0]
        with pytest.raises(krep_amd.KrepGpuError):
            plan.scan(d.data_ptr(), n, 5003, n)  # no record: refused, never approximated
        cuts = [0, 1, 4099, 5003, 11_002, 11_503, n]
        pos = torch.zeros(2 * 20_000, dtype=torch.int64, device="cuda")
        got, carry = [], None
        for lo, hi in zip(cuts[:-1], cuts[1:]):
            o, carry = plan.scan_seq(d.data_ptr(), n, lo, hi, 0, pos.data_ptr(), 20_000, carry_in=carry)
            got.append(pos[: 2 * o.stored].cpu().numpy().astype(np.uint64).reshape(-1, 2))
        assert np.array_equal(np.concatenate(got), want[1]), pat
        plan.close()


@pytest.mark.parametrize("shards", [2, 3, 8])
def test_block_loop_count_lines_in_pieces(gpu, oracle_engine, shards):
    """-c through simd_avx512_search (33..64 B), simd_avx2_search -w (17..32 B) and neon_search (2..16 B): the block grid restarts at every counted
    line (krep.c:5203-5218, :5000-5013), so the END of the scan depends on the line-skip history of the whole text.  In pieces
    only the last one replays the end; the history {last accepted occurrence, first newline behind it} rides on the boundary
    record.  Sharded and streamed, against the compiled reference."""
    rng = np.random.RandomState(900 + shards)
    fix0 = gpu.chain_fixups()
    jobs = [(abi.REF_AVX512, 40, dict(count_lines=True)), (abi.REF_AVX512, 64, dict(count_lines=True, whole_word=True)),
            (abi.REF_AVX2, 20, dict(count_lines=True, whole_word=True)), (abi.REF_AVX2, 32, dict(count_lines=True, whole_word=True)),
            (abi.REF_AVX512, 33, dict(count_lines=True)),
            # neon_search (arm64 builds): no restart on an unterminated line — the grid origin of the previous counted line
            # rides on the record as well
            (abi.REF_NEON, 3, dict(count_lines=True)), (abi.REF_NEON, 9, dict(count_lines=True, whole_word=True)),
            (abi.REF_NEON, 16, dict(count_lines=True))]
    for n in (150_003, 64 * 2500 + 17, 3 * (1 << 20) + 99):
        share = (n + shards - 1) // shards
        for variant in range(4):
            # newline-free text / sparse newlines / dense newlines / newlines only in the first half (long open last line)
            alpha = [b"abcd_ ", b"abcd_ " * 40 + b"\n", b"abcd_ \n", b"abcd_ " * 40 + b"\n"][variant]
            text = cases.rand_text(rng, n, alpha)
            if variant == 3:
                text[n // 2:][text[n // 2:] == 10] = ord("_")
            for level, m, kw in jobs:
                pat = cases.rand_text(rng, m, b"abcd").tobytes()
                spots = [n - m, n - 100, n - 255 - m, n - 257, n - 300, n - 700, 5, n // 2] + \
                        [g * share - d for g in range(1, shards) for d in (0, 9, m - 1, m + 3)]
                if variant == 2:
                    spots = spots[:3]  # few occurrences: the last one lies far in front of the end
                for sp in spots:
                    if 0 <= sp and sp + m <= n and rng.rand() < 0.8:
                        text[sp:sp + m] = np.frombuffer(pat, dtype=np.uint8)
                gpu.set_reference_simd(level)
                p = abi.Params([pat], **kw)
                algo = gpu.mirror_select(p, n)
                assert algo in (abi.RA_AVX512, abi.RA_AVX2, abi.RA_NEON)
                cfg = gpu.default_config()
                cfg.reference_simd = level
                gpu.set_thread_config(cfg)
                try:
                    assert gpu.split_mode(p, n) == abi.SPLIT_CHAIN
                finally:
                    gpu.set_thread_config(None)
                want_ret, _ = oracle_engine.call(algo, abi.Params([pat], **kw), text)
                rc, cnt, _ = gpu.search_buffer(p, text, num_gpus=shards)
                assert cnt == want_ret, ("sharded", abi.RA_NAMES[algo], m, kw, n, variant, shards, cnt, want_ret)
                if n > (2 << 20):
                    gpu.set_stream_chunk(1 << 20)
                    try:
                        rc, cnt, _ = gpu.search_buffer(p, text, num_gpus=1)
                        assert cnt == want_ret, ("streamed", abi.RA_NAMES[algo], m, kw, n, variant, cnt, want_ret)
                        rc, cnt, _ = gpu.search_buffer(p, text, num_gpus=2)
                        assert cnt == want_ret, ("streamed x2", abi.RA_NAMES[algo], m, kw, n, variant, cnt, want_ret)
                    finally:
                        gpu.set_stream_chunk(0)
    gpu.set_reference_simd(abi.REF_AVX2)
    # the piece that ends the text is the only one that depends on the record, and only through its end-of-text replay: where
    # the record of the text in front of it turned out different, the replay alone ran again (krep_gpu_replay_tail) — no piece
    # of this family was staged and scanned a second time
    rescans, replays = (b - a for a, b in zip(fix0, gpu.chain_fixups()))
    assert replays > 0 and rescans == 0, (rescans, replays)


def test_rccl_all_reduce_really_runs(gpu, oracle_engine):
    """The shard counters meet in ONE ncclAllReduce issued from C (kg_comm.hip).  1-GPU box: a clique of one device."""
    assert gpu.rccl_version() > 0
    rng = np.random.RandomState(3)
    text = cases.rand_text(rng, 200_003, b"abcd \n")
    gpu.set_reference_simd(abi.REF_AVX2)
    for pats, kw in (([b"abcd"], dict()), ([b"d"], dict(count_lines=True)), ([b"ab", b"bcd", b"d a"], dict())):
        p = abi.Params(pats, **kw)
        algo = abi.RA_AHO_CORASICK if len(pats) > 1 else gpu.mirror_select(p, text.size)
        want = oracle_engine.call(algo, abi.Params(pats, **kw), text)
        before = gpu.rccl_calls()
        rc, cnt, pos = gpu.search_buffer(p, text, num_gpus=3)
        assert gpu.rccl_calls() == before + 1, "search_buffer(num_gpus=3) must issue exactly one all-reduce"
        assert cnt == want[0] if kw.get("count_lines") else np.array_equal(pos, want[1])
        before = gpu.rccl_calls()
        gpu.search_buffer(p, text, num_gpus=1)
        assert gpu.rccl_calls() == before  # a single shard has nothing to reduce
    # the search_func_t operators shard by configuration (the reference CLI: KREP_GPU_NUM)
    gpu.set_num_gpus(4)
    try:
        before = gpu.rccl_calls()
        p = abi.Params([b"abcd"])
        got = gpu.search(p, text)
        want = oracle_engine.call(gpu.mirror_select(p, text.size), abi.Params([b"abcd"]), text)
        assert gpu.rccl_calls() == before + 1 and got[0] == want[0] and np.array_equal(got[1], want[1])
    finally:
        gpu.set_num_gpus(1)


def test_rank_communicator_one_rank(gpu):
    """The one-process-per-GPU entry points bench.py uses: unique id -> init_rank -> all-reduce -> destroy."""
    ident = gpu.comm_unique_id()
    assert len(ident) == 128
    gpu.comm_init_rank(ident, 1, 0, 0)
    try:
        before = gpu.rccl_calls()
        assert gpu.comm_allreduce([5, 7, 2**40 + 3]) == [5, 7, 2**40 + 3]
        assert gpu.rccl_calls() == before + 1
    finally:
        gpu.comm_destroy()


def test_multi_pattern_count_lines_with_a_newline_pattern_in_pieces(gpu, oracle_engine):
    """Round 5: aho_corasick_search -c with a '\\n' inside a pattern counts emission-order line CHANGES
    (aho_corasick.c:383-396) — one window only until now.  A piece owns the matches that END in it; what it needs from the text in
    front of it is the number of newlines so far and the line of the last match's start (krep_gpu_seq_carry_t::nl_before /
    last_line).  Whole text == device windows chained through krep_gpu_scan_device_seq() == the streamed host operator == logical
    shards on several devices, against the compiled reference: patterns that span one and two line breaks, nested ones whose
    longer match starts on an EARLIER line than the shorter one before it, -w, -i, cuts inside a multi-line match."""
    import torch
    rng = np.random.RandomState(9090)
    n = 3 * (1 << 20) + 4567
    text = cases.rand_text(rng, n, b"ab \n")
    dicts = [[b"a\nb", b"ab"], [b"b\na\nb", b"\nb", b"aa"], [b"ab\n", b"\n\n", b"b a"], [b" \na", b"a \na b", b"ba"]]
    for pats in dicts:
        for kw in (dict(), dict(whole_word=True), dict(case_sensitive=False)):
            kwc = dict(count_lines=True, **kw)
            p = abi.Params(pats, **kwc)
            assert gpu.split_mode(p, n) == abi.SPLIT_CHAIN
            want = oracle_engine.call(abi.RA_AHO_CORASICK, abi.Params(pats, **kwc), text)[0]
            assert gpu.search(p, text, want_result=False)[0] == want, (pats, kw, "whole")
            # device windows in text order, each with its predecessor's record; cuts everywhere, also one byte apart
            d = torch.from_numpy(text).cuda()
            plan = gpu.plan(p)
            cuts = [0, 1, 2, 777, 65536, 65537, (1 << 20) + 3, (2 << 20) - 1, 2 << 20, n]
            carry, total = None, 0
            for lo, hi in zip(cuts[:-1], cuts[1:]):
                out, carry = plan.scan_seq(d.data_ptr(), n, lo, hi, 0, global_len=n, carry_in=carry)
                total += out.line_count
            assert total == want, (pats, kw, "device windows", total, want)
            # the same windows as buffers of their own (a halo of the longest pattern in front, global_base behind it)
            lmax = max(len(q) for q in pats)
            carry, total = None, 0
            for lo, hi in zip(cuts[:-1], cuts[1:]):
                b0 = max(0, lo - lmax - 1)
                b1 = min(n, hi + lmax + 1)
                out, carry = plan.scan_seq(d.data_ptr() + b0, b1 - b0, lo - b0, hi - b0, b0, global_len=n, carry_in=carry)
                total += out.line_count
            assert total == want, (pats, kw, "own buffers", total, want)
            plan.close()
            del d
            # the host operator: streamed in 1 MiB pieces, and sharded over 2 and 3 logical devices (optimistic first pieces,
            # the fix-up re-scans what the true record changes)
            gpu.set_stream_chunk(1 << 20)
            try:
                assert gpu.search(p, text, want_result=False)[0] == want, (pats, kw, "streamed")
                for shards in (2, 3):
                    rc, cnt, _ = gpu.search_buffer(p, text, num_gpus=shards, want_result=False)
                    assert cnt == want and rc == (0 if want else 1), (pats, kw, "shards", shards, cnt, want)
            finally:
                gpu.set_stream_chunk(0)


def test_count_lines_with_a_newline_inside_a_single_pattern_in_pieces(gpu, oracle_engine):
    """Round 5 (second half): -c through simd_sse42_search / kmp_search with a '\\n' inside the pattern
    (krep.c:4785-4795, :1703-1707) — after a counted line the scan resumes INSIDE the match, for SSE4.2 at a point that depends on
    the phase of its window grid.  One window only until now; a piece now takes where the reference's scan stands, the line it
    counted last and the newlines so far from the text in front of it (krep_gpu_seq_carry_t::resume / last_line / nl_before).
    Device windows chained through krep_gpu_scan_device_seq(), windows as buffers of their own, the streamed host operator and
    logical shards — every layout the whole-text count of the compiled reference; -w, max_count, the KMP override."""
    import torch
    rng = np.random.RandomState(4711)
    n = 2 * (1 << 20) + 333
    for alpha in (b"ab\n", b"ab \n\n"):
        text = cases.rand_text(rng, n, alpha)
        d = torch.from_numpy(text).cuda()
        for pat in (b"a\nb", b"a\n", b"\na", b"ab\nab", b"b\na\nb", b"\n\n"):
            for level, override in ((abi.REF_AVX2, abi.ALGO_AUTO), (abi.REF_SSE42, abi.ALGO_AUTO), (abi.REF_AVX2, abi.ALGO_KMP)):
                for kw in (dict(), dict(whole_word=True)):
                    gpu.set_reference_simd(level)
                    gpu.set_algo_override(override)
                    try:
                        p = abi.Params([pat], count_lines=True, **kw)
                        algo = gpu.mirror_select(p, n)
                        if algo not in (abi.RA_SSE42, abi.RA_KMP):
                            continue
                        assert gpu.split_mode(p, n) == abi.SPLIT_CHAIN
                        want = oracle_engine.call(algo, abi.Params([pat], count_lines=True, **kw), text)[0]
                        assert gpu.search(p, text, want_result=False)[0] == want, (pat, level, override, kw, "whole")
                        plan = gpu.plan(p)
                        cuts = [0, 1, 777, 65536, 65537, (1 << 20) + 3, n]
                        carry, total = None, 0
                        for lo, hi in zip(cuts[:-1], cuts[1:]):
                            out, carry = plan.scan_seq(d.data_ptr(), n, lo, hi, 0, global_len=n, carry_in=carry)
                            total += out.line_count
                        assert total == want, (pat, level, override, kw, "device windows", total, want)
                        carry, total = None, 0
                        for lo, hi in zip(cuts[:-1], cuts[1:]):
                            b0, b1 = max(0, lo - 40), min(n, hi + 40)
                            out, carry = plan.scan_seq(d.data_ptr() + b0, b1 - b0, lo - b0, hi - b0, b0, global_len=n, carry_in=carry)
                            total += out.line_count
                        assert total == want, (pat, level, override, kw, "own buffers", total, want)
                        plan.close()
                        gpu.set_stream_chunk(1 << 19)
                        try:
                            assert gpu.search(p, text, want_result=False)[0] == want, (pat, level, override, kw, "streamed")
                            for shards in (2, 3):
                                cfg = gpu.default_config()
                                cfg.reference_simd, cfg.algo_override = level, override
                                rc, cnt, _ = gpu.search_buffer(p, text, num_gpus=shards, want_result=False, cfg=cfg)
                                assert cnt == want, (pat, level, override, kw, "shards", shards, cnt, want)
                        finally:
                            gpu.set_stream_chunk(0)
                    finally:
                        gpu.set_algo_override(abi.ALGO_AUTO)
        del d
    gpu.set_reference_simd(abi.REF_AVX2)
