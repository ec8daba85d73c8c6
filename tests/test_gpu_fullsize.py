"""Full BASELINE sizes on the GPU (32 GiB literal, 8 GiB single byte): the oracle cannot scan these in seconds, so
parity is established through size-independent properties — the closed-form match count of the synthetic
generator, strict sortedness, every reported offset really holding the pattern — plus oracle comparisons on
windows (incl. every GiB boundary plant)."""
import os

import numpy as np
import pytest

import oracle_lib as ol
from krep_amd import abi

pytestmark = pytest.mark.gpu

SEED, PERIOD, PAT = 20260925, 10000, b"Sherlock"
GIB = 1 << 30
M64 = (1 << 64) - 1


def splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & np.uint64(M64)
    x = ((x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & np.uint64(M64)
    x = ((x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & np.uint64(M64)
    return x ^ (x >> np.uint64(31))


def expected_kind2_starts(n, plen=8):
    """Start offsets the generator plants (krep_amd/csrc/kg_synth.h, kind 2), closed form."""
    with np.errstate(over="ignore"):
        k = np.arange((n + PERIOD - 1) // PERIOD, dtype=np.uint64)
        h = splitmix64(np.uint64(SEED) ^ np.uint64(0xA5A5A5A5DEADBEEF) ^ (k * np.uint64(0x9FB21C651E98DF25)))
        s = k * np.uint64(PERIOD) + h % np.uint64(PERIOD - plen + 1)
    s = s.astype(np.int64)
    b = ((s + GIB // 2) // GIB) * GIB
    near = (b != 0) & (s + 2 * plen + 3 > b) & (s < b + 2 * plen)
    s = s[~near & (s + plen <= n)]
    bp = np.arange(1, n // GIB + 2, dtype=np.int64) * GIB - 3
    bp = bp[bp + plen <= n]
    return np.sort(np.concatenate([s, bp]))


@pytest.fixture(scope="module")
def gpu():
    import krep_amd
    e = krep_amd.load()
    assert e.device_count() >= 1
    return e


def test_literal_32gib_properties(gpu):
    import torch
    n = 32 * GIB
    free, _ = torch.cuda.mem_get_info()
    if free < n + (2 << 30):
        pytest.skip("not enough free HBM for the 32 GiB haystack")
    buf = torch.empty(n + 64, dtype=torch.uint8, device="cuda")
    gpu.generate(buf.data_ptr(), n, 0, 2, SEED, PAT, PERIOD)
    want = expected_kind2_starts(n)
    cap = len(want) + 4096
    pos = torch.empty(2 * cap, dtype=torch.int64, device="cuda")
    plan = gpu.plan(abi.Params([PAT]))
    out = plan.scan(buf.data_ptr(), n, 0, n, 0, pos.data_ptr(), cap)
    assert not out.overflow
    # (1) closed-form count, (2) the exact list
    assert out.count == out.total_matches == len(want) == out.stored
    rec = pos[: 2 * out.stored].view(-1, 2)
    starts = rec[:, 0]
    assert torch.equal(starts.cpu(), torch.from_numpy(want))
    assert bool(torch.all(rec[:, 1] - rec[:, 0] == len(PAT)))
    # (3) strict sortedness (size-independent), (4) every offset really holds the literal
    assert bool(torch.all(starts[1:] > starts[:-1]))
    idx = starts[:, None] + torch.arange(len(PAT), device="cuda")[None, :]
    assert bool(torch.all(buf[idx] == torch.tensor(list(PAT), dtype=torch.uint8, device="cuda")[None, :]))
    # (5) count-only and -c modes agree with the list (every planted line is distinct or not: check vs windows below)
    cnt = gpu.plan(abi.Params([PAT], count_lines=True, only_match=True)).scan(buf.data_ptr(), n)
    assert cnt.count == len(want)
    # (6) oracle on windows: around every 4th GiB boundary plant and a few interior spots
    o = ol.checker()  # the compiled reference (oracle/_ref), function by function; the restatement only where it is absent
    spots = [0, n - (1 << 20)] + [j * GIB - (1 << 19) for j in range(1, 32, 4)] + [5 * GIB + 12345, 17 * GIB + 999]
    for lo in spots:
        hi = min(n, lo + (1 << 20))
        win = buf[lo:hi].cpu().numpy()
        _, wpos = o.call(abi.RA_BMH, abi.Params([PAT]), win)
        inside = want[(want >= lo) & (want + len(PAT) <= hi)]
        assert np.array_equal(wpos[:, 0].astype(np.int64) + lo, inside), lo
        # distinct lines in the window through the sharded device API == oracle -c on the window
        lines_want, _ = o.call(abi.RA_BMH, abi.Params([PAT], count_lines=True), win)
        pl = gpu.plan(abi.Params([PAT], count_lines=True))
        got = pl.scan(buf.data_ptr() + lo, hi - lo)
        assert got.count == lines_want, lo
    plan.close()


def test_single_byte_8gib_checksum(gpu):
    import torch
    n = 8 * GIB
    buf = torch.empty(n + 64, dtype=torch.uint8, device="cuda")
    gpu.generate(buf.data_ptr(), n, 0, 3, SEED, b"#", 0)
    truth = 0
    csum = 0
    chunk = 1 << 30
    for lo in range(0, n, chunk):  # reference count and checksum of offsets, chunked to bound temporaries
        nz = torch.nonzero(buf[lo:lo + chunk] == ord("#")).flatten()
        truth += int(nz.numel())
        csum += int(nz.sum().item()) + lo * int(nz.numel())
    cap = truth + 4096
    pos = torch.empty(2 * cap, dtype=torch.int64, device="cuda")
    out = gpu.plan(abi.Params([b"#"])).scan(buf.data_ptr(), n, 0, n, 0, pos.data_ptr(), cap)
    assert out.count == truth == out.stored and not out.overflow
    starts = pos[: 2 * truth].view(-1, 2)[:, 0]
    assert bool(torch.all(starts[1:] > starts[:-1]))
    assert int(starts.sum().item()) == csum  # checksum of all offsets
    assert bool(torch.all(buf[starts] == ord("#")))


def test_thousand_patterns_2gib_against_threaded_reference(gpu):
    """BASELINE config 4 at 2 GiB: the complete GPU match list (11-byte average patterns, ~0.7 M matches) against the
    CPU checker run on all host cores — chunked with overlap, each chunk keeping the matches whose START it owns."""
    import ctypes as C
    import threading
    import torch
    import random as pyrandom
    import struct
    rng = pyrandom.Random(1234)
    pats = [bytes(rng.randrange(97, 123) for _ in range(rng.randint(4, 16))) for _ in range(1000)]
    head = struct.pack("<I", len(pats))
    off, body = 4 + 8 * len(pats), b""
    for p in pats:
        head += struct.pack("<II", off + len(body), len(p))
        body += p
    n = 2 * GIB
    buf = torch.empty(n + 64, dtype=torch.uint8, device="cuda")
    gpu.generate(buf.data_ptr(), n, 0, 4, SEED, head + body, 4096)
    cap = n // 1500
    pos = torch.empty(2 * cap, dtype=torch.int64, device="cuda")
    out = gpu.plan(abi.Params(pats)).scan(buf.data_ptr(), n, 0, n, 0, pos.data_ptr(), cap)
    assert not out.overflow and out.stored == out.total_matches
    got = pos[: 2 * out.stored].view(-1, 2).cpu().numpy().astype(np.uint64)
    # the reference's emission order: end ascending, then start ascending
    assert np.all((got[1:, 1] > got[:-1, 1]) | ((got[1:, 1] == got[:-1, 1]) & (got[1:, 0] >= got[:-1, 0])))
    text = buf[:n].cpu().numpy()
    eng = ol.ref(abi.REF_SCALAR) or ol.ref(abi.REF_AVX2) or ol.oracle()
    threads = min(64, os.cpu_count() or 8)
    chunk = (n + threads - 1) // threads
    parts = [None] * threads

    def work(i):
        lo, hi = i * chunk, min(n, (i + 1) * chunk)
        b1 = min(n, hi + 16)
        _, p = eng.call(abi.RA_AHO_CORASICK, abi.Params(pats), text[lo:b1])
        p = p + np.uint64(lo)
        parts[i] = p[(p[:, 0] >= lo) & (p[:, 0] < hi)]

    ths = [threading.Thread(target=work, args=(i,)) for i in range(threads)]
    [t.start() for t in ths]
    [t.join() for t in ths]
    want = np.concatenate(parts)
    want = want[np.lexsort((want[:, 0], want[:, 1]))]
    assert len(want) == out.count
    assert np.array_equal(got, want)


def _wrap64(x):
    return x & M64


def test_single_byte_32gib_full_size(gpu):
    """BASELINE config 3 AT FULL SIZE (32 GiB, 1 % hits, ~3.4e8 records = 5.5 GB of match_position_t): count, a
    checksum of all offsets (mod 2^64), strict sortedness, every offset holds the byte, and EXACT record lists on windows
    above 4, 8, 16 and 31 GiB — any 32-bit truncation of an offset on the staging/gather path fails here."""
    import torch
    n = 32 * GIB
    free, _ = torch.cuda.mem_get_info()
    if free < n + (12 << 30):
        pytest.skip("not enough free HBM for the 32 GiB haystack + 5.5 GB of records")
    buf = torch.empty(n + 64, dtype=torch.uint8, device="cuda")
    gpu.generate(buf.data_ptr(), n, 0, 3, SEED, b"#", 0)
    truth, csum = 0, 0
    chunk = 1 << 30
    for lo in range(0, n, chunk):
        nz = torch.nonzero(buf[lo:lo + chunk] == ord("#")).flatten()
        truth += int(nz.numel())
        csum = _wrap64(csum + int(nz.sum().item()) + lo * int(nz.numel()))
        del nz
    cap = truth + 4096
    pos = torch.empty(2 * cap, dtype=torch.int64, device="cuda")
    out = gpu.plan(abi.Params([b"#"])).scan(buf.data_ptr(), n, 0, n, 0, pos.data_ptr(), cap)
    assert out.count == truth == out.stored and not out.overflow
    rec = pos[: 2 * truth].view(-1, 2)
    starts = rec[:, 0].contiguous()
    assert bool(torch.all(starts[1:] > starts[:-1]))
    assert bool(torch.all(rec[:, 1] == starts + 1))
    assert _wrap64(int(starts.sum().item())) == csum
    assert int(starts[-1].item()) > 31 * GIB
    for lo in range(0, truth, 1 << 26):  # every reported offset really holds the byte (chunked gather)
        assert bool(torch.all(buf[starts[lo:lo + (1 << 26)]] == ord("#")))
    for wlo in (4 * GIB - (1 << 19), 4 * GIB + 12345, 8 * GIB - 77, 16 * GIB + (1 << 20) + 5, 31 * GIB + 999, n - (1 << 20)):
        whi = min(n, wlo + (1 << 20))
        want = torch.nonzero(buf[wlo:whi] == ord("#")).flatten() + wlo
        i0 = int(torch.searchsorted(starts, torch.tensor([wlo], device="cuda")).item())
        i1 = int(torch.searchsorted(starts, torch.tensor([whi], device="cuda")).item())
        assert torch.equal(starts[i0:i1], want), wlo
    # the line count (-c) at full size against the same truth: distinct lines holding a '#'
    is_nl = None
    lines = 0
    carry_open = False  # the line entering the chunk already holds a '#'
    for lo in range(0, n, chunk):
        seg = buf[lo:lo + chunk]
        hit = seg == ord("#")
        nl = seg == 10
        line_id = torch.cumsum(nl.to(torch.int32), 0)  # line index of every byte relative to the chunk (0 = entering line)
        ids = torch.unique(line_id[hit])
        k = int(ids.numel())
        first_is_entering = k > 0 and int(ids[0].item()) == 0
        # a '#' that IS at a newline position cannot happen ('#' != '\n'); bytes after a '\n' have the incremented id
        lines += k - (1 if (first_is_entering and carry_open) else 0)
        last_id = int(line_id[-1].item())
        if last_id == 0:
            carry_open = carry_open or k > 0
        else:
            carry_open = k > 0 and int(ids[-1].item()) == last_id
        del seg, hit, nl, line_id, ids
    got = gpu.plan(abi.Params([b"#"], count_lines=True)).scan(buf.data_ptr(), n)
    assert got.count == lines


def _dictionary_1000():
    import random as pyrandom
    import struct
    rng = pyrandom.Random(1234)
    pats = [bytes(rng.randrange(97, 123) for _ in range(rng.randint(4, 16))) for _ in range(1000)]
    head = struct.pack("<I", len(pats))
    off, body = 4 + 8 * len(pats), b""
    for p in pats:
        head += struct.pack("<II", off + len(body), len(p))
        body += p
    return pats, head + body


def test_thousand_patterns_32gib_full_size(gpu):
    """BASELINE config 4 AT FULL SIZE (32 GiB, 1000 patterns of 4-16 bytes, ~1.1e7 matches):
      * the whole list is in the reference's emission order (end ascending, then start ascending), in bounds, lengths 4..16;
      * its length equals the SUM of the 1000 single-literal all-occurrence counts taken by the literal kernel at full size
        (an independent code path; the reference's own check, test/test_multiple_patterns.c:350-466, at 1 MiB);
      * exact (start, end) lists against the oracle's aho_corasick_search on 1 MiB windows above 4, 8, 16 and 31 GiB —
        a 32-bit truncation of the staged word (start << 11 | len) or of the gather would fail here."""
    import torch
    n = 32 * GIB
    free, _ = torch.cuda.mem_get_info()
    if free < n + (4 << 30):
        pytest.skip("not enough free HBM for the 32 GiB haystack")
    pats, packed = _dictionary_1000()
    buf = torch.empty(n + 64, dtype=torch.uint8, device="cuda")
    gpu.generate(buf.data_ptr(), n, 0, 4, SEED, packed, 4096)
    cap = n // 1500
    pos = torch.empty(2 * cap, dtype=torch.int64, device="cuda")
    out = gpu.plan(abi.Params(pats)).scan(buf.data_ptr(), n, 0, n, 0, pos.data_ptr(), cap)
    assert not out.overflow and out.stored == out.total_matches == out.count
    rec = pos[: 2 * out.stored].view(-1, 2)
    st, en = rec[:, 0], rec[:, 1]
    assert bool(torch.all((en[1:] > en[:-1]) | ((en[1:] == en[:-1]) & (st[1:] >= st[:-1]))))
    ln = en - st
    assert int(ln.min().item()) >= 4 and int(ln.max().item()) <= 16 and int(st.min().item()) >= 0 and int(en.max().item()) <= n
    assert int(st.max().item()) > 31 * GIB
    # sum of single-literal all-occurrence counts (boyer_moore_search family: --algo=bm) at full size
    gpu.set_algo_override(abi.ALGO_BM)
    try:
        total = 0
        for p in pats:
            pl = gpu.plan(abi.Params([p], count_lines=True, only_match=True))
            total += pl.scan(buf.data_ptr(), n).count
            pl.close()
    finally:
        gpu.set_algo_override(abi.ALGO_AUTO)
    assert total == out.count
    # -c at full size (the list road: records + newline gaps) against an independent path: the 1-based line number of every
    # record's start (kg_format.hip, krep_gpu_line_numbers) -> the number of distinct lines
    lines_dev = torch.empty(out.stored, dtype=torch.int64, device="cuda")
    gpu.line_numbers(buf.data_ptr(), n, pos.data_ptr(), out.stored, lines_dev.data_ptr())
    distinct = int(torch.unique(lines_dev).numel())
    got_c = gpu.plan(abi.Params(pats, count_lines=True)).scan(buf.data_ptr(), n)
    assert got_c.count == distinct == got_c.line_count and got_c.total_matches == out.count
    del lines_dev
    o = ol.checker()  # aho_corasick_search of the compiled reference
    order = torch.argsort(st, stable=True)  # by start, ties keep the (end) order
    st_sorted = st[order]
    for wlo in (4 * GIB - (1 << 19), 4 * GIB + 4321, 8 * GIB - 100, 16 * GIB + (1 << 20) + 7, 31 * GIB + 555, n - (1 << 20)):
        whi = min(n, wlo + (1 << 20))
        b0, b1 = max(0, wlo - 16), min(n, whi + 16)
        win = buf[b0:b1].cpu().numpy()
        _, wpos = o.call(abi.RA_AHO_CORASICK, abi.Params(pats), win)
        wpos = wpos.astype(np.int64) + b0
        keep = (wpos[:, 0] >= wlo) & (wpos[:, 0] < whi)
        want = wpos[keep]  # already in (end, start) order
        i0 = int(torch.searchsorted(st_sorted, torch.tensor([wlo], device="cuda")).item())
        i1 = int(torch.searchsorted(st_sorted, torch.tensor([whi], device="cuda")).item())
        idx = torch.sort(order[i0:i1]).values  # back to emission order
        got = rec[idx].cpu().numpy()
        assert np.array_equal(got, want), (wlo, len(got), len(want))


def test_dense_literal_1gib_overflowing_slots(gpu):
    """A text with ~47 hits per 32 KiB unit: every unit overflows the 16-entry staging slot (emit-mode re-scan), and the
    plan moves to 64-entry slots for its second scan — both lists must be the planted one."""
    import torch
    n, period = GIB, 700
    buf = torch.empty(n + 64, dtype=torch.uint8, device="cuda")
    gpu.generate(buf.data_ptr(), n, 0, 2, SEED, PAT, period)
    plan = gpu.plan(abi.Params([PAT]))
    cnt = gpu.plan(abi.Params([PAT], count_lines=True, only_match=True)).scan(buf.data_ptr(), n)
    cap = cnt.count + 4096
    pos = torch.empty(2 * cap, dtype=torch.int64, device="cuda")
    first = None
    for rep in range(3):
        pos.zero_()
        out = plan.scan(buf.data_ptr(), n, 0, n, 0, pos.data_ptr(), cap)
        assert out.count == out.stored == cnt.count and not out.overflow
        assert out.count > n // period - 8  # (the generator drops the plants next to a GiB boundary)
        rec = pos[: 2 * out.stored].view(-1, 2)
        starts = rec[:, 0]
        assert bool(torch.all(starts[1:] > starts[:-1]))
        assert bool(torch.all(rec[:, 1] - rec[:, 0] == len(PAT)))
        idx = starts[:, None] + torch.arange(len(PAT), device="cuda")[None, :]
        assert bool(torch.all(buf[idx] == torch.tensor(list(PAT), dtype=torch.uint8, device="cuda")[None, :]))
        if first is None:
            first = starts.clone()
        else:
            assert torch.equal(first, starts)
    # the compiled reference on a window
    o = ol.checker()
    win = buf[3 << 20: 5 << 20].cpu().numpy()
    _, wpos = o.call(abi.RA_BMH, abi.Params([PAT]), win)
    lo, hi = 3 << 20, 5 << 20
    inside = first[(first >= lo) & (first + len(PAT) <= hi)].cpu().numpy()
    assert np.array_equal(wpos[:, 0].astype(np.int64) + lo, inside)
    plan.close()


def test_one_pass_writers_8gib_properties(gpu):
    """The round-5 one-pass record writers at a size where a wave scans hundreds of tickets and the resolver runs over 10^5 of them
    (the small-text tests reach them through starved grids): a dense 2-byte literal (kg_single.hip MULTI), a dense dictionary with
    a long length (kg_ac_tiny.hip DENSE) and a sparse one (its item flavour), 8 GiB of the config-2 text.  Properties that do not
    need the oracle: the count equals the count-only scan's; the list is in the reference's order (END ascending, longest first)
    and has no duplicates; every record holds one of the patterns; the number of records of a single-byte pattern equals an
    independent device-side count of that byte — and windows of the list equal the compiled reference's."""
    import torch
    n = 8 * GIB
    free, _ = torch.cuda.mem_get_info()
    if free < n + (24 << 30):
        pytest.skip("not enough free HBM")
    buf = torch.empty(n + 64, dtype=torch.uint8, device="cuda")
    gpu.generate(buf.data_ptr(), n, 0, 2, SEED, PAT, PERIOD)
    o = ol.checker()
    jobs = [([b" a"], "single"), ([b"a", b"Sherlock"], "dense"), ([b"he", b"she", b"hers"], "items")]
    for pats, road in jobs:
        algo = abi.RA_AHO_CORASICK if len(pats) > 1 else gpu.mirror_select(abi.Params(pats), n)
        cnt = gpu.plan(abi.Params(pats, count_lines=True, only_match=True)).scan(buf.data_ptr(), n).count
        cap = cnt + 4096
        pos = torch.empty(2 * cap, dtype=torch.int64, device="cuda")
        plan = gpu.plan(abi.Params(pats))
        before = (gpu.single_launches(), gpu.tiny_dense_launches(), gpu.tiny_launches())
        for rep in range(3):  # (the first scan of a plan learns the density; the later ones take the road it chose)
            pos.zero_()
            out = plan.scan(buf.data_ptr(), n, 0, n, 0, pos.data_ptr(), cap)
            assert out.count == out.stored == cnt and not out.overflow, (pats, rep, out.count, cnt)
        after = (gpu.single_launches(), gpu.tiny_dense_launches(), gpu.tiny_launches())
        took = {"single": after[0] - before[0], "dense": after[1] - before[1], "items": (after[2] - before[2]) - (after[1] - before[1])}[road]
        assert took >= 2, (pats, road, before, after)
        rec = pos[: 2 * cnt].view(-1, 2)
        st, en = rec[:, 0], rec[:, 1]
        # order: END ascending, longest first among equal ENDs (a single literal: starts ascending) — and no duplicates
        assert bool(torch.all((en[1:] > en[:-1]) | ((en[1:] == en[:-1]) & (st[1:] > st[:-1])))), pats
        assert int(st.min()) >= 0 and int(en.max()) <= n
        ln = en - st
        seen = torch.zeros(cnt, dtype=torch.bool, device="cuda")
        for p in pats:
            sel = ln == len(p)
            ss = st[sel]
            ok = torch.ones(ss.numel(), dtype=torch.bool, device="cuda")
            for k, c in enumerate(p):  # (byte by byte: no (records x length) gather of a billion rows)
                ok &= buf[ss + k] == c
            assert bool(torch.all(ok)), (pats, p)
            seen |= sel
            if len(p) == 1:
                assert int(sel.sum()) == int(torch.count_nonzero(buf[:n] == p[0])), (pats, p)
        assert bool(torch.all(seen)), pats
        # windows against the compiled reference (records are owned by their START in a window scan: select by start)
        for lo in (0, (3 << 30) - 70_000, n - (1 << 20)):
            hi = min(n, lo + (1 << 20))
            want = o.call(algo, abi.Params(pats), buf[lo:hi].cpu().numpy())[1].astype(np.int64)
            inside = (st >= lo) & (en <= hi)
            got = rec[inside].cpu().numpy()
            assert np.array_equal(got - lo, want), (pats, lo)
        plan.close()
        del pos


@pytest.mark.parametrize("short_words", [False, True])
def test_word_dictionary_8gib_of_word_text(gpu, short_words):
    """The multi-pattern scan on WORD text at a size where its round-6 machinery is all in play (8 GiB: tickets of 8 units — the verify
    stage deferred over the ticket —, anchors with the five-class index, the exact dictionary; with two short words in the list the
    two-part scan and its merged record list, kg_scan.hip scan_ac_split):
      * the whole list is in aho_corasick_search's emission order (/root/reference/aho_corasick.c:383-437: end ascending, longest first);
      * its length equals the SUM of the single-literal all-occurrence counts of every word (an independent code path, at full size);
      * exact (start, end) lists against the compiled reference's aho_corasick_search on 1-MiB windows, one of them above 4 GiB."""
    import torch
    import wordlist
    n = 8 * GIB
    free, _ = torch.cuda.mem_get_info()
    if free < n + (6 << 30):
        pytest.skip("not enough free HBM")
    W = wordlist.word_list()
    pats = wordlist.dictionary(W, "rare")
    if short_words:
        pats = pats + [w for w in W if len(w) == 3][:2] + [b"of"]
    buf = torch.empty(n + 64, dtype=torch.uint8, device="cuda")
    gpu.generate(buf.data_ptr(), n, 0, 5, 20260930, wordlist.pack(W), 80)
    cap = n // (32 if short_words else 96)  # (the first three-letter words of the list are frequent ones: ~1e8 records)
    pos = torch.empty(2 * cap, dtype=torch.int64, device="cuda")
    before = gpu.anchored_launches()
    plan = gpu.plan(abi.Params(pats))
    out = plan.scan(buf.data_ptr(), n, 0, n, 0, pos.data_ptr(), cap)
    out2 = plan.scan(buf.data_ptr(), n, 0, n, 0, pos.data_ptr(), cap)
    assert not out.overflow and out.stored == out.total_matches == out.count == out2.count
    assert gpu.anchored_launches() > before and plan.split_state() == (2 if short_words else 1)
    plan.close()
    rec = pos[: 2 * out.stored].view(-1, 2)
    st, en = rec[:, 0], rec[:, 1]
    assert bool(torch.all((en[1:] > en[:-1]) | ((en[1:] == en[:-1]) & (st[1:] >= st[:-1]))))
    ln = en - st
    assert int(ln.min().item()) >= (2 if short_words else 4) and int(ln.max().item()) <= 16 and int(st.min().item()) >= 0 and int(en.max().item()) <= n
    gpu.set_algo_override(abi.ALGO_BM)
    try:
        total = 0
        for p in pats:
            pl = gpu.plan(abi.Params([p], count_lines=True, only_match=True))
            total += pl.scan(buf.data_ptr(), n).count
            pl.close()
    finally:
        gpu.set_algo_override(abi.ALGO_AUTO)
    assert total == out.count, (total, out.count)
    o = ol.checker()
    order = torch.argsort(st, stable=True)
    st_sorted = st[order]
    for wlo in (0, 123457, 2 * GIB - (1 << 19), 4 * GIB + 4321, 7 * GIB + 99, n - (1 << 20)):
        whi = min(n, wlo + (1 << 20))
        b0, b1 = max(0, wlo - 16), min(n, whi + 16)
        _, wpos = o.call(abi.RA_AHO_CORASICK, abi.Params(pats), buf[b0:b1].cpu().numpy())
        wpos = wpos.astype(np.int64) + b0
        want = wpos[(wpos[:, 0] >= wlo) & (wpos[:, 0] < whi)]
        i0 = int(torch.searchsorted(st_sorted, torch.tensor([wlo], device="cuda")).item())
        i1 = int(torch.searchsorted(st_sorted, torch.tensor([whi], device="cuda")).item())
        got = rec[torch.sort(order[i0:i1]).values].cpu().numpy()
        assert np.array_equal(got, want), (short_words, wlo, len(got), len(want))
