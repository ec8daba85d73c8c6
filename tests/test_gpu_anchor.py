"""GPU parity of the ANCHORED multi-pattern scan (krep_amd/csrc/kg_ac_anchor.hip, ac_scan_kernel<.., ANCH>): anchor grams chosen
by rarity in a sample of the text, the ends they name verified by the end-anchored verifier.  The anchors may only decide which
ends are LOOKED at; count, every (start, end) record and the emission order must stay aho_corasick_search's
(/root/reference/aho_corasick.c:328-437).  $KREP_GPU_AC_ANCHOR=1 forces anchors by plain minimum (every offset 0..12 gets used) on
texts where they gain nothing, so that the random cases reach every corner: ends named across unit boundaries (the seven tested
positions in front of a unit), the first 16 bytes of a text, ownership windows, -w, -i, max_count, overflowing staging slots."""
import os

import numpy as np
import pytest

import cases
import wordlist
from krep_amd import abi

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    import krep_amd
    e = krep_amd.load()
    assert e.device_count() >= 1
    return e


@pytest.fixture(params=["four classes", "five classes", "five classes, tickets of 8 units", "four classes, tickets of 3 units",
                        "five classes, tickets of 8 units, verify at every unit's end"])
def forced(request):
    """$KREP_GPU_AC_ANCHOR: anchors whatever the gain; + $KREP_GPU_AC_ANCHOR5: the five-class index (ac_scan_kernel<.., ANCH = 2>) whatever
    the number of table slots it takes (4- and 5-byte patterns leave classes free: 32 or 1024 slots each).  $KREP_GPU_AC_UPT: the tickets of
    several units that only large texts get by themselves — the verify stage is deferred over a ticket (its marked END pairs collected one
    per lane, 64 per batch, a batch spanning units); $KREP_GPU_AC_NODEFER: not deferred, as first built."""
    os.environ["KREP_GPU_AC_ANCHOR"] = "1"
    if "five classes" in request.param:
        os.environ["KREP_GPU_AC_ANCHOR5"] = "1"
    else:
        os.environ["KREP_GPU_AC_NO_ANCHOR5"] = "1"
    if "tickets of 8" in request.param:
        os.environ["KREP_GPU_AC_UPT"] = "8"
    if "tickets of 3" in request.param:
        os.environ["KREP_GPU_AC_UPT"] = "3"
    if "every unit" in request.param:
        os.environ["KREP_GPU_AC_NODEFER"] = "1"
    yield request.param
    for k in ("KREP_GPU_AC_ANCHOR", "KREP_GPU_AC_ANCHOR5", "KREP_GPU_AC_NO_ANCHOR5", "KREP_GPU_AC_UPT", "KREP_GPU_AC_NODEFER"):
        os.environ.pop(k, None)


class _DevicePlan:
    """One plan and one device copy of a text; scan(lo, hi, base) -> (ScanOut, records ndarray)."""

    def __init__(self, gpu, pats, kw, text):
        import torch
        self.n = len(text)
        self.buf = torch.from_numpy(np.ascontiguousarray(text)).cuda()
        self.cap = max(4096, self.n // 2)
        self.pos = torch.empty(2 * self.cap, dtype=torch.int64, device="cuda")
        self.plan = gpu.plan(abi.Params(pats, **kw))

    def scan(self, lo=0, hi=None, base=0):
        n = self.n
        out = self.plan.scan(self.buf.data_ptr(), n, lo, n if hi is None else hi, base, self.pos.data_ptr(), self.cap, global_len=base + n)
        rec = self.pos[: 2 * out.stored].view(-1, 2).cpu().numpy().copy()
        return out, rec

    def close(self):
        self.plan.close()


def _scan_device(gpu, pats, kw, text):
    """whole text, twice (the anchor decision is taken by the first scan) -> (ScanOut, records, plan.anchor_info())"""
    d = _DevicePlan(gpu, pats, kw, text)
    out, rec = d.scan()
    out2, rec2 = d.scan()
    assert out2.count == out.count and np.array_equal(rec2, rec)
    info = d.plan.anchor_info()
    d.close()
    return out, rec, info


@pytest.mark.parametrize("seed", range(6))
def test_forced_anchors_on_random_texts(gpu, oracle_engine, forced, seed):
    rng = np.random.RandomState(9100 + seed)
    before = gpu.anchored_launches()
    used = 0
    for it in range(6):
        alpha = [b"ab", b"abc\n", b"abcdefgh \n", bytes(range(97, 123)) + b"  \n", b"abAB -\n"][(seed + it) % 5]
        n = (1 << 20) + [0, 1, 15, 16, 17, 8191, 16384, 16385, 40000, 123457][rng.randint(0, 10)]
        text = cases.rand_text(rng, n, alpha)
        k = [2, 5, 9, 40, 300][rng.randint(0, 5)]
        lens = [[4, 5, 6], [4, 5, 8, 16], [5, 9, 13, 16, 17], [6, 7, 30, 64], [4, 4, 4, 12]][rng.randint(0, 5)]
        pats = [cases.pick_pattern(rng, text, lens[rng.randint(0, len(lens))], alpha) for _ in range(k)]
        if rng.rand() < 0.3:
            pats.append(pats[0])  # a duplicate pattern is reported once per copy (aho_corasick.c:383-437)
        # matches that END inside the first 16 bytes and straddle the 16-KiB units
        for s in (0, 3, 16384 - 5, 16384 - 2, 32768 - 9, 3 * 16384 - 1, n - len(pats[0])):
            p = np.frombuffer(pats[rng.randint(0, len(pats))], dtype=np.uint8)
            if 0 <= s and s + p.size <= n:
                text[s:s + p.size] = p
        kw = dict(case_sensitive=bool(rng.rand() < 0.6), whole_word=bool(rng.rand() < 0.25),
                  max_count=[abi.SIZE_MAX, abi.SIZE_MAX, abi.SIZE_MAX, 1, 77][rng.randint(0, 5)])
        mode = ["pos", "pos", "count", "lines"][rng.randint(0, 4)]
        if mode == "count":
            kw.update(count_lines=True, only_match=True)
        elif mode == "lines":
            if any(b"\n" in p for p in pats):
                continue
            kw.update(count_lines=True)
        want = oracle_engine.call(abi.RA_AHO_CORASICK, abi.Params(pats, **kw), text)
        got = gpu.search(abi.Params(pats, **kw), text)
        assert got[0] == want[0], (seed, it, pats[:4], kw, got[0], want[0])
        assert np.array_equal(got[1], want[1]), (seed, it, pats[:4], kw, got[1][:6], want[1][:6])
        used += 1
    assert used and gpu.anchored_launches() > before  # the anchored instantiation is what ran


def test_forced_anchors_in_ownership_windows_and_small_slots(gpu, oracle_engine, forced):
    """Device windows (start ownership, global base beyond 2^32) cut at and around unit boundaries, with 16-entry staging slots
    overflowing into the emit-mode re-scan."""
    rng = np.random.RandomState(77)
    n = (2 << 20) + 333
    alpha = b"abcd \n"
    text = cases.rand_text(rng, n, alpha)
    pats = sorted({cases.pick_pattern(rng, text, [4, 5, 6, 7, 9, 12][rng.randint(0, 6)], alpha) for _ in range(60)})
    _, want = oracle_engine.call(abi.RA_AHO_CORASICK, abi.Params(pats), text)
    want = want.astype(np.int64)
    base = (5 << 32) + 12345
    d = _DevicePlan(gpu, pats, {}, text)
    for lo, hi in ((0, n), (16384, 5 * 16384), (16384 - 7, 16384 + 9), (100001, 1900003), (n - 20000, n), (3, 40), (16, 48), (32768, 32768 + 16)):
        out, rec = d.scan(lo, hi, base)
        info = d.plan.anchor_info()
        assert info is not None and info[0] == 2, info  # anchored (decided by the first, whole-text scan)
        sel = want[(want[:, 0] >= lo) & (want[:, 0] < hi)] + base
        assert out.count == len(sel) and np.array_equal(rec, sel), (lo, hi, len(rec), len(sel))
    gpu.force_stage_cap(2)
    try:
        out, rec = d.scan()
        assert np.array_equal(rec, want)
    finally:
        gpu.force_stage_cap(0)
        d.close()


def test_decision_keeps_end_grams_on_iid_text_and_anchors_word_text(gpu, oracle_engine):
    """Not forced: BASELINE config 4's shape (random patterns on i.i.d. letters) keeps the round-5 kernel — no pattern moves off
    its end; a word dictionary on word text moves most of its patterns and estimates several times fewer candidates."""
    rng = np.random.RandomState(5)
    az = bytes(range(97, 123))
    pats = [cases.rand_text(rng, rng.randint(4, 17), az).tobytes() for _ in range(1000)]
    text = cases.rand_text(rng, 2 << 20, az + b"  \n")
    out, rec, info = _scan_device(gpu, pats, {}, text)
    assert info[0] == 1 and info[1] == 0, info
    _, want = oracle_engine.call(abi.RA_AHO_CORASICK, abi.Params(pats), text)
    assert np.array_equal(rec, want.astype(np.int64))
    w = wordlist.word_list()
    wtext = gpu.generate_host(3 << 20, 0, 5, 20260930, wordlist.pack(w), 80)
    wp = wordlist.dictionary(w, "rare")
    out, rec, info = _scan_device(gpu, wp, {}, wtext)
    assert info[0] == 2 and info[1] > 500 and info[3] < 0.5 * info[2], info
    _, want = oracle_engine.call(abi.RA_AHO_CORASICK, abi.Params(wp), wtext)
    assert np.array_equal(rec, want.astype(np.int64))


def test_decision_follows_the_text_a_plan_meets_later(gpu, oracle_engine):
    """One plan, two kinds of text.  The decision is taken on the first text of >= 1 MiB: i.i.d. letters here, on which a word
    dictionary's end grams are as rare as any (state 1).  The word text after it passes ~40x the candidates the estimate named; the
    kernel counts them (Counters::candidates), the scan that measured them re-opens the decision and the NEXT scan samples the word
    text and anchors.  Same records as aho_corasick_search (/root/reference/aho_corasick.c:328-437) before, at and after the switch;
    $KREP_GPU_AC_NO_RESAMPLE=1 keeps the first decision."""
    import torch
    w = wordlist.word_list()
    wp = wordlist.dictionary(w, "rare")
    rng = np.random.RandomState(77)
    iid = cases.rand_text(rng, 3 << 20, bytes(range(97, 123)) + b"  \n")
    wtext = np.frombuffer(gpu.generate_host(3 << 20, 0, 5, 20260930, wordlist.pack(w), 80), dtype=np.uint8).copy()
    want_iid = oracle_engine.call(abi.RA_AHO_CORASICK, abi.Params(wp), iid)[1].astype(np.int64)
    want_w = oracle_engine.call(abi.RA_AHO_CORASICK, abi.Params(wp), wtext)[1].astype(np.int64)
    for resample in (True, False):
        if not resample:
            os.environ["KREP_GPU_AC_NO_RESAMPLE"] = "1"
        try:
            plan = gpu.plan(abi.Params(wp))
            cap = max(len(want_w), len(want_iid)) + 16
            pos = torch.empty(2 * cap, dtype=torch.int64, device="cuda")

            def scan(text):
                d = torch.from_numpy(text).cuda()
                out = plan.scan(d.data_ptr(), len(text), 0, len(text), 0, pos.data_ptr(), cap)
                return pos[: 2 * out.stored].view(-1, 2).cpu().numpy().copy()

            assert np.array_equal(scan(iid), want_iid)
            assert plan.anchor_info()[0] == 1 and plan.anchor_measured()[1] == 0
            low = plan.anchor_measured()[0]
            assert np.array_equal(scan(wtext), want_w)  # (still the end grams: this scan is the one that measures)
            high, n = plan.anchor_measured()
            assert high > 0.008 and high > 10 * low, (low, high)
            assert n == (1 if resample else 0) and plan.anchor_info()[0] == (0 if resample else 1)
            before = gpu.anchored_launches()
            assert np.array_equal(scan(wtext), want_w)
            assert plan.anchor_info()[0] == (2 if resample else 1)
            assert (gpu.anchored_launches() > before) == resample
            if resample:
                est = plan.anchor_info()[3]
                got = plan.anchor_measured()[0]
                assert got < 0.5 * high and got < 2.0 * est + 0.002, (got, est, high)  # the estimate holds on the text it was taken on
                assert np.array_equal(scan(iid), want_iid)  # anchored tables on the other text: same records, few candidates, no repeat
                assert np.array_equal(scan(wtext), want_w)
                assert plan.anchor_measured()[1] == 1 and plan.anchor_info()[0] == 2
            plan.close()
        finally:
            os.environ.pop("KREP_GPU_AC_NO_RESAMPLE", None)


def test_end_owned_window_that_starts_on_a_16_byte_boundary(gpu, oracle_engine):
    """Multi-pattern -c on the record-list road owns a match by its END.  A window whose first owned byte is 16-byte aligned starts
    a scan unit there, and a unit verifies the ends BEHIND its first byte (its first byte is the last end of the unit in front):
    the match whose last byte is exactly the window's first one needs the unit in front of the window to be scanned as well
    (found in round 6 while the END bitmaps of the anchored scan were laid out; the round-5 kernel lost that match)."""
    import torch
    rng = np.random.RandomState(31)
    n = 40 << 20
    text = cases.rand_text(rng, n, bytes(range(97, 123)) + b"  \n")
    pats = [b"QWERTY", b"ZXCVB", b"ASDFGHJK", b"POIUY"]
    cuts = [16 << 20, (33 << 20) + 16, 35 << 20]
    for c in cuts:  # a pattern whose LAST byte is the first byte of the window [c, ...), alone on its line
        p = np.frombuffer(pats[c % len(pats)], dtype=np.uint8)
        text[c - 40:c + 40] = ord("x")
        text[c - 41] = text[c + 40] = 10
        text[c - p.size + 1:c + 1] = p
    d = torch.from_numpy(text).cuda()
    kwc = dict(count_lines=True)
    want = oracle_engine.call(abi.RA_AHO_CORASICK, abi.Params(pats, **kwc), text)[0]
    assert want >= 3
    plan = gpu.plan(abi.Params(pats, **kwc))
    whole = plan.scan(d.data_ptr(), n)
    assert whole.count == want
    edges = [0] + cuts + [n]
    outs = [plan.scan(d.data_ptr(), n, lo, hi) for lo, hi in zip(edges[:-1], edges[1:])]
    assert sum(o.total_matches for o in outs) == whole.total_matches
    assert gpu.lib.krep_gpu_combine_line_counts((abi.ScanOut * len(outs))(*outs), len(outs)) == want
    plan.close()
