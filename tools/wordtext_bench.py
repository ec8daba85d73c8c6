"""Device-resident rates on WORD TEXT (generator kind 5: 80-byte lines of Zipf-drawn words from a 65536-word list with shared
affixes — the natural-language-like counterpart of the i.i.d. letter texts of BASELINE's configs; VERDICT r05 missing #2) and,
beside every row, the same scan on the i.i.d. text of kind 2.  For the dictionaries also the filter's candidate count
(KREP_GPU_AC_NOPROBE: a count-only scan that returns the number of candidates) and a parity check of four 1-MiB windows
against the compiled reference (oracle/_ref) — tools/ may use the checker, the product never does.
usage: python tools/wordtext_bench.py <gib> [reps]            -> the table of profiles/r06_wordtext.txt"""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import krep_amd
from krep_amd import abi
import wordlist
import bench

gib = float(sys.argv[1]) if len(sys.argv) > 1 else 8.0
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
n = int(gib * (1 << 30))
SEED, LINE = 20260930, 80
e = krep_amd.load()
W = wordlist.word_list()
blob = wordlist.pack(W)
buf = torch.empty(n + 64, dtype=torch.uint8, device="cuda")
cap = n // 24 + 4096
pos = torch.empty(2 * cap, dtype=torch.int64, device="cuda")


def timed(pats, kw, want_pos, env=None):
    for k, v in (env or {}).items():
        os.environ[k] = v
    try:
        plan = e.plan(abi.Params(pats, **kw))
        ts, first = [], None
        for i in range(reps + 1):
            out = plan.scan(buf.data_ptr(), n, 0, n, 0, pos.data_ptr() if want_pos else 0, cap if want_pos else 0, time_it=True)
            if i == 0:
                first = out.kernel_ms
            else:
                ts.append(out.kernel_ms)
        global last_info
        last_info = plan.anchor_info() if len(pats) > 1 else None
        plan.close()
    finally:
        for k in (env or {}):
            os.environ.pop(k, None)
    assert not out.overflow, (pats[:3], out.count, cap)
    return statistics.median(ts), first, out


def check_windows(pats, out):
    import oracle_lib as ol
    o = ol.checker()
    rec = pos[: 2 * out.stored].view(-1, 2)
    st = rec[:, 0]
    order = torch.argsort(st, stable=True)
    ss = st[order]
    tot = 0
    for wlo in (0, (n // 3) | 12345, n // 2 + 777, n - (1 << 20)):
        whi = min(n, wlo + (1 << 20))
        b0, b1 = max(0, wlo - 16), min(n, whi + 16)
        _, wp = o.call(abi.RA_AHO_CORASICK if len(pats) > 1 else e.mirror_select(abi.Params(pats), b1 - b0), abi.Params(pats), buf[b0:b1].cpu().numpy())
        wp = wp.astype(np.int64) + b0
        want = wp[(wp[:, 0] >= wlo) & (wp[:, 0] < whi)]
        i0 = int(torch.searchsorted(ss, torch.tensor([wlo], device="cuda")).item())
        i1 = int(torch.searchsorted(ss, torch.tensor([whi], device="cuda")).item())
        got = rec[torch.sort(order[i0:i1]).values].cpu().numpy()
        if not np.array_equal(got, want):
            return f"MISMATCH at {wlo}: {len(got)} vs {len(want)}"
        tot += len(want)
    return f"4 windows == reference ({tot} records)"


rows = []
last_info = None
texts = (("word text (kind 5)", lambda: e.generate(buf.data_ptr(), n, 0, 5, SEED, blob, LINE)),
         ("i.i.d. letters (kind 2)", lambda: e.generate(buf.data_ptr(), n, 0, 2, SEED, b"Sherlock", 10000)))
dicts = [(k, wordlist.dictionary(W, k)) for k in ("rare", "uniform", "common")] + [("BASELINE cfg 4 (random 4-16 B)", bench.ac_patterns())]
lits = [("8-byte literal, rare word", [next(w for w in W[40000:] if len(w) == 8)]), ("8-byte literal 'Sherlock' (absent)", [b"Sherlock"]),
        ("single byte 'q' (~1 % of word text)", [b"q"]), ("single byte 'e'", [b"e"]), ("'the' -w", [b"the"]), ("'tion'", [b"tion"])]
print(f"# {gib:g} GiB resident in HBM, median of {reps} launches after one (the first launch beside it), hipEvent kernel time; GB/s of text", flush=True)
for tname, gen in texts:
    gen()
    torch.cuda.synchronize()
    print(f"## {tname}", flush=True)
    for name, pats in lits:
        kw = dict(whole_word=True) if name.endswith("-w") else {}
        try:
            t_c, _, oc = timed(pats, dict(count_lines=True, only_match=True, **kw), False)
            try:
                t_p, f_p, op = timed(pats, kw, True)
            except AssertionError:  # more records than this tool's list holds (a frequent single byte): counted only
                print(f"{name:38s} {oc.count:12d} matches  count {n / t_c / 1e6:6.0f} GB/s ({t_c:.2f} ms)  (offsets: more records than the tool's {cap}-record list)", flush=True)
                continue
            chk = check_windows(pats, op) if not kw else "-"
            print(f"{name:38s} {oc.count:12d} matches  count {n / t_c / 1e6:6.0f}  offsets {n / t_p / 1e6:6.0f} GB/s ({t_p:.2f} ms, first {f_p:.2f})  {chk}", flush=True)
        except Exception as ex:
            print(f"{name:38s} failed: {ex!r}", flush=True)
    for name, pats in dicts:
        try:
            t_c, _, oc = timed(pats, dict(count_lines=True, only_match=True), False)
            _, _, ocand = timed(pats, dict(count_lines=True, only_match=True), False, env={"KREP_GPU_AC_NOPROBE": "1"})
            t_f, _, _ = timed(pats, dict(count_lines=True, only_match=True), False, env={"KREP_GPU_AC_NOVERIFY": "1"})
            t_p, f_p, op = timed(pats, {}, True)
            t_l, _, ol_ = timed(pats, dict(count_lines=True), False)
            info = last_info
            chk = check_windows(pats, op)
            chk += f"  anchors: state {info[0]}, {info[1]} patterns moved, est. candidates/tested position {100 * info[2]:.3f} % -> {100 * info[3]:.3f} %" if info else ""
            print(f"1000 words, {name:32s} {oc.count:12d} matches  {ocand.count:12d} candidates ({ocand.count / n * 100:.3f} % of bytes)  "
                  f"filter alone {n / t_f / 1e6:6.0f}  count {n / t_c / 1e6:6.0f}  offsets {n / t_p / 1e6:6.0f} GB/s ({t_p:.2f} ms = {n / t_p / 8e9:.3f} of 8 TB/s, first {f_p:.2f})  "
                  f"-c {n / t_l / 1e6:6.0f} ({ol_.count} lines)  {chk}", flush=True)
        except Exception as ex:
            print(f"1000 words, {name:32s} failed: {ex!r}", flush=True)

# One plan that meets BOTH texts (round 6, kg_ac.hip ac_scan: the kernel counts its candidates, and a measurement that contradicts the
# estimate re-opens the anchor decision): the i.i.d. text first (decision: end grams), then the word text three times.
print("## one plan, i.i.d. text first, word text after (count-only kernel ms per scan; state = anchor decision after the scan)", flush=True)
for name, pats in dicts[:2]:
    for env in ({}, {"KREP_GPU_AC_NO_RESAMPLE": "1"}):
        os.environ.update(env)
        try:
            plan = e.plan(abi.Params(pats, count_lines=True, only_match=True))
            texts[1][1]()
            torch.cuda.synchronize()
            o = plan.scan(buf.data_ptr(), n, 0, n, 0, 0, 0, time_it=True)
            row = [f"i.i.d. {o.kernel_ms:.2f} ms (state {plan.anchor_info()[0]})"]
            texts[0][1]()
            torch.cuda.synchronize()
            for i in range(4):
                o = plan.scan(buf.data_ptr(), n, 0, n, 0, 0, 0, time_it=True)
                row.append(f"words #{i + 1} {o.kernel_ms:.2f} ms = {n / o.kernel_ms / 1e6:.0f} GB/s (state {plan.anchor_info()[0]}, measured {100 * plan.anchor_measured()[0]:.2f} %)")
            plan.close()
            print(f"1000 words, {name:8s} {'first decision kept ($KREP_GPU_AC_NO_RESAMPLE)' if env else 'decision follows the text':48s} " + "; ".join(row), flush=True)
        finally:
            for k in env:
                os.environ.pop(k, None)
