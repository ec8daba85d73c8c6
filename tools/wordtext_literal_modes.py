"""Single literals on 32 GiB of word text across the modes krep's CLI offers (offsets / -c lines / -c -o count, each plain, -i, -w, -i -w):
median kernel ms of four scans after the first, GB/s of text.  A sweep for slow corners (round 6).   usage: python tools/wordtext_literal_modes.py [gib]"""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, krep_amd, wordlist
from krep_amd import abi

gib = float(sys.argv[1]) if len(sys.argv) > 1 else 32.0
n = int(gib * (1 << 30))
e = krep_amd.load()
W = wordlist.word_list(); blob = wordlist.pack(W)
buf = torch.empty(n + 64, dtype=torch.uint8, device="cuda")
e.generate(buf.data_ptr(), n, 0, 5, 20260930, blob, 80)
cap = n // 24
pos = torch.empty(2 * cap, dtype=torch.int64, device="cuda")
rare8 = next(w for w in W[40000:] if len(w) == 8)
PATS = [b"the", b"tion", b"of", rare8, W[300], b"ing ", b"e", b"q"]
MODES = [("offsets", {}, True), ("-c", dict(count_lines=True), False), ("-c -o", dict(count_lines=True, only_match=True), False)]
FLAGS = [("", {}), ("-i", dict(case_sensitive=False)), ("-w", dict(whole_word=True)), ("-i -w", dict(case_sensitive=False, whole_word=True))]
print(f"# {gib:g} GiB of word text (kind 5), kernel ms (GB/s of text), median of 4 scans after the first; matches in brackets")
for pat in PATS:
    for fname, fkw in FLAGS:
        row = []
        for mname, mkw, wp in MODES:
            try:
                plan = e.plan(abi.Params([pat], **fkw, **mkw))
                ts = []
                for i in range(5):
                    out = plan.scan(buf.data_ptr(), n, 0, n, 0, pos.data_ptr() if wp else 0, cap if wp else 0, time_it=True)
                    ts.append(out.kernel_ms)
                plan.close()
                t = statistics.median(ts[1:])
                row.append(f"{mname} {t:7.2f} ({n / t / 1e6:5.0f}){' OVERFLOW' if out.overflow else ''}")
                cnt = out.count if mname != "-c" else cnt
            except Exception as ex:
                row.append(f"{mname} failed: {str(ex)[:60]}")
        print(f"{pat.decode()!r:12s} {fname:6s} [{cnt:11d}]  " + "   ".join(row), flush=True)
