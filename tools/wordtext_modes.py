"""The 1000-word dictionaries on 32 GiB of word text under -w and -i (offsets produced): first scan of a fresh plan and the median of the
next four.  usage: python tools/wordtext_modes.py   -> the last block of profiles/r06_wordtext.txt"""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, krep_amd, wordlist
from krep_amd import abi
n = 32 << 30
e = krep_amd.load()
W = wordlist.word_list(); blob = wordlist.pack(W)
buf = torch.empty(n + 64, dtype=torch.uint8, device="cuda")
e.generate(buf.data_ptr(), n, 0, 5, 20260930, blob, 80)
cap = n // 64
pos = torch.empty(2 * cap, dtype=torch.int64, device="cuda")
for kind in ("rare", "uniform"):
    pats = wordlist.dictionary(W, kind)
    for kw in ({}, dict(whole_word=True), dict(case_sensitive=False), dict(whole_word=True, case_sensitive=False)):
        plan = e.plan(abi.Params(pats, **kw))
        ts = []
        for i in range(5):
            out = plan.scan(buf.data_ptr(), n, 0, n, 0, pos.data_ptr(), cap, time_it=True)
            ts.append(out.kernel_ms)
        plan.close()
        print(f"{kind:8s} {str(kw):50s} {out.count:10d} matches  first {ts[0]:8.2f}  median of the rest {statistics.median(ts[1:]):8.2f} ms = {n / statistics.median(ts[1:]) / 1e6:6.0f} GB/s", flush=True)
