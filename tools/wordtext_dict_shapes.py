"""Dictionary SHAPES on 32 GiB of word text (offsets produced): what a few long, short or many patterns do to the anchored multi-pattern scan.
A sweep for cliffs (round 6).   usage: python tools/wordtext_dict_shapes.py [gib]"""
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
cap = n // 48
pos = torch.empty(2 * cap, dtype=torch.int64, device="cuda")
rare = wordlist.dictionary(W, "rare")
line = bytes(buf[80 * 1000: 80 * 1000 + 79].cpu().numpy())
phrase = line[:line.rfind(b" ", 0, 30)]  # a phrase of the text itself, 20-30 bytes
three = [w for w in W[20000:] if len(w) == 3][:20]
SHAPES = [("1000 rare words (4-16 B)", rare), ("... + one phrase of %d bytes" % len(phrase), rare + [phrase]), ("... + twenty 3-byte words", rare + three),
          ("100 rare words", rare[:100]), ("10 rare words", rare[:10]), ("5000 rare words", wordlist.dictionary(W, "rare", n=5000)),
          ("1000 words of 8-16 bytes", wordlist.dictionary(W, "rare", min_len=8)), ("1000 words of 4-6 bytes", wordlist.dictionary(W, "rare", max_len=6))]
print(f"# {gib:g} GiB of word text, offsets produced: first scan of a fresh plan | median of the next four, GB/s of text; anchor state after")
for name, pats in SHAPES:
    try:
        plan = e.plan(abi.Params(pats))
        ts = []
        for i in range(5):
            out = plan.scan(buf.data_ptr(), n, 0, n, 0, pos.data_ptr(), cap, time_it=True)
            ts.append(out.kernel_ms)
        info = plan.anchor_info()
        plan.close()
        planc = e.plan(abi.Params(pats, count_lines=True))
        tc = []
        for i in range(4):
            oc = planc.scan(buf.data_ptr(), n, time_it=True)
            tc.append(oc.kernel_ms)
        planc.close()
        t = statistics.median(ts[1:])
        print(f"{name:40s} {len(pats):5d} patterns {out.count:11d} matches   first {ts[0]:8.2f} ms | {t:8.2f} ms = {n / t / 1e6:6.0f} GB/s ({n / t / 8e9:.3f})"
              f"{' OVERFLOW' if out.overflow else ''}   state {info[0] if info else '-'}   -c (lines) {statistics.median(tc[1:]):8.2f} ms ({oc.count} lines)", flush=True)
    except Exception as ex:
        print(f"{name:40s} failed: {str(ex)[:120]}", flush=True)
