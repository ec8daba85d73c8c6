"""Where the anchored multi-pattern scan's time goes (development aid): the same count-only scan of word text under the ablation hooks.
usage: python tools/anchor_stage_probe.py <gib> [rare|uniform]"""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, krep_amd, wordlist
from krep_amd import abi
gib = float(sys.argv[1]) if len(sys.argv) > 1 else 32.0
kind = sys.argv[2] if len(sys.argv) > 2 else "rare"
n = int(gib * (1 << 30))
e = krep_amd.load()
W = wordlist.word_list()
buf = torch.empty(n + 64, dtype=torch.uint8, device="cuda")
e.generate(buf.data_ptr(), n, 0, 5, 20260930, wordlist.pack(W), 80)
pats = wordlist.dictionary(W, kind)
for label, env in (("full count", {}), ("filter only (NOVERIFY)", {"KREP_GPU_AC_NOVERIFY": "1"}), ("filter + enumeration (NOPROBE: count = candidates)", {"KREP_GPU_AC_NOPROBE": "1"}),
                   ("stages 1 + 2 (NOSTAGE3: count = marked END pairs)", {"KREP_GPU_AC_NOSTAGE3": "1"}),
                   ("four classes, full count", {"KREP_GPU_AC_NO_ANCHOR5": "1"}), ("four classes, stages 1 + 2", {"KREP_GPU_AC_NO_ANCHOR5": "1", "KREP_GPU_AC_NOSTAGE3": "1"}),
                   ("end grams (no anchors), full count", {"KREP_GPU_AC_NO_ANCHOR": "1"})):
    os.environ.update(env)
    plan = e.plan(abi.Params(pats, count_lines=True, only_match=True))
    ts = []
    for i in range(5):
        out = plan.scan(buf.data_ptr(), n, time_it=True)
        if i:
            ts.append(out.kernel_ms)
    plan.close()
    for k in env:
        os.environ.pop(k, None)
    print(f"{label:52s} {statistics.median(ts):8.2f} ms   count {out.count}", flush=True)
