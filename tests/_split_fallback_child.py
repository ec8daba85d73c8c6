"""Child process of tests/test_gpu_wordtext.py::test_split_scan_falls_back_when_the_merged_list_is_too_long: $KREP_GPU_AC_SPLIT_MAX is read
when the library is loaded, so the limit is set before this process loads it."""
import os, sys
os.environ["KREP_GPU_AC_SPLIT_MAX"] = "1000"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, HERE)
import numpy as np, torch, krep_amd, wordlist, oracle_lib as ol
from krep_amd import abi

e = krep_amd.load(); o = ol.checker()
W = wordlist.word_list()
n = 3 * (1 << 20) + 99
text = e.generate_host(n, 0, 5, 20260930, wordlist.pack(W), 80)
pats = wordlist.dictionary(W, "rare", n=300) + [b"of", b"the"]
d = torch.from_numpy(np.ascontiguousarray(text)).cuda()
states = []
for kw in ({}, dict(max_count=50)):
    want = o.call(abi.RA_AHO_CORASICK, abi.Params(pats, **kw), text)
    plan = e.plan(abi.Params(pats, **kw))
    cap = int(want[0]) + 8
    pos = torch.zeros(2 * cap, dtype=torch.int64, device="cuda")
    for rep in range(2):
        out = plan.scan(d.data_ptr(), n, 0, n, 0, pos.data_ptr(), cap)
        got = pos[: 2 * out.stored].view(-1, 2).cpu().numpy().astype(np.uint64)
        assert out.count == want[0] and np.array_equal(got, want[1]), (kw, rep, out.count, want[0])
    states.append(plan.split_state())
    plan.close()
assert states == [1, 2], states  # the whole list: over the limit, one scan from then on; 50 records of each part: merged
print("split fallback ok", states)
