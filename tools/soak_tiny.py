"""Randomised soak of the tiny-dictionary kernel (kg_ac_tiny.hip) against the compiled reference: more seeds, sizes, alphabets
and shapes than tests/test_gpu_ac_tiny.py; every case also through a re-used plan (the dense policy) and logical shards.
usage: python tools/soak_tiny.py [seeds]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import numpy as np, torch
import krep_amd, oracle_lib as ol, cases
from krep_amd import abi
gpu = krep_amd.load(); o = ol.checker()
bad = 0
def fail(*a):
    global bad
    bad += 1; print("FAIL", *[str(x)[:200] for x in a], flush=True)
    if bad > 5: sys.exit(1)
nseeds = int(sys.argv[1]) if len(sys.argv) > 1 else 30
cases_run = 0
for seed in range(7000, 7000 + nseeds):
    rng = np.random.RandomState(seed)
    for it in range(30):
        alpha = [b"ab", b"abc\n", b"abAB -\n", bytes(range(97, 105)) + b" \n", bytes(range(256)), b"\x00\x01a\n", b"etaoin shrdlu\n" * 3 + b"ETAOIN"][it % 7]
        n = [1, 2, 3, 4, 5, 17, 500, 8191, 8192, 8195, 16383, 16384, 16389, 40000, 131072, 140000, 300007, 1 << 20, (1 << 21) + 77][rng.randint(0, 19)]
        text = cases.rand_text(rng, n, alpha)
        lens = [[1], [2], [1, 2], [1, 2, 3, 4], [2, 3, 4], [2, 4], [1, 4], [1, 3], [2, 2, 3], [1, 5], [2, 6, 6], [1, 3, 7], [2, 8]][rng.randint(0, 13)]
        pats = []
        for _ in range(60):
            if len(pats) >= [2, 3, 4, 6, 9][rng.randint(0, 5)]: break
            p = cases.pick_pattern(rng, text, lens[rng.randint(0, len(lens))], alpha)
            if p not in pats and sum(len(q) == len(p) for q in pats) < 4: pats.append(p)
        if len(pats) < 2 or min(len(p) for p in pats) > 2: continue
        cs = bool(rng.rand() < 0.6)
        if not cs and len({bytes(c + 32 if 65 <= c <= 90 else c for c in p) for p in pats}) != len(pats): continue
        kw = dict(case_sensitive=cs, max_count=([abi.SIZE_MAX] * 5 + [1, 4, 77, 5000])[rng.randint(0, 9)])
        mode = ["pos", "pos", "lines", "count"][rng.randint(0, 4)]
        if mode == "lines":
            if any(b"\n" in p for p in pats): continue
            kw.update(count_lines=True)
        elif mode == "count":
            kw.update(count_lines=True, only_match=True)
        P = abi.Params(pats, **kw)
        want = o.call(abi.RA_AHO_CORASICK, P, text)
        before = gpu.tiny_launches()
        got = gpu.search(abi.Params(pats, **kw), text)
        cases_run += 1
        if got[0] != want[0] or not np.array_equal(got[1], want[1]): fail("search", seed, it, pats, kw, n, got[0], want[0])
        if kw["max_count"] and gpu.tiny_launches() == before and max(len(p) for p in pats) > 1: fail("tiny kernel not used", seed, it, pats, kw)  # (single bytes with records: kg_single.hip)
        # logical shards through the host operator
        if n >= 8192 and rng.rand() < 0.5:
            g = int(rng.randint(2, 6))
            rc, cnt, pos = gpu.search_buffer(abi.Params(pats, **kw), text, num_gpus=g)
            wn = min(int(want[0]), kw["max_count"]) if mode != "lines" else int(want[0])
            if mode == "pos" and kw["max_count"] == abi.SIZE_MAX:
                if cnt != want[0] or not np.array_equal(pos, want[1]): fail("shards", seed, it, pats, kw, n, g, cnt, want[0])
            elif kw["max_count"] == abi.SIZE_MAX and cnt != want[0]: fail("shards count", seed, it, pats, kw, n, g, cnt, want[0])
        # a re-used plan: the second and third scan may run under the dense policy
        if mode == "pos" and kw["max_count"] == abi.SIZE_MAX and n >= 16384 and rng.rand() < 0.6:
            d = torch.from_numpy(text).cuda()
            cap = int(want[0]) + 3
            pos = torch.zeros(2 * max(cap, 1), dtype=torch.int64, device="cuda")
            plan = gpu.plan(abi.Params(pats, **kw))
            starve = n >= (1 << 20) and rng.rand() < 0.7  # a few workgroups: every wave scans many tickets, the one-pass rings fill
            for rep in range(4 if starve else 3):         # up as on a large text and the plan moves to the DENSE flavour
                pos.zero_()
                if starve: gpu.force_single_grid(int(rng.randint(2, 8)))
                try:
                    out = plan.scan(d.data_ptr(), n, 0, n, 0, pos.data_ptr(), cap)
                finally:
                    gpu.force_single_grid(0)
                g2 = pos[:2 * out.count].cpu().numpy().astype(np.uint64).reshape(-1, 2)
                if out.count != want[0] or not np.array_equal(g2, want[1]): fail("plan rep", rep, seed, it, pats, kw, n, out.count, want[0])
            plan.close()
print("soak done:", cases_run, "cases, failures:", bad, "- launches of the one-pass DENSE flavour:", gpu.tiny_dense_launches())
