"""Worker of tests/test_shard_gloo.py: world_size ranks over gloo on CPU.  Each rank scans ITS shard (the
oracle stands in for the GPU scan — test infrastructure) with start-offset ownership; the counts meet in
one all-reduce, the line carries in one all-gather.  Rank 0 checks against the single-chunk oracle."""
import os
import sys

import numpy as np
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as ol  # noqa: E402
from krep_amd import abi  # noqa: E402
import shard_model as shard  # noqa: E402


def owned_scan(o, algo, pats, kw, text, lo, hi, halo):
    """matches with start in [lo, hi), scanning text[lo-halo : hi+halo] only."""
    b0, b1 = max(0, lo - halo), min(text.size, hi + halo)
    ret, pos = o.call(algo, abi.Params(pats, **kw), text[b0:b1].copy())
    pos = pos + np.uint64(b0)
    keep = (pos[:, 0] >= lo) & (pos[:, 0] < hi)
    return pos[keep]


def line_summary(text, lo, hi, starts):
    seg = text[lo:hi]
    nl = np.flatnonzero(seg == 10) + lo
    starts = np.asarray(starts, dtype=np.int64)
    line_id = np.searchsorted(nl, starts, side="left")  # number of newlines strictly before the start
    has_nl = nl.size > 0
    head = bool(np.any(starts <= nl[0])) if has_nl else starts.size > 0
    tail = bool(np.any(starts > nl[-1])) if has_nl else starts.size > 0
    return shard.LineSummary(int(np.unique(line_id).size), head, tail, has_nl)


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    rng = np.random.RandomState(2026)
    text = rng.choice(np.frombuffer(b"abcd \n", dtype=np.uint8), size=200_001).astype(np.uint8)
    o = ol.oracle()
    fails = 0
    for pats, algo in (([b"abcd"], abi.RA_BMH), ([b"d"], abi.RA_MEMCHR), ([b"ab", b"bcd", b"d a"], abi.RA_AHO_CORASICK)):
        lmax = max(len(p) for p in pats)
        lo, hi = shard.shard_bounds(text.size, world, rank)
        mine = owned_scan(o, algo, pats, {}, text, lo, hi, shard.halo_bytes(lmax))
        total, = shard.allreduce_counts([len(mine)])
        summ = shard.allgather_line_summaries(line_summary(text, lo, hi, mine[:, 0]))
        lines = shard.combine_line_counts(summ)
        if rank == 0:
            want_n, want_pos = o.call(algo, abi.Params(pats), text)
            want_lines, _ = o.call(algo, abi.Params(pats, count_lines=True), text)
            if total != want_n or lines != want_lines:
                print("MISMATCH", pats, total, want_n, lines, want_lines, flush=True)
                fails += 1
    # ---- the chained families across ranks (krep_gpu_seq_carry_t::resume): greedy non-overlapping selection of a bordered
    # pattern.  Every rank scans its shard optimistically (nothing consumed in front of it), the boundary records meet in ONE
    # all-gather, and a rank whose assumption was wrong scans again — until no record changes (here the oracle's
    # simd_sse42_search on the slice that starts at the resume point stands in for the GPU walk, which owns the starts from
    # max(lo, resume): the reference's loop after `advance` IS a fresh scan from that point).
    pat = b"aa"
    m = len(pat)
    ctext = text.copy()
    ctext[: ctext.size // 2] = ord("a")  # one giant cluster across the first cuts, random text behind it
    lo, hi = shard.shard_bounds(ctext.size, world, rank)

    def greedy_scan(resume):
        s0 = min(max(lo, resume), hi)
        ret, pos = o.call(abi.RA_SSE42, abi.Params([pat]), ctext[s0:min(ctext.size, hi + m)].copy())
        pos = pos + np.uint64(s0)
        pos = pos[pos[:, 0] < hi]
        return pos, (int(pos[-1, 0]) + m if len(pos) else 0)

    used = 0
    mine, out = greedy_scan(used)
    for _ in range(world + 1):
        recs = shard.allgather_ints([lo, used, out])
        true_in, stale = shard.true_resumes([r[0] for r in recs], [r[1] for r in recs], [r[2] for r in recs])
        if not any(stale):
            break
        if len(stale) > rank and stale[rank]:
            used = true_in[rank]
            mine, out = greedy_scan(used)
    total, = shard.allreduce_counts([len(mine)])
    if rank == 0:
        want_n, want_pos = o.call(abi.RA_SSE42, abi.Params([pat]), ctext)
        if total != want_n or any(stale):
            print("MISMATCH chained", total, want_n, stale, flush=True)
            fails += 1
    if rank == 0:
        print("DIST_OK" if fails == 0 else "DIST_FAIL", flush=True)
    dist.destroy_process_group()
    sys.exit(1 if fails else 0)


if __name__ == "__main__":
    main()
