"""TEST SUPPORT (moved out of the package in round 5: the product shards in C, kg_ops.hip run_pieces): a Python model of the
rank-level sharding for the one-process-per-GPU path (torch.distributed; backend "nccl" == RCCL on ROCm,
"gloo" in the CPU tests).  The haystack is sharded by CONTIGUOUS chunk; a rank reports a match iff its
START lies in the rank's window (start-offset ownership) and reads `halo` bytes past the
window so that such matches complete.  The only collective on the data path is ONE all-reduce of the
per-rank counters; -c needs one all-gather of four small integers per rank for the line carry.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Sequence


def shard_bounds(total_len: int, world: int, rank: int) -> tuple[int, int]:
    """[lo, hi) owned by `rank`: chunk = ceil(N / world), like krep's chunking (krep.c:2816-2829)."""
    chunk = (total_len + world - 1) // world
    lo = min(total_len, rank * chunk)
    return lo, min(total_len, lo + chunk)


def halo_bytes(max_pattern_len: int) -> int:
    """Bytes of context a shard needs on each side: pattern_len-1 to complete straddling matches, +1 for the
    -w neighbour test, +1 of slack (kg_multi.hip uses the same figure)."""
    return max_pattern_len + 1


@dataclass
class LineSummary:
    line_count: int
    head_line_hit: bool
    tail_line_hit: bool
    has_newline: bool


def combine_line_counts(shards: Sequence[LineSummary]) -> int:
    """Python twin of krep_gpu_combine_line_counts(): fold the shard summaries left to right."""
    total, open_ = 0, False
    for s in shards:
        total += s.line_count
        if open_ and s.head_line_hit:
            total -= 1
        open_ = s.tail_line_hit if s.has_newline else (open_ or s.head_line_hit)
    return total


def allreduce_counts(values: Sequence[int], device=None):
    """The single all-reduce of the data path.  Returns the summed values as Python ints."""
    import torch
    import torch.distributed as dist
    t = torch.tensor(list(values), dtype=torch.int64, device=device)
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t)
    return [int(x) for x in t.tolist()]


def allgather_line_summaries(mine: LineSummary, device=None) -> list[LineSummary]:
    import torch
    import torch.distributed as dist
    t = torch.tensor([mine.line_count, int(mine.head_line_hit), int(mine.tail_line_hit), int(mine.has_newline)],
                     dtype=torch.int64, device=device)
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return [mine]
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [LineSummary(int(o[0]), bool(o[1]), bool(o[2]), bool(o[3])) for o in out]


def allgather_ints(values: Sequence[int], device=None) -> list[list[int]]:
    """One all-gather of a few integers per rank (the boundary records of the sequential families)."""
    import torch
    import torch.distributed as dist
    t = torch.tensor(list(values), dtype=torch.int64, device=device)
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return [[int(x) for x in t.tolist()]]
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [[int(x) for x in o.tolist()] for o in out]


def true_resumes(lows: Sequence[int], resume_used: Sequence[int], resume_out: Sequence[int]) -> tuple[list[int], list[bool]]:
    """The exchange step of the chained families (krep_gpu_seq_carry_t::resume, kg_ops.hip::run_pieces): given, per shard in
    text order, the record it was scanned with and the record it left, the record each shard SHOULD have been scanned with and
    whether its scan has to be repeated.  A repeated shard's own record is unknown until it has been re-scanned, so the walk
    stops at the first stale shard: call again after re-scanning it (a text that is one cluster re-scans every shard)."""
    true_in, stale = [], []
    cur = 0
    for lo, used, out in zip(lows, resume_used, resume_out):
        true_in.append(cur)
        bad = max(used, lo) != max(cur, lo)
        stale.append(bad)
        if bad:
            break
        cur = max(cur, out)
    return true_in, stale
