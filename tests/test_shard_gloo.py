"""The N > 1 path on CPU: world_size-2 (and 3) gloo ranks exercise the shard partition, start-offset
ownership, the single count all-reduce and the line-carry combine of krep_amd/shard.py."""
import os
import subprocess
import sys

import pytest

from krep_amd import shard

HERE = os.path.dirname(os.path.abspath(__file__))


def test_shard_bounds_partition():
    for n in (0, 1, 7, 4096, 100_003):
        for w in (1, 2, 3, 8):
            spans = [shard.shard_bounds(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            for a, b in zip(spans[:-1], spans[1:]):
                assert a[1] == b[0] and a[0] <= a[1]


def test_combine_line_counts_cases():
    L = shard.LineSummary
    # one line spanning three shards with a match in each: counted once
    assert shard.combine_line_counts([L(1, True, True, False)] * 3) == 1
    # newline inside the middle shard, matches on both sides of it
    assert shard.combine_line_counts([L(1, True, True, False), L(2, True, True, True), L(1, True, True, False)]) == 2
    # middle shard without matches and without newline keeps the line open
    assert shard.combine_line_counts([L(1, True, True, True), L(0, False, False, False), L(1, True, False, True)]) == 1
    # a newline-only shard closes it
    assert shard.combine_line_counts([L(1, True, True, True), L(0, False, False, True), L(1, True, False, True)]) == 2


@pytest.mark.parametrize("world", [2, 3])
def test_gloo_ranks(world):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(29650 + world), os.path.join(HERE, "_dist_worker.py")]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "DIST_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
