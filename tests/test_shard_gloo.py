"""The N > 1 path on CPU: world_size-2 (and 3) gloo ranks exercise the shard partition, start-offset
ownership, the single count all-reduce and the line-carry combine of tests/shard_model.py."""
import os
import subprocess
import sys

import pytest

import shard_model as shard

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


def test_bench_spawns_its_own_ranks_dry_run():
    """`python bench.py --gpus N` outside torchrun starts the N ranks itself (the driver's plain invocation died
    on an assert).  CPU dry run: gloo, no scan, no number — rendezvous on 127.0.0.1, barrier, the one all-reduce of the counts,
    MAX-over-ranks timing and the single JSON line of rank 0."""
    import json
    root = os.path.dirname(HERE)
    env = dict(os.environ, MASTER_PORT="29711")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "3",
                        "--warmup", "1", "--gib", "0.01"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["dry_run"] is True and j["steps"] == 3 and j["value"] is None
    assert j["config"]["matches"] == 2 * (int(0.01 * (1 << 30)) // 10000)
    # a mismatching launch is an error message, not an assert trace
    env2 = dict(env, WORLD_SIZE="3", RANK="0")
    r2 = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo"], env=env2,
                        capture_output=True, text=True, timeout=60)
    assert r2.returncode == 2 and "WORLD_SIZE=3" in r2.stderr


def test_bench_eight_ranks_dry_run():
    """The driver's 8-GPU launch line, on CPU: `torch.distributed.run --nproc-per-node 8 ... bench.py
    --gpus 8 --backend gloo` — eight ranks rendezvous on 127.0.0.1, barrier, ONE all-reduce of the counts per step, MAX over
    ranks, one JSON line from rank 0 whose closed-form count is the sum over the eight contiguous shards."""
    import json
    root = os.path.dirname(HERE)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=8", "--master-addr", "127.0.0.1",
           "--master-port", "29788", os.path.join(root, "bench.py"), "--gpus", "8", "--backend", "gloo", "--steps", "4",
           "--warmup", "1", "--gib", "0.01"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 8 and j["dry_run"] is True and j["steps"] == 4 and j["scaling"] == "weak"
    assert j["config"]["matches"] == 8 * (int(0.01 * (1 << 30)) // 10000)
