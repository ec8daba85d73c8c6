"""The reference CLI itself, built with the GPU backend wired in as INTEGRATION.md describes
(integration/make_krep_gpu_cli.py -> oracle/_ref/krep_gpu_cli): with KREP_GPU=1 the scan runs on the MI355X,
without it the unchanged CPU functions run.  Outputs must be byte-identical (CPU side pinned to one thread:
krep's own multi-chunk path double-counts at chunk boundaries)."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "oracle", "_ref", "krep_gpu_cli")


def run(args, gpu, **extra):
    env = dict(os.environ)
    for k in ("KREP_GPU", "KREP_GPU_NUM", "KREP_GPU_MIN_BYTES", "KREP_GPU_COST_MODEL", "KREP_GPU_COST", "KREP_GPU_DISABLE", "KREP_GPU_INJECT_FAILURE"):
        env.pop(k, None)
    if gpu:
        env["KREP_GPU"] = "1"
        env["KREP_GPU_MIN_BYTES"] = "0"   # the small fixtures go through the backend too: neither the size threshold nor the
        env["KREP_GPU_COST_MODEL"] = "0"  # cost model keeps them on the CPU (krep_gpu_worthwhile has its own test)
    env.update({k: str(v) for k, v in extra.items()})
    r = subprocess.run([CLI] + args, env=env, capture_output=True, timeout=300)
    return r.returncode, r.stdout, r.stderr


@pytest.mark.skipif(not os.path.exists(CLI), reason="oracle/_ref/krep_gpu_cli not built (needs /root/reference)")
def test_cli_gpu_equals_cpu(tmp_path):
    import krep_amd
    e = krep_amd.load()
    big = e.generate_host(12 << 20, 0, 2, 99, b"Sherlock", 5000)
    big[1000:1008] = np.frombuffer(b"sherLOCK", dtype=np.uint8)
    f_big = tmp_path / "big.txt"
    f_big.write_bytes(big.tobytes())
    f_small = tmp_path / "small.txt"
    f_small.write_bytes(b"The quick brown fox\nSherlock Holmes and sherlock\nnothing here\nfoxSherlock fox\n")
    cases = [
        (["-c", "Sherlock"], f_big), (["-c", "-i", "sherlock"], f_big), (["-c", "-w", "Sherlock"], f_big),
        (["-c", "-o", "Sherlock"], f_big), (["-c", "e"], f_big), (["-c", "-e", "Sherlock", "-e", "the", "-e", "qz"], f_big),
        (["-c", "-m", "7", "Sherlock"], f_big), (["Sherlock"], f_small), (["-o", "fox"], f_small),
        (["-i", "SHERLOCK"], f_small), (["-w", "fox"], f_small), (["-e", "fox", "-e", "Holmes"], f_small),
        (["-c", "absent-pattern"], f_big),
        # overlapping matches of several patterns, printed one per line: the GPU records arrive already in the
        # formatter's (start, end) order (sorted in HBM, the patched CLI skips its qsort)
        (["-o", "-e", "Sherlock", "-e", "the", "-e", "he", "-e", "her"], f_big),
        (["-e", "Sherlock", "-e", "lock"], f_big),
        # round 2: memchr_short_search under -o (2-3 byte -i patterns) now runs on the GPU (candidate walk, krep.c:4495)
        (["-o", "-i", "th"], f_big), (["-o", "-i", "-w", "he"], f_big), (["-c", "-o", "-i", "ock"], f_big),
        # memchr_search with max_count a multiple of its 4096-entry batch: the displaced record (krep.c:3976-3991) must
        # still come out in file order although the patched CLI skips its qsort for GPU results
        (["-o", "-m", "4096", "e"], f_big),
        # -c through simd_sse42_search with a newline in the pattern (left to the CPU until round 3, now the walk of
        # kg_greedy.hip (3)): output unchanged, nothing on stderr
        (["-c", "fox\nSherlock"], f_small), (["-c", "e\nS"], f_big),
    ]
    for args, path in cases:
        cpu = run(["-t", "1", "--color=never"] + args + [str(path)], gpu=False)
        gpu = run(["-t", "1", "--color=never"] + args + [str(path)], gpu=True)
        assert b"krep-gpu:" not in gpu[2], gpu[2]
        assert gpu[0] == cpu[0] and gpu[1] == cpu[1], (args, cpu[:2], gpu[:2])
    # KREP_GPU_NUM=3: the operators shard every text over three devices of the process (one physical device here: a clique of
    # one), the shards' counters meet in the RCCL all-reduce issued from C; sequential families take the chained road
    for args, path in cases[:8] + cases[13:18]:
        cpu = run(["-t", "1", "--color=never"] + args + [str(path)], gpu=False)
        gpu3 = run(["-t", "1", "--color=never"] + args + [str(path)], gpu=True, KREP_GPU_NUM=3)
        assert b"krep-gpu:" not in gpu3[2], gpu3[2]
        assert gpu3[:2] == cpu[:2], ("KREP_GPU_NUM=3", args, cpu[:2], gpu3[:2])
    # and the GPU path really ran: the algorithm is no CPU function -> no chunking, same answer with -t 8
    multi = run(["-t", "8", "-c", "Sherlock", str(f_big)], gpu=True)
    single = run(["-t", "1", "-c", "Sherlock", str(f_big)], gpu=False)
    assert multi[1] == single[1]


@pytest.mark.skipif(not os.path.exists(CLI), reason="oracle/_ref/krep_gpu_cli not built (needs /root/reference)")
def test_config1_one_gib_count(tmp_path):
    """BASELINE.json configs[0] at full size: `krep -c -F Sherlock` over a 1 GiB file made of a 1 MiB block of
    <= 80-byte ASCII lines repeated 1024 times, `Sherlock` on one line in 100 — the reference CLI's own CPU path
    (one thread: its multi-chunk path double-counts) against the same CLI with the MI355X backend."""
    import random
    rng = random.Random(7)
    words = [b"the", b"quick", b"brown", b"fox", b"jumps", b"over", b"lazy", b"dog", b"Holmes", b"Watson", b"Baker", b"street"]
    block = bytearray()
    lines = 0
    while len(block) < (1 << 20) - 100:
        line = b" ".join(rng.choice(words) for _ in range(rng.randint(3, 12)))[:70]
        if lines % 100 == 17:
            line = line[:30] + b" Sherlock " + line[30:60]
        block += line + b"\n"
        lines += 1
    block += b"x" * ((1 << 20) - 1 - len(block)) + b"\n"
    assert len(block) == 1 << 20
    path = tmp_path / "cfg1.txt"
    with open(path, "wb") as f:
        for _ in range(1024):
            f.write(block)
    want_lines = sum(1 for ln in bytes(block).split(b"\n") if b"Sherlock" in ln) * 1024
    cpu = run(["-t", "1", "-c", "-F", "Sherlock", str(path)], gpu=False)
    gpu = run(["-t", "1", "-c", "-F", "Sherlock", str(path)], gpu=True)
    assert b"krep-gpu:" not in gpu[2], gpu[2]
    assert cpu[0] == 0 and gpu[0] == 0
    assert gpu[1] == cpu[1], (cpu[1][:100], gpu[1][:100])
    assert str(want_lines).encode() in gpu[1]
