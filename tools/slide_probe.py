"""Placement probe inside ONE allocation: the single-byte workload (34.4 GB read + 5.5 GB of records
written) with the record buffer's base slid against the text in steps from 4 KiB to 64 MiB, and the text's base slid against
the allocation.  A channel / bank-hash interaction between the read stream and the write front shows up as a pattern that is
periodic in the offset; physical page placement does not (it does not change when a pointer moves inside its allocation).
usage: python tools/slide_probe.py [GiB]"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import krep_amd
from krep_amd import abi

gib = float(sys.argv[1]) if len(sys.argv) > 1 else 32.0
n = int(gib * (1 << 30))
e = krep_amd.load()
cap = n // 80 + 4096                                  # ~1 % hits + slack
slack = 160 << 20
arena = torch.empty(n + 64 + 16 * cap + 2 * slack, dtype=torch.uint8, device="cuda")
base = arena.data_ptr()
base += (-base) % (2 << 20)                           # 2 MiB aligned start


def run(text_off, pos_off, reps=4):
    d_text = base + text_off
    d_pos = base + slack + n + 64 + pos_off
    d_pos += (-d_pos) % 16
    e.generate(d_text, n, 0, 3, 20260925, b"#", 0)
    plan = e.plan(abi.Params([b"#"]))
    t = []
    for r in range(reps):
        out = plan.scan(d_text, n, 0, n, 0, d_pos, cap, time_it=True)
        if r:
            t.append(out.kernel_ms)
    plan.close()
    return statistics.median(t), min(t), out.count


print(f"# single byte, {gib:g} GiB, one allocation of {arena.numel() / 2**30:.1f} GiB at {base:#x}")
print("# record base slid (text fixed at +0)")
for off in [0, 4 << 10, 8 << 10, 16 << 10, 32 << 10, 64 << 10, 128 << 10, 256 << 10, 512 << 10, 1 << 20, (1 << 20) + (4 << 10), 2 << 20,
            3 << 20, 4 << 20, 8 << 20, 16 << 20, 32 << 20, 64 << 20, (64 << 20) + (512 << 10)]:
    med, mn, cnt = run(0, off)
    print(f"pos +{off / 1024:10.0f} KiB   median {med:6.3f} ms   min {mn:6.3f}   count {cnt}", flush=True)
print("# text base slid (records fixed)")
for off in [4 << 10, 64 << 10, 1 << 20, 2 << 20, 16 << 20, 128 << 20]:
    med, mn, cnt = run(off, 0)
    print(f"text +{off / 1024:9.0f} KiB   median {med:6.3f} ms   min {mn:6.3f}   count {cnt}", flush=True)
# and the same through fresh allocations of the record buffer (round 3's observation), for contrast
del arena
torch.cuda.empty_cache()
text = torch.empty(n + 64, dtype=torch.uint8, device="cuda")
e.generate(text.data_ptr(), n, 0, 3, 20260925, b"#", 0)
plan = e.plan(abi.Params([b"#"]))
junk = []
for i in range(6):
    pos = torch.empty(2 * cap, dtype=torch.int64, device="cuda")
    t = [plan.scan(text.data_ptr(), n, 0, n, 0, pos.data_ptr(), cap, time_it=True).kernel_ms for _ in range(4)][1:]
    print(f"fresh record buffer {i} at {pos.data_ptr():#x}: median {statistics.median(t):6.3f} ms", flush=True)
    junk.append(torch.empty((i + 1) * (37 << 20), dtype=torch.uint8, device="cuda"))
    del pos
