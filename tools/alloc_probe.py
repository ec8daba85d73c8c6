"""Allocation-pattern probe for the mixed read + record-write workload (single byte, 1 % hits): the same scan with the text and
the record buffer (a) in two allocations, text first; (b) in ONE allocation, records behind the text; (c) two allocations,
records first; (d) ONE allocation, records in front — round robin in one process, so that a box property shows as a constant
and an allocation-pattern property as a pattern.  usage: python tools/alloc_probe.py [GiB] [rounds]"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import krep_amd
from krep_amd import abi

gib = float(sys.argv[1]) if len(sys.argv) > 1 else 32.0
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 2
n = int(gib * (1 << 30))
e = krep_amd.load()
cap = n // 80 + 4096
rec_bytes = 16 * cap


def measure(d_text, d_pos, label):
    e.generate(d_text, n, 0, 3, 20260925, b"#", 0)
    plan = e.plan(abi.Params([b"#"]))
    t = [plan.scan(d_text, n, 0, n, 0, d_pos, cap, time_it=True).kernel_ms for _ in range(5)][1:]
    lit = e.plan(abi.Params([b"Sherlock"]))
    t8 = [lit.scan(d_text, n, 0, n, 0, d_pos, cap, time_it=True).kernel_ms for _ in range(4)][1:]
    plan.close(); lit.close()
    print(f"{label:46s} single byte+records {statistics.median(t):6.3f} ms   8-byte literal {statistics.median(t8):6.3f} ms", flush=True)


def al(nbytes):
    return torch.empty(nbytes, dtype=torch.uint8, device="cuda")


def a16(p):
    return p + (-p) % 256


for rnd in range(rounds):
    text = al(n + 64); pos = al(rec_bytes)
    measure(a16(text.data_ptr()), a16(pos.data_ptr()), f"[{rnd}] two allocations, text first")
    del text, pos; torch.cuda.empty_cache()
    arena = al(n + rec_bytes + 4096)
    measure(a16(arena.data_ptr()), a16(arena.data_ptr() + n + 1024), f"[{rnd}] ONE allocation, records behind the text")
    del arena; torch.cuda.empty_cache()
    pos = al(rec_bytes); text = al(n + 64)
    measure(a16(text.data_ptr()), a16(pos.data_ptr()), f"[{rnd}] two allocations, records first")
    del text, pos; torch.cuda.empty_cache()
    arena = al(n + rec_bytes + 4096)
    measure(a16(arena.data_ptr() + rec_bytes + 1024), a16(arena.data_ptr()), f"[{rnd}] ONE allocation, records in front")
    del arena; torch.cuda.empty_cache()
    text = al(n + 64); junk = al(3 << 30); pos = al(rec_bytes)
    measure(a16(text.data_ptr()), a16(pos.data_ptr()), f"[{rnd}] two allocations, 3 GiB of junk between")
    del text, pos, junk; torch.cuda.empty_cache()
