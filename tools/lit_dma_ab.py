"""A/B of the LDS-DMA literal kernel (kg_literal_dma.hip) against the register-load kernel (kg_literal.hip) IN ONE PROCESS on the
same HBM buffers: $KREP_GPU_LIT_NO_DMA is read per launch.   usage: python tools/lit_dma_ab.py <gib> [reps]
       python tools/lit_dma_ab.py <gib> <reps> rate   -> the two kernels against the share of 1-KiB cells that hold the pattern's first byte
                                                         (i.i.d. text with that byte sprinkled in), and what the library's own choice does:
                                                         profiles/r06_ldsdma_byte_rate.txt"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import krep_amd
from krep_amd import abi

gib = float(sys.argv[1]) if len(sys.argv) > 1 else 32.0
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 7
n = int(gib * (1 << 30))
e = krep_amd.load()
buf = torch.empty(n + 64, dtype=torch.uint8, device="cuda")
cap = n // 2000 + 4096
pos = torch.empty(2 * cap, dtype=torch.int64, device="cuda")
base = b"Sherlock Holmes"
if len(sys.argv) > 3 and sys.argv[3] == "rate":
    pat = base[:8]
    e.generate(buf.data_ptr(), n, 0, 2, 42, pat, 10000)
    print(f"# {gib:g} GiB i.i.d. a-z/blank/newline with '{pat.decode()}' planted every 10000 bytes and the byte 'S' sprinkled in at rate f; match offsets produced;\n"
          f"# median kernel ms of {reps} launches each, alternating in one process.  dma = kg_literal_dma.hip kept whatever the text ($KREP_GPU_LIT_DMA_KEEP),\n"
          f"# regs = kg_literal.hip ($KREP_GPU_LIT_NO_DMA), auto = the library's choice (a fresh plan: the sample before its first launch decides)")
    done = 0.0
    for f in (0.0, 1e-5, 3e-5, 1e-4, 2e-4, 5e-4, 1e-3, 3e-3, 1e-2, 3e-2):
        step = 1 << 28
        for o in range(0, n, step):  # (top up from `done` to f)
            k = min(step, n - o)
            m_ = torch.rand(k, device="cuda") < (f - done) / max(1e-12, 1.0 - done)
            buf[o:o + k][m_] = ord("S")
            del m_
        done = f
        torch.cuda.synchronize()
        ENV = {"dma": {"KREP_GPU_LIT_DMA_KEEP": "1"}, "regs": {"KREP_GPU_LIT_NO_DMA": "1"}, "auto": {}}
        plans = {k: e.plan(abi.Params([pat])) for k in ENV}
        t = {k: [] for k in ENV}
        cnt = {}
        for rep in range(reps + 1):
            for which, env in ENV.items():
                os.environ.update(env)
                out = plans[which].scan(buf.data_ptr(), n, 0, n, 0, pos.data_ptr(), cap, time_it=True)
                for k in env:
                    os.environ.pop(k, None)
                cnt[which] = out.count
                if rep:
                    t[which].append(out.kernel_ms)
        st_keep, st_auto = plans["dma"].literal_dma_state(), plans["auto"].literal_dma_state()
        for pl_ in plans.values():
            pl_.close()
        med = {k: statistics.median(v) for k, v in t.items()}
        print(f"f = {f:7.5f}  cells that hold 'S': {100 * st_keep[2]:5.1f} % (counted by the kernel; the sample said {100 * st_auto[2]:5.1f} %)   " +
              "   ".join(f"{k} {v:6.3f} ({n / v / 1e6:5.0f})" for k, v in med.items()) +
              f"   auto chose {'regs' if st_auto[1] else 'dma'}   count {cnt['dma']}  same: {len(set(cnt.values())) == 1}", flush=True)
    sys.exit(0)
print(f"# {gib:g} GiB, median of {reps} launches each, alternating in one process; ms (GB/s)   [dma = kg_literal_dma.hip (where eligible), regs = kg_literal.hip, +pf = with the rare-first-byte prefilter]")
for m, kw, label in ((8, {}, "m=8 offsets"), (8, dict(count_lines=True, only_match=True), "m=8 count"), (8, dict(case_sensitive=False), "m=8 -i offsets"),
                     (8, dict(whole_word=True), "m=8 -w offsets"), (5, {}, "m=5 offsets"), (4, {}, "m=4 offsets"), (3, {}, "m=3 offsets"), (2, {}, "m=2 offsets")):
    pat = base[:m]
    e.generate(buf.data_ptr(), n, 0, 2, 42, pat, 10000)
    torch.cuda.synchronize()
    want_pos = "count" not in label
    plan = e.plan(abi.Params([pat], **kw))
    ENV = {"dma": {}, "regs+pf": {"KREP_GPU_LIT_NO_DMA": "1"}, "regs": {"KREP_GPU_LIT_NO_DMA": "1", "KREP_GPU_LIT_NO_PREFILTER": "1"}}
    t = {k: [] for k in ENV}
    cnt = {}
    for rep in range(reps + 1):
        for which, env in ENV.items():
            os.environ.update(env)
            out = plan.scan(buf.data_ptr(), n, 0, n, 0, pos.data_ptr() if want_pos else 0, cap if want_pos else 0, time_it=True)
            for k in env:
                os.environ.pop(k, None)
            cnt[which] = (out.count, int(pos[: 2 * min(out.stored, 1000)].sum().item()) if want_pos else 0)
            if rep:
                t[which].append(out.kernel_ms)
    plan.close()
    med = {k: statistics.median(v) for k, v in t.items()}
    print(f"{label:16s} " + "   ".join(f"{k} {v:6.3f} ({n / v / 1e6:5.0f})" for k, v in med.items()) +
          f"   count {cnt['dma'][0]}   same result: {len(set(cnt.values())) == 1}   dma launches so far {e.literal_dma_launches()}", flush=True)
