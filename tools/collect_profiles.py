#!/usr/bin/env python3
"""Summarise gpurun_out/<tag>_<workload>_* (written by tools/profile_round.sh) into profiles/:
<tag>_<workload>_bench.json, _kernel_stats.csv, _pmc_fetch_write.csv and traffic.json (read by bench.py).

HBM bytes per launch of the dominant kernel = 2 x FETCH_SIZE + WRITE_SIZE, counters in KiB: FETCH_SIZE counts
32-byte-request units and under-reports 16 B/lane streams by 2x on gfx950 (MI355X_MICROARCH.md, HBM section); the
generator's WRITE_SIZE (exactly the bytes it fills) calibrates the unit.
"""
import csv
import glob
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT, PROF = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
DOMINANT = {"literal8": "kg::lit_scan", "memchr1": "kg::single_fused", "ac1000": "kg::ac_scan_kernel", "words1000": "kg::ac_scan_kernel"}
# the kernels of one step besides the dominant one (launched once per step: their bytes are added per step)
POST = {"literal8": ("kg::post_",), "memchr1": (), "ac1000": ("kg::post_",), "words1000": ("kg::post_",)}
# the sources whose change makes a workload's traffic figure stale (bench.py checks the hash before it quotes the figure)
KERNEL_SOURCES = {"literal8": ["kg_literal_dma.hip", "kg_literal.hip", "kg_post.hip", "kg_common.h"], "memchr1": ["kg_single.hip", "kg_tickets.h", "kg_common.h"],
                  "ac1000": ["kg_ac.hip", "kg_ac_common.h", "kg_ac_tables.h", "kg_post.hip", "kg_common.h"],
                  "words1000": ["kg_ac.hip", "kg_ac_anchor.hip", "kg_ac_common.h", "kg_ac_tables.h", "kg_post.hip", "kg_common.h"]}


def sources_sha(workload):
    import hashlib
    h = hashlib.sha256()
    for f in KERNEL_SOURCES[workload]:
        h.update(open(os.path.join(ROOT, "krep_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]
WARMUP_DROPPED = 3  # launches of every kernel left out of the steady-state statistics (tools/profile_round.sh runs --warmup 3)


def steady_stats(trace_csv, out_csv):
    """rocprofv3's *_kernel_stats.csv averages EVERY launch, the cold first one included (literal8 avg 5.274 ms
    with it, 5.144 without).  From the per-dispatch kernel trace: the same statistics without the first WARMUP_DROPPED
    launches of each kernel."""
    import statistics
    per = {}
    for row in csv.DictReader(open(trace_csv)):
        name = short(row["Kernel_Name"])
        if not name.startswith("kg::"):
            continue
        per.setdefault(name, []).append((int(row["Start_Timestamp"]), int(row["End_Timestamp"]) - int(row["Start_Timestamp"])))
    with open(out_csv, "w") as f:
        f.write("kernel,calls,warmup_dropped,avg_ns_steady,median_ns_steady,min_ns,max_ns_steady,avg_ns_all\n")
        for name, lst in sorted(per.items()):
            lst.sort()
            d = [x[1] for x in lst]
            st = d[WARMUP_DROPPED:] if len(d) > WARMUP_DROPPED else d
            f.write(f'"{name}",{len(d)},{len(d) - len(st)},{sum(st) / len(st):.0f},{statistics.median(st):.0f},{min(d)},{max(st)},'
                    f'{sum(d) / len(d):.0f}\n')


def short(name):
    return re.sub(r"^void ", "", name).split("(")[0]


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
    tpath = os.path.join(PROF, "traffic.json")
    traffic = json.load(open(tpath)) if os.path.exists(tpath) else {}
    for w, dom in DOMINANT.items():
        b = os.path.join(OUT, f"{tag}_{w}_bench.json")
        if not os.path.exists(b):
            continue
        line = [l for l in open(b) if l.startswith("{")][-1]
        bench = json.loads(line)
        open(os.path.join(PROF, f"{tag}_{w}_bench.json"), "w").write(line)
        ks = glob.glob(os.path.join(OUT, f"{tag}_{w}_kt", "**", "*kernel_stats.csv"), recursive=True)
        if ks:
            shutil.copy(ks[0], os.path.join(PROF, f"{tag}_{w}_kernel_stats.csv"))
        kt = glob.glob(os.path.join(OUT, f"{tag}_{w}_kt", "**", "*kernel_trace.csv"), recursive=True)
        if kt:
            steady_stats(kt[0], os.path.join(PROF, f"{tag}_{w}_kernel_steady.csv"))
            # the bench line printed BY THE TRACED PROCESS (round 6): its hipEvent kernel_ms and the trace's per-kernel averages are
            # one process, one placement draw — the pair to recompute the roofline fraction from (the plain <tag>_<w>_bench.json is
            # another process and draws its own placement, 1-2 % apart)
            ktlog = os.path.join(OUT, f"{tag}_{w}_kt.log")
            tl = [l for l in open(ktlog) if l.startswith("{")] if os.path.exists(ktlog) else []
            if tl:
                open(os.path.join(PROF, f"{tag}_{w}_bench_under_trace.json"), "w").write(tl[-1])
                tr = json.loads(tl[-1])["roofline"]
                step_ns = 0.0
                for row in csv.DictReader(open(os.path.join(PROF, f"{tag}_{w}_kernel_steady.csv"))):
                    if row["kernel"].startswith(dom) and int(row["calls"]) >= 10 or any(row["kernel"].startswith(p) for p in POST[w]):
                        step_ns += float(row["avg_ns_steady"])
                print(f"{w}: traced process: hipEvent kernel_ms {tr['kernel_ms']:.4f} (frac {tr['frac']:.4f}) vs kernel-trace steady sum {step_ns / 1e6:.4f} ms "
                      f"({(tr['kernel_ms'] * 1e6 / step_ns - 1) * 100:+.2f} %)")
        agg = {}
        for c in ("FETCH_SIZE", "WRITE_SIZE"):
            for f in glob.glob(os.path.join(OUT, f"{tag}_{w}_{c}", "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    k = (short(row["Kernel_Name"]), row["Counter_Name"])
                    s = agg.setdefault(k, [0, 0.0])
                    s[0] += 1
                    s[1] += float(row["Counter_Value"])
        if not agg:
            continue
        with open(os.path.join(PROF, f"{tag}_{w}_pmc_fetch_write.csv"), "w") as f:
            f.write("kernel,counter,dispatches,avg_value_KiB\n")
            for (k, c), (n, v) in sorted(agg.items()):
                if k.startswith("kg::") or k.startswith("synth"):
                    f.write(f'"{k}",{c},{n},{v / n:.3f}\n')
        # ONE instantiation: the one of this family with the most dispatches x bytes (round 5 matched the family prefix and so
        # averaged lit_scan<8,...> with the lit_scan<1,...> launches of bench.py's placement probe)
        fam = {}
        for (k, c), v in agg.items():
            if k.startswith(dom) and c == "FETCH_SIZE":
                fam[k] = v[1]
        inst = max(fam, key=fam.get) if fam else None
        fk = [(k, v) for (k, c), v in agg.items() if k == inst and c == "FETCH_SIZE"]
        wk = [(k, v) for (k, c), v in agg.items() if k == inst and c == "WRITE_SIZE"]
        # the step's other kernels (post-pass): average bytes per dispatch x dispatches per dominant dispatch
        post_f = post_w = 0.0
        n_dom = sum(v[0] for _, v in fk) or 1
        for (k, c), v in agg.items():
            if any(k.startswith(p) for p in POST[w]):
                if c == "FETCH_SIZE":
                    post_f += v[1] / n_dom
                elif c == "WRITE_SIZE":
                    post_w += v[1] / n_dom
        if fk and wk:
            fetch = sum(v[1] for _, v in fk) / sum(v[0] for _, v in fk)
            write = sum(v[1] for _, v in wk) / sum(v[0] for _, v in wk)
            alg = bench["roofline"]["algorithmic_bytes_per_launch"]
            hbm = int((2 * fetch + write) * 1024)
            note = ""
            filt = []
            for f in glob.glob(os.path.join(OUT, f"{tag}_{w}_FETCH_SIZE_filter", "**", "*counter_collection.csv"), recursive=True):
                filt += [float(r["Counter_Value"]) for r in csv.DictReader(open(f))
                         if short(r["Kernel_Name"]) == inst and r["Counter_Name"] == "FETCH_SIZE"]
            if filt:
                # streamed reads (the filter-only pass) are under-reported 2x, the verify stage's gathers are exact
                # (tools/ubench/fetch_calib.hip): traffic = 2 x filter + 1 x (full - filter) + writes
                ff = sum(filt) / len(filt)
                hbm = int((2 * ff + max(0.0, fetch - ff) + write) * 1024)
                note = (f"; streamed part {ff:.0f} KiB (KREP_GPU_AC_NOVERIFY pass) doubled, the verify stage's gathers "
                        f"({fetch - ff:.0f} KiB) counted as reported (profiles/r03_fetch_size_calibration.txt)")
            hbm_step = hbm + int((2 * post_f + post_w) * 1024)  # (the post-pass reads are counted double too: an upper bound)
            traffic[w] = {
                "hbm_bytes_per_launch": hbm, "hbm_bytes_per_step_with_post_pass": hbm_step,
                "FETCH_SIZE_KiB": fetch, "WRITE_SIZE_KiB": write, "post_pass_FETCH_SIZE_KiB": post_f, "post_pass_WRITE_SIZE_KiB": post_w,
                "algorithmic_bytes": alg, "ratio": round(hbm / alg, 4), "ratio_with_post_pass": round(hbm_step / alg, 4),
                "kernel": inst, "measured_by": tag,
                "kernel_sources": KERNEL_SOURCES[w], "kernel_sources_sha": sources_sha(w),
                "method": f"tools/profile_round.sh {tag}: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate "
                          f"passes of `python bench.py --workload {w} --steps 2 --warmup 1 --no-cpu-baseline`; "
                          "FETCH_SIZE doubled (16 B/lane streams on gfx950, MI355X_MICROARCH.md HBM section); counters "
                          "in KiB (the generator's WRITE_SIZE calibrates to exactly the bytes it fills)" + note}
            bench["roofline"]["traffic"] = hbm          # the PMC passes of this very run
            open(os.path.join(PROF, f"{tag}_{w}_bench.json"), "w").write(json.dumps(bench) + "\n")
            print(w, "traffic ratio", traffic[w]["ratio"], "value", bench["value"], "frac", bench["roofline"]["frac"])
    json.dump(traffic, open(tpath, "w"), indent=1)


if __name__ == "__main__":
    main()
