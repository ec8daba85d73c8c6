#!/bin/bash
# per-placement counters: does the L2 translation cache (UTCL2) or the DRAM credit path tell a slow placement from a fast one?
ulimit -c 0
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; T=$(date +%s)
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --pmc GRBM_UTCL2_BUSY TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum TCC_EA0_WRREQ_STALL_sum --output-format csv -d $O/boxprobe6_$T -o pmc -- python $R/tools/placement_probe_m1.py 32 8 > $O/boxprobe6_$T.log 2>&1
cd $R
python - "$O/boxprobe6_$T" <<'PY' | tee -a $O/boxprobe6_$T.log
import csv, glob, sys, collections
rows = []
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "single_fused" in r["Kernel_Name"]:
            rows.append((int(r["Dispatch_Id"]), r["Counter_Name"], float(r["Counter_Value"])))
ids = sorted({d for d, _, _ in rows})
groups = [ids[i:i + 6] for i in range(0, len(ids), 6)]
for gi, g in enumerate(groups):
    agg = collections.defaultdict(list)
    for d, c, v in rows:
        if d in g:
            agg[c].append(v)
    print(f"placement group {gi}: " + "  ".join(f"{c}={sum(v) / len(v):.4g}" for c, v in sorted(agg.items())))
PY
grep "placement" $O/boxprobe6_$T.log | head -30
