#!/bin/bash
# usage (on the GPU box): tools/pmc_pass.sh <tag> "<counters>" <bench args...>
# one rocprofv3 --pmc pass of bench.py; per-kernel sums printed and left in gpurun_out/<tag>/
set -u
TAG=$1; CTRS=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --pmc $CTRS --output-format csv -d $O -o pmc -- python $R/bench.py "$@" --steps 2 --warmup 1 --no-cpu-baseline --no-extra > $O/run.log 2>&1
python - "$O" <<'PY' | tee $O/summary.txt
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = (r["Kernel_Name"].split("(")[0][-48:], r["Counter_Name"])
        agg[k][0] += 1
        agg[k][1] += float(r["Counter_Value"])
for (k, c), (n, v) in sorted(agg.items()):
    if "kg::" in k:
        print(f"{k:50s} {c:28s} n={n} avg={v / n:.4g}")
PY
tail -3 $O/run.log | cut -c1-300
