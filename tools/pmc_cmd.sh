#!/bin/bash
# usage (on the GPU box): tools/pmc_cmd.sh <tag> "<counters>" <kernel substring> <cells per launch> <command ...>
# one rocprofv3 --pmc pass of a command; per-kernel averages (and per 1-KiB cell) printed, raw output in gpurun_out/<tag>/
set -u
TAG=$1; CTRS=$2; KSUB=$3; CELLS=$4; shift 4
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --pmc $CTRS --output-format csv -d $O -o pmc -- "$@" > $O/run.log 2>&1
python - "$O" "$KSUB" "$CELLS" <<'PY' | tee $O/summary.txt
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if sys.argv[2] in r["Kernel_Name"]:
            a = agg[r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
cells = float(sys.argv[3])
for c, (n, v) in sorted(agg.items()):
    print(f"{c:28s} launches={n} avg={v / n:.4g}  per cell {v / n / cells:.2f}")
PY
