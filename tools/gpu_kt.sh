#!/bin/bash
# per-kernel times of one library variant: tools/gpu_kt.sh <kind> <mode> <variant.so> [...]
R=${GRAFT_REPO_ROOT:-$(pwd)}; K=$1; M=$2; shift 2
cd /tmp; export TMPDIR=/tmp
for v in "$@"; do
  rm -rf /tmp/kt_$v
  AB_REPS=6 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$v -o kt -- python $R/tools/ab_bench.py 32 $K $M $v > /tmp/kt_$v.log 2>&1
  grep median /tmp/kt_$v.log | cut -c1-100
  python3 - /tmp/kt_$v <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "kg::" in r["Name"] and float(r["AverageNs"]) > 2e4:
            print("    %-44s calls %3s avg %.3f min %.3f max %.3f" % (r["Name"][:44], r["Calls"], float(r["AverageNs"]) / 1e6, float(r["MinNs"]) / 1e6, float(r["MaxNs"]) / 1e6))
PY
done
