#!/bin/bash
# which kernel carries the process-to-process bimodality of the offsets-producing scan?
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; cd /tmp; export TMPDIR=/tmp
for i in 1 2 3 4 5 6; do
  rm -rf /tmp/kt$i
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt$i -o kt -- python $R/tools/quick_bench.py 32 2 7 > /tmp/kt$i.log 2>&1
  grep "mode=pos" /tmp/kt$i.log | sed 's/count=.*overflow=0//'
  python3 - /tmp/kt$i <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "kg::" in r["Name"]:
            print("    %-40s calls %3s avg %.3f min %.3f max %.3f" % (r["Name"][:40], r["Calls"], float(r["AverageNs"]) / 1e6, float(r["MinNs"]) / 1e6, float(r["MaxNs"]) / 1e6))
PY
done
