#!/bin/bash
# usage (on the GPU box): tools/kt_cmd.sh <tag> <command ...> — rocprofv3 kernel trace of a command, every kg:: launch listed in order
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O -o kt -- "$@" > $O/run.log 2>&1
python - "$O" <<'PY' | tee $O/summary.txt
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
for r in rows:
    n = r["Kernel_Name"].split("(")[0].replace("void ", "")[:64]
    if "kg::" in n and "synth" not in n:
        print(f"{(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6:9.3f} ms  {n}")
PY
tail -4 $O/run.log | cut -c1-200
