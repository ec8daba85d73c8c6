#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/r03h_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r03h_pytest.log
tail -12 gpurun_out/r03h_pytest.log | cut -c1-400
timeout 300 python tools/literal_sweep.py 32 64,128 > gpurun_out/r03h_sweep_long.txt 2>&1; tail -2 gpurun_out/r03h_sweep_long.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $GRAFT_REPO_ROOT/gpurun_out/r03h_counters_avail.txt 2>&1
grep -i -E "utcl|tlb|WRREQ_STALL|EA_WRREQ|TCC_EA" $GRAFT_REPO_ROOT/gpurun_out/r03h_counters_avail.txt | cut -c1-160 | head -40
hipcc --offload-arch=gfx950 -O3 -o /tmp/fc $GRAFT_REPO_ROOT/tools/ubench/fetch_calib.hip && timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r03h_fetch_calib -o pmc -- /tmp/fc > $GRAFT_REPO_ROOT/gpurun_out/r03h_fetch_calib.log 2>&1
python - <<'PY'
import csv, glob, os
for f in glob.glob(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r03h_fetch_calib/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        print(r["Kernel_Name"][:40], r["Counter_Name"], r["Counter_Value"])
PY
