#!/bin/bash
ulimit -c 0
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O; T=$(date +%s)
timeout 400 python tools/placement_probe_m1.py 32 6 2>&1 | grep -v amdgpu.ids > $O/boxprobe4_$T.txt
cat $O/boxprobe4_$T.txt
