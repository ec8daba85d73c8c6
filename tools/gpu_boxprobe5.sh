#!/bin/bash
ulimit -c 0
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O; T=$(date +%s)
{ for i in 1 2 3; do for m in arena separate posfirst; do timeout 200 python tools/arena_probe.py 32 $m 2>&1 | grep -v amdgpu.ids; done; done; } > $O/boxprobe5_$T.txt
cat $O/boxprobe5_$T.txt
