#!/bin/bash
ulimit -c 0
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
hipcc --offload-arch=gfx950 -O3 -o /tmp/wc tools/ubench/write_ceiling.hip && timeout 120 /tmp/wc 32 > gpurun_out/r03k_write_ceiling.txt 2>&1
cat gpurun_out/r03k_write_ceiling.txt
