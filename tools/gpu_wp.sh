#!/bin/bash
ulimit -c 0
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
hipcc --offload-arch=gfx950 -O3 -o /tmp/wp tools/ubench/window_probe.hip && timeout 200 /tmp/wp > gpurun_out/r03_window_probe.txt 2>&1
cat gpurun_out/r03_window_probe.txt
