#!/bin/bash
ulimit -c 0
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python tools/literal_sweep.py 32 2,3,4,5,8,9,12,16,17,24,32,48,64,128 2>&1 | grep -v amdgpu.ids > gpurun_out/r03_literal_sweep_32gib_final.txt
cat gpurun_out/r03_literal_sweep_32gib_final.txt
