#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
hipcc --offload-arch=gfx950 -O3 -o /tmp/wc tools/ubench/write_ceiling.hip && /tmp/wc 32 > gpurun_out/r03e_write_ceiling.txt 2>&1
cat gpurun_out/r03e_write_ceiling.txt
L="$PWD/krep_amd/lib/libkrep_gpu.so"
timeout 600 python tools/ab_bench.py 32 3 pos "$L" "$L:KREP_GPU_NO_FUSED1=1" > gpurun_out/r03e_ab_m1.txt 2>&1
tail -2 gpurun_out/r03e_ab_m1.txt
