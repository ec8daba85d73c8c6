#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
L="$PWD/krep_amd/lib/libkrep_gpu.so"; E="$PWD/krep_amd/lib/exp"
timeout 600 python tools/ab_bench.py 32 3 pos "$L" "$E/libkrep_gpu_s1nostore.so" "$E/libkrep_gpu_s1nowait.so" "$L:KREP_GPU_NO_FUSED1=1" > gpurun_out/r03c_ab_m1.txt 2>&1
tail -4 gpurun_out/r03c_ab_m1.txt
timeout 600 python tools/ab_bench.py 32 3 count "$L" > gpurun_out/r03c_ab_m1_count.txt 2>&1
tail -1 gpurun_out/r03c_ab_m1_count.txt
