#!/bin/bash
ulimit -c 0
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
L="$PWD/krep_amd/lib/libkrep_gpu.so"; E="$PWD/krep_amd/lib/exp"
timeout 400 python tools/ab_bench.py 32 3 pos "$L" "$E/libkrep_gpu_upt2.so" "$E/libkrep_gpu_upt2n8.so" 2>&1 | tail -3 | tee gpurun_out/ab_upt_$(date +%s).txt
