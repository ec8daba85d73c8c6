#!/bin/bash
# round 3: the one-pass single-byte kernel (kg_single.hip) — parity, then A/B against the two-pass path in one process
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r03b_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r03b_pytest.log
tail -8 gpurun_out/r03b_pytest.log
L="$PWD/krep_amd/lib/libkrep_gpu.so"
timeout 600 python tools/ab_bench.py 32 3 pos "$L" "$L:KREP_GPU_NO_FUSED1=1" > gpurun_out/r03b_ab_m1.txt 2>&1
tail -3 gpurun_out/r03b_ab_m1.txt
timeout 600 python tools/ab_bench.py 8 3 pos "$L" "$L:KREP_GPU_NO_FUSED1=1" > gpurun_out/r03b_ab_m1_8g.txt 2>&1
tail -3 gpurun_out/r03b_ab_m1_8g.txt
