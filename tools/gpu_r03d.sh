#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_literal.py tests/test_gpu_fullsize.py tests/test_gpu_multi.py tests/test_gpu_carveouts.py tests/test_golden_vectors.py -m gpu -x -q > gpurun_out/r03d_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r03d_pytest.log
tail -5 gpurun_out/r03d_pytest.log
L="$PWD/krep_amd/lib/libkrep_gpu.so"; E="$PWD/krep_amd/lib/exp"
timeout 600 python tools/ab_bench.py 32 3 pos "$L" "$E/libkrep_gpu_s1v1.so" > gpurun_out/r03d_ab_m1.txt 2>&1
tail -2 gpurun_out/r03d_ab_m1.txt
