#!/bin/bash
ulimit -c 0
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_ac.py tests/test_golden_vectors.py -m gpu -x -q > gpurun_out/r03m_pytest.log 2>&1; rc=$?; echo "pytest rc=$rc" >> gpurun_out/r03m_pytest.log
tail -3 gpurun_out/r03m_pytest.log | cut -c1-200
[ $rc -ne 0 ] && exit 1
L="$PWD/krep_amd/lib/libkrep_gpu.so"; E="$PWD/krep_amd/lib/exp"
timeout 300 python tools/ab_bench.py 32 4 pos "$L" "$E/libkrep_gpu_base.so" 2>&1 | tail -2
timeout 300 python tools/ab_bench.py 32 4 count "$L" "$E/libkrep_gpu_base.so" 2>&1 | tail -2
