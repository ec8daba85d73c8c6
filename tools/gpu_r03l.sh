#!/bin/bash
ulimit -c 0
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_chain.py tests/test_gpu_multi.py tests/test_gpu_carveouts.py -m gpu -x -q > gpurun_out/r03l_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r03l_pytest.log
tail -25 gpurun_out/r03l_pytest.log | cut -c1-250
