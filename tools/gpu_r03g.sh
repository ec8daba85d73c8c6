#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r03g_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r03g_pytest.log
tail -3 gpurun_out/r03g_pytest.log
timeout 900 python tools/literal_sweep.py 32 2,3,4,5,8,9,12,16,17,24,32,48,64,128 > gpurun_out/r03g_literal_sweep_32gib.txt 2>&1
cat gpurun_out/r03g_literal_sweep_32gib.txt
