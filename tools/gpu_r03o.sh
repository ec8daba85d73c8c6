#!/bin/bash
ulimit -c 0
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_cli.py tests/test_gpu_failover.py -m gpu -x -q > gpurun_out/r03o_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r03o_pytest.log
tail -12 gpurun_out/r03o_pytest.log | cut -c1-300
