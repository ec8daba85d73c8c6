#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
ulimit -c 0
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/r03i_pytest.log 2>&1; rc=$?; echo "pytest rc=$rc" >> gpurun_out/r03i_pytest.log
tail -4 gpurun_out/r03i_pytest.log | cut -c1-300
[ $rc -ne 0 ] && exit 1   # no evidence batch on top of a failing suite
bash tools/gpu_batch_r03.sh
