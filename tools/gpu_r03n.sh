#!/bin/bash
ulimit -c 0
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 500 python tools/soak_literal.py 60 > gpurun_out/r03n_soak_literal.txt 2>&1; tail -3 gpurun_out/r03n_soak_literal.txt
timeout 300 python tools/soak_ac.py 30 > gpurun_out/r03n_soak_ac.txt 2>&1; tail -2 gpurun_out/r03n_soak_ac.txt
