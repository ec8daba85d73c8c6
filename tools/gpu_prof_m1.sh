#!/bin/bash
ulimit -c 0
cd "$GRAFT_REPO_ROOT" || exit 1
timeout -s KILL 330 bash tools/profile_round.sh r03 memchr1 > gpurun_out/r03_profile_round_m1.log 2>&1
tail -c 300 gpurun_out/r03_memchr1_bench.json
