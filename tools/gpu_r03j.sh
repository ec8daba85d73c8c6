#!/bin/bash
# second half of the round-3 evidence batch (after the gather fix): everything that touches the multi-pattern kernels
ulimit -c 0
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/r03j_pytest.log 2>&1; rc=$?; echo "pytest rc=$rc" >> $O/r03j_pytest.log
tail -4 $O/r03j_pytest.log | cut -c1-300
[ $rc -ne 0 ] && exit 1
timeout 600 python bench.py > $O/r03_default_bench.json 2> $O/r03_default_bench.err; tail -c 300 $O/r03_default_bench.json; echo
bash tools/profile_round.sh r03 ac1000 > $O/r03_profile_round_ac.log 2>&1
timeout 300 python tools/ac_modes_bench.py 32 > $O/r03_ac_modes.log 2>&1; tail -4 $O/r03_ac_modes.log
timeout 300 python tools/ac_small_bench.py 8 > $O/r03_ac_small.log 2>&1; tail -12 $O/r03_ac_small.log
timeout 300 python tools/ac_dense_bench.py > $O/r03_ac_dense.log 2>&1; tail -8 $O/r03_ac_dense.log
timeout 300 python tools/host_path_bench.py > $O/r03_host_path.log 2>&1; tail -5 $O/r03_host_path.log
timeout 300 python tools/host_latency_bench.py > $O/r03_host_latency.log 2>&1; tail -6 $O/r03_host_latency.log
