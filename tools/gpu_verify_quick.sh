#!/bin/bash
# hard wall-clock limits around everything: a hung kernel must cost seconds, not the round's GPU budget
ulimit -c 0
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -s KILL 200 python -m pytest tests/test_golden_vectors.py tests/test_gpu_ac.py tests/test_gpu_literal.py tests/test_gpu_greedy.py tests/test_gpu_format.py -m gpu -x -q > gpurun_out/vq_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/vq_pytest.log
tail -3 gpurun_out/vq_pytest.log | cut -c1-200
timeout -s KILL 100 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
