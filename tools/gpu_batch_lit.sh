#!/bin/bash
# literal-kernel iteration batch: parity of the literal tests + the length sweep
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_literal.py tests/test_gpu_carveouts.py tests/test_golden_vectors.py -m gpu -q -x --timeout 300 --timeout-method thread -p no:cacheprovider 2>&1 | tail -4
timeout 600 python tools/literal_sweep.py 8 ${1:-8,9,16,17,24,32,33,48,64,65,128} 2>&1 | tee $O/sweep_${2:-x}.log
