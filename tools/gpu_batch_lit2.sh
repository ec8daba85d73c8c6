#!/bin/bash
# literal-kernel iteration: parity + quick 32 GiB timings for literal8 / memchr1 at several blocks-per-CU
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_literal.py tests/test_gpu_carveouts.py tests/test_golden_vectors.py tests/test_gpu_greedy.py tests/test_gpu_multi.py -m gpu -q -x --timeout 300 --timeout-method thread -p no:cacheprovider 2>&1 | tail -4
for b in ${BPCS:-4 2 8}; do
  echo "== blocks/CU $b"; KREP_GPU_LIT_BLOCKS_PER_CU=$b timeout 300 python tools/quick_bench.py 32 2 7 2>&1 | grep kind
  KREP_GPU_LIT_BLOCKS_PER_CU=$b timeout 300 python tools/quick_bench.py 32 3 5 2>&1 | grep kind
done
