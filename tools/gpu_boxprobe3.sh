#!/bin/bash
# same box, same kernel: PyTorch's bundled HIP runtime (bench.py) vs the system HIP runtime (tools/notorch_bench.py)
ulimit -c 0
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O; T=$(date +%s)
{ for w in memchr1 literal8; do
    timeout 300 python bench.py --workload $w --no-extra --no-cpu-baseline --steps 10 --warmup 2 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.readline()); print('torch runtime  $w kernel_ms', j['roofline']['kernel_ms'], 'frac', j['roofline']['frac'])"
    timeout 300 python tools/notorch_bench.py 32 $w 2>&1 | grep -v amdgpu.ids | tail -2
  done
  timeout 300 python bench.py --workload memchr1 --no-extra --no-cpu-baseline --steps 10 --warmup 2 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.readline()); print('torch runtime  memchr1 again kernel_ms', j['roofline']['kernel_ms'])"
} > $O/boxprobe3_$T.txt 2>&1
cat $O/boxprobe3_$T.txt
