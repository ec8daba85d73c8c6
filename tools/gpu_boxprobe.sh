#!/bin/bash
# is the box-to-box spread of the single-byte workload a property of the box?  the bare read+write ubench next to the kernel
ulimit -c 0
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
T=$(date +%s)
{ rocm-smi --showmemuse --showclocks --showperflevel 2>/dev/null | grep -E "mclk|sclk|fclk|Perf|Memory" | head -8
  hipcc --offload-arch=gfx950 -O3 -o /tmp/wc tools/ubench/write_ceiling.hip && timeout 120 /tmp/wc 32 | head -4
  timeout 300 python bench.py --workload memchr1 --no-extra --no-cpu-baseline --steps 10 --warmup 2 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.readline()); print('memchr1 kernel_ms', j['roofline']['kernel_ms'], 'frac', j['roofline']['frac'])"
  KREP_GPU_NO_FUSED1=1 timeout 300 python bench.py --workload memchr1 --no-extra --no-cpu-baseline --steps 10 --warmup 2 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.readline()); print('memchr1 two-pass kernel_ms', j['roofline']['kernel_ms'], 'frac', j['roofline']['frac'])"
  timeout 300 python bench.py --workload literal8 --no-extra --no-cpu-baseline --steps 10 --warmup 2 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.readline()); print('literal8 kernel_ms', j['roofline']['kernel_ms'], 'frac', j['roofline']['frac'])"
} > gpurun_out/boxprobe_$T.txt 2>&1
cat gpurun_out/boxprobe_$T.txt
