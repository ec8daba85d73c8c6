#!/bin/bash
# on a box where the single-byte workload is slow: what do the translation / write-path counters say there?
ulimit -c 0
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
T=$(date +%s)
MS=$(timeout 300 python bench.py --workload memchr1 --no-extra --no-cpu-baseline --steps 10 --warmup 2 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.readline()); print(j['roofline']['kernel_ms'])")
echo "memchr1 kernel_ms $MS" | tee $O/boxprobe2_$T.txt
SLOW=$(python -c "print(1 if float('$MS') > 7.0 else 0)")
bash tools/pmc_pass.sh boxprobe2_${T}_tlb "TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum" --workload memchr1 > /dev/null 2>&1
grep single_fused $O/boxprobe2_${T}_tlb/summary.txt | cut -c1-140 | tee -a $O/boxprobe2_$T.txt
if [ "$SLOW" = "1" ]; then
  bash tools/pmc_pass.sh boxprobe2_${T}_tlb2 "TCC_TAG_STALL_sum TCC_EA0_WRREQ_DRAM_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum GRBM_UTCL2_BUSY" --workload memchr1 > /dev/null 2>&1
  grep single_fused $O/boxprobe2_${T}_tlb2/summary.txt | cut -c1-140 | tee -a $O/boxprobe2_$T.txt
  cd /tmp && hipcc --offload-arch=gfx950 -O3 -o /tmp/wc $R/tools/ubench/write_ceiling.hip && timeout 120 /tmp/wc 32 | head -3 | tee -a $O/boxprobe2_$T.txt
  cd $R
  timeout 300 python bench.py --workload memchr1 --no-extra --no-cpu-baseline --steps 10 --warmup 2 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.readline()); print('memchr1 again kernel_ms', j['roofline']['kernel_ms'])" | tee -a $O/boxprobe2_$T.txt
fi
