#!/bin/bash
# round-3 evidence batch (on the GPU box): default bench line, 1-rank RCCL self-test through the C communicator, profile set
# (kernel stats with warm-up launches dropped + FETCH/WRITE passes per workload, the filter-only FETCH pass of ac1000),
# TLB / write-stall counters of the literal scan with and without offsets, memchr1 run-to-run over six processes,
# read/write ceilings, dictionaries, host path.
set -u
ulimit -c 0   # a faulting kernel must not fill the box's disk with core dumps (it cost the first run of this batch 37 minutes)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 600 python bench.py > $O/r03_default_bench.json 2> $O/r03_default_bench.err; tail -c 300 $O/r03_default_bench.json; echo
timeout 300 python bench.py --force-dist --no-extra --no-cpu-baseline --steps 5 > $O/r03_force_dist_1rank_rccl.json 2> $O/r03_force_dist_1rank_rccl.err; tail -1 $O/r03_force_dist_1rank_rccl.err
bash tools/profile_round.sh r03 > $O/r03_profile_round.log 2>&1
for i in 1 2 3 4 5 6; do
  timeout 300 python bench.py --workload memchr1 --no-extra --no-cpu-baseline --steps 10 --warmup 2 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.readline()); print('memchr1 process $i: kernel_ms', j['roofline']['kernel_ms'], 'frac', j['roofline']['frac'])"
done > $O/r03_memchr1_run_to_run.txt 2>&1; cat $O/r03_memchr1_run_to_run.txt
for i in 1 2 3 4; do
  timeout 300 python bench.py --workload literal8 --no-extra --no-cpu-baseline --steps 10 --warmup 2 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.readline()); print('literal8 process $i: kernel_ms', j['roofline']['kernel_ms'], 'frac', j['roofline']['frac'])"
done > $O/r03_literal8_run_to_run.txt 2>&1; cat $O/r03_literal8_run_to_run.txt
TLB="TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCC_EA0_WRREQ_STALL_sum"
bash tools/pmc_pass.sh r03_lit_tlb "$TLB" --workload literal8 > $O/r03_lit_tlb.log 2>&1
bash tools/pmc_pass.sh r03_lit_tlb2 "TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_DRAM_sum TCC_TAG_STALL_sum" --workload literal8 > $O/r03_lit_tlb2.log 2>&1
bash tools/pmc_pass.sh r03_m1_tlb "$TLB" --workload memchr1 > $O/r03_m1_tlb.log 2>&1
grep -h "lit_scan\|single_fused" $O/r03_lit_tlb/summary.txt $O/r03_lit_tlb2/summary.txt $O/r03_m1_tlb/summary.txt | cut -c1-150
cd /tmp && hipcc --offload-arch=gfx950 -O3 -o /tmp/wc $R/tools/ubench/write_ceiling.hip && /tmp/wc 32 > $O/r03_write_ceiling.log 2>&1; cd $R; head -4 $O/r03_write_ceiling.log
timeout 300 python tools/ac_modes_bench.py 32 > $O/r03_ac_modes.log 2>&1; tail -4 $O/r03_ac_modes.log
timeout 300 python tools/ac_small_bench.py 8 > $O/r03_ac_small.log 2>&1; tail -12 $O/r03_ac_small.log
timeout 300 python tools/ac_dense_bench.py > $O/r03_ac_dense.log 2>&1; tail -8 $O/r03_ac_dense.log
timeout 300 python tools/host_path_bench.py > $O/r03_host_path.log 2>&1; tail -5 $O/r03_host_path.log
timeout 300 python tools/host_latency_bench.py > $O/r03_host_latency.log 2>&1; tail -6 $O/r03_host_latency.log
