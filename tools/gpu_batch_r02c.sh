#!/bin/bash
# round-2 final evidence batch (on the GPU box): full GPU test-suite, default bench line, 1-rank RCCL self-test, profile set
# (kernel stats + FETCH/WRITE passes per workload), SQ counters (literal + multi-pattern), length sweep, host-path numbers.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 900 python -m pytest tests -m gpu -q --timeout 400 --timeout-method thread -p no:cacheprovider > $O/pytest_r2c.log 2>&1; tail -3 $O/pytest_r2c.log
timeout 600 python bench.py > $O/r02_default_bench.json 2> $O/r02_default_bench.err; tail -c 300 $O/r02_default_bench.json
timeout 300 python bench.py --force-dist --no-extra --no-cpu-baseline --steps 5 > $O/r02_force_dist_1rank.json 2> $O/r02_force_dist_1rank.err
bash tools/profile_round.sh r02 > $O/r02_profile_round.log 2>&1
SQ="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
bash tools/pmc_pass.sh r02_ac_sq "$SQ" --workload ac1000 --gib 8 > $O/r02_ac_sq.log 2>&1
bash tools/pmc_pass.sh r02_ac_sq2 "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVES" --workload ac1000 --gib 8 > $O/r02_ac_sq2.log 2>&1
KREP_GPU_AC_NOVERIFY=1 bash tools/pmc_pass.sh r02_ac_noverify_fetch "FETCH_SIZE" --workload ac1000 --gib 8 > $O/r02_ac_noverify_fetch.log 2>&1
bash tools/pmc_pass.sh r02_ac_full_fetch "FETCH_SIZE" --workload ac1000 --gib 8 > $O/r02_ac_full_fetch.log 2>&1
bash tools/pmc_pass.sh r02_lit_sq1 "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY" --workload literal8 > $O/r02_lit_sq1.log 2>&1
bash tools/pmc_pass.sh r02_lit_sq2 "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_WAVES SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_BRANCH" --workload literal8 > $O/r02_lit_sq2.log 2>&1
timeout 600 python tools/literal_sweep.py 8 > $O/r02_literal_sweep.log 2>&1; tail -20 $O/r02_literal_sweep.log
timeout 300 python tools/host_path_bench.py > $O/r02_host_path.log 2>&1; tail -5 $O/r02_host_path.log
timeout 300 python tools/host_latency_bench.py > $O/r02_host_latency.log 2>&1; tail -6 $O/r02_host_latency.log
cd /tmp && hipcc --offload-arch=gfx950 -O3 -o /tmp/rc $R/tools/ubench/read_ceiling.hip && /tmp/rc 32 > $O/r02_read_ceiling.log 2>&1; grep -v "lds-dma 16" $O/r02_read_ceiling.log | head -12
