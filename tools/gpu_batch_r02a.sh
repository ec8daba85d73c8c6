#!/bin/bash
# round-2 batch A (on the GPU box): full GPU test-suite, the default bench line (N=1 with the extra workloads), the
# 1-rank RCCL self-test, the per-workload profile set (kernel stats + FETCH/WRITE passes) and the SQ counter passes of the
# multi-pattern kernel (full kernel and filter-only).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -q --timeout 400 --timeout-method thread -p no:cacheprovider > $O/pytest_r2b.log 2>&1
tail -5 $O/pytest_r2b.log
timeout 600 python bench.py > $O/r02_default_bench.json 2> $O/r02_default_bench.err
tail -c 600 $O/r02_default_bench.json
timeout 300 python bench.py --force-dist --no-extra --no-cpu-baseline --steps 5 > $O/r02_force_dist_1rank.json 2> $O/r02_force_dist_1rank.err
tail -c 300 $O/r02_force_dist_1rank.json
bash tools/profile_round.sh r02 > $O/r02_profile_round.log 2>&1
tail -5 $O/r02_profile_round.log
SQ="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
bash tools/pmc_pass.sh r02_ac_sq "$SQ" --workload ac1000 --gib 8 > $O/r02_ac_sq.log 2>&1
bash tools/pmc_pass.sh r02_ac_sq2 "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVES" --workload ac1000 --gib 8 > $O/r02_ac_sq2.log 2>&1
KREP_GPU_AC_NOVERIFY=1 bash tools/pmc_pass.sh r02_ac_noverify_fetch "FETCH_SIZE" --workload ac1000 --gib 8 > $O/r02_ac_noverify_fetch.log 2>&1
bash tools/pmc_pass.sh r02_ac_full_fetch "FETCH_SIZE" --workload ac1000 --gib 8 > $O/r02_ac_full_fetch.log 2>&1
tail -12 $O/r02_ac_sq.log $O/r02_ac_sq2.log $O/r02_ac_noverify_fetch.log $O/r02_ac_full_fetch.log
