#!/bin/bash
# multi-pattern kernel: filter-only time (KREP_GPU_AC_NOVERIFY) and two SQ counter passes at 8 GiB
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
V=${1:-base.so pairmix.so}
echo "--- filter only (no verify)"; KREP_GPU_AC_NOVERIFY=1 AB_REPS=5 python tools/ab_bench.py 32 4 count $V 2>&1 | grep median | cut -c1-100
tools/pmc_pass.sh ac_sq1 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS" --workload ac1000 --gib 8 | grep ac_scan
tools/pmc_pass.sh ac_sq2 "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVES" --workload ac1000 --gib 8 | grep ac_scan
