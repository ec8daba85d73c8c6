#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd /tmp
hipcc --offload-arch=gfx950 -O3 -o /tmp/rc $R/tools/ubench/read_ceiling.hip && /tmp/rc 32 | tee $O/read_ceiling_r02b.log
cd $R
bash tools/pmc_pass.sh r02_lit_sq1 "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY" --workload literal8 > $O/r02_lit_sq1.log 2>&1
bash tools/pmc_pass.sh r02_lit_sq2 "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_WAVES SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_BRANCH" --workload literal8 > $O/r02_lit_sq2.log 2>&1
grep lit_scan $O/r02_lit_sq1/summary.txt $O/r02_lit_sq2/summary.txt
