#!/bin/bash
# A/B of library builds in ONE GPU session (box-to-box variance is +-4 %): tools/gpu_ab.sh "<variant> ..." [kind list] [rounds]
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
VARS=${1:-"A B"}; KINDS=${2:-"2 3"}; ROUNDS=${3:-2}
for rep in $(seq 1 $ROUNDS); do
  for v in $VARS; do
    for k in $KINDS; do
      echo -n "[$v] "; KREP_GPU_LIB=$R/krep_amd/lib/variants/$v.so timeout 300 python tools/quick_bench.py 32 $k 7 2>&1 | grep "mode=pos" | sed 's/count=.*overflow=0//'
    done
  done
done
