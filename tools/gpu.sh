#!/bin/bash
# The ONE runner for work on the GPU box (through gpurun): every step under its own hard wall-clock limit — a hung kernel must
# cost seconds, not the round's GPU budget — with its output kept under gpurun_out/ and the tail echoed.
#   usage: tools/gpu.sh <label> '<seconds>::<shell command>' ['<seconds>::<shell command>' ...]
#   e.g.   gpurun --timeout 900 -- tools/gpu.sh r04a '300::python -m pytest tests/test_gpu_ac.py -m gpu -x -q' \
#                                                     '200::python tools/ab_bench.py 32 4 pos /root/repo/krep_amd/lib/libkrep_gpu.so'
# Step i writes gpurun_out/<label>_<i>.log; <label>_steps.txt lists step, exit code and seconds.  Profiling recipes that need
# more than a command line: tools/profile_round.sh (kernel stats + FETCH/WRITE passes), tools/pmc_pass.sh (one counter pass).
set -u
ulimit -c 0 # a faulting kernel must not fill the box's disk with core dumps
LABEL=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p "$O"; cd "$R" || exit 1
export TMPDIR=/tmp
: > "$O/${LABEL}_steps.txt"
i=0
for step in "$@"; do
  i=$((i + 1))
  secs=${step%%::*}; cmd=${step#*::}
  log="$O/${LABEL}_${i}.log"
  t0=$(date +%s)
  timeout -s KILL "$secs" bash -c "$cmd" > "$log" 2>&1
  rc=$?
  t1=$(date +%s)
  echo "step $i rc=$rc $((t1 - t0))s :: $cmd" | tee -a "$O/${LABEL}_steps.txt" | cut -c1-220
  tail -n "${GPU_SH_TAIL:-12}" "$log" | cut -c1-260
done
