#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): rocprofv3 kernel statistics + separate FETCH_SIZE / WRITE_SIZE passes of the
# bench command for each workload; raw output under gpurun_out/<tag>_*, summarised into profiles/ by
# tools/collect_profiles.py.   usage: tools/profile_round.sh r01 [workload ...]
set -u
TAG=${1:-r01}; shift || true
WL=${@:-literal8 memchr1 ac1000}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for w in $WL; do
  timeout 600 python $R/bench.py --workload $w --no-extra > $O/${TAG}_${w}_bench.json 2> $O/${TAG}_${w}_bench.err
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_${w}_kt -o kt -- \
      python $R/bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline --no-extra > $O/${TAG}_${w}_kt.log 2>&1
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 600 rocprofv3 --pmc $c --output-format csv -d $O/${TAG}_${w}_$c -o pmc -- \
        python $R/bench.py --workload $w --steps 2 --warmup 1 --no-cpu-baseline --no-extra > $O/${TAG}_${w}_$c.log 2>&1
  done
  if [ "$w" = "ac1000" ] || [ "$w" = "words1000" ]; then
    # the filter alone (ablation hook): its streamed reads are the part of FETCH_SIZE that has to be doubled; the verify
    # stage's gathers are counted exactly (profiles/r03_fetch_size_calibration.txt)
    KREP_GPU_AC_NOVERIFY=1 timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/${TAG}_${w}_FETCH_SIZE_filter -o pmc -- \
        python $R/bench.py --workload $w --steps 2 --warmup 1 --no-cpu-baseline --no-extra > $O/${TAG}_${w}_FETCH_SIZE_filter.log 2>&1
  fi
  tail -c 400 $O/${TAG}_${w}_bench.json | cut -c1-400
done
