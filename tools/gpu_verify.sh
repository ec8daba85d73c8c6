#!/bin/bash
# what the driver does at round end, in one call: the -m gpu suite, smoke(), the default bench line
ulimit -c 0
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -s KILL 360 python -m pytest tests -m gpu -x -q > gpurun_out/verify_pytest.log 2>&1; rc=$?; echo "pytest rc=$rc" >> gpurun_out/verify_pytest.log
tail -3 gpurun_out/verify_pytest.log | cut -c1-200
[ $rc -ne 0 ] && exit 1
timeout -s KILL 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout -s KILL 240 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/verify_bench.json 2> gpurun_out/verify_bench.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/verify_bench.json').readline())
print('literal8', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('traffic_source','')[:60])
for k,e in d['extra'].items(): print(k, e['value'], e['ms_per_step'], e['roofline']['frac'])
PY
