#!/bin/bash
# round 3, first GPU batch: the whole -m gpu suite (failover, chained pieces, RCCL self-tests, direct _ref checker),
# the C-level communicator through bench.py, and an A/B of the multi-pattern scan with non-temporal stream loads
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r03a_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r03a_pytest.log
tail -15 gpurun_out/r03a_pytest.log
python bench.py --force-dist --steps 5 --warmup 2 --no-extra --no-cpu-baseline > gpurun_out/r03a_forcedist.json 2> gpurun_out/r03a_forcedist.err
tail -3 gpurun_out/r03a_forcedist.err; head -c 600 gpurun_out/r03a_forcedist.json; echo
python tools/ab_bench.py 32 4 pos "$PWD/krep_amd/lib/libkrep_gpu.so" "$PWD/krep_amd/lib/exp/libkrep_gpu_acnt.so" > gpurun_out/r03a_ab_ac.txt 2>&1
cat gpurun_out/r03a_ab_ac.txt | tail -4
python tools/ab_bench.py 32 4 count "$PWD/krep_amd/lib/libkrep_gpu.so" "$PWD/krep_amd/lib/exp/libkrep_gpu_acnt.so" > gpurun_out/r03a_ab_ac_count.txt 2>&1
cat gpurun_out/r03a_ab_ac_count.txt | tail -4
