#!/usr/bin/env bash
# final evidence of round 6 (final sources): the GPU suite, smoke, rocprofv3 kernel stats + PMC passes + un-profiled lines
set -u
export PYTHONPATH=$PWD HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/r06zzz
mkdir -p $OUT
timeout 2400 python -m pytest tests -m gpu -q --timeout 1500 > $OUT/pytest_gpu.txt 2>&1; echo "pytest gpu rc=$?: $(tail -1 $OUT/pytest_gpu.txt)" | tee -a $OUT/summary.txt
grep -E "^FAILED|^ERROR" $OUT/pytest_gpu.txt | head -10 | cut -c1-300 | tee -a $OUT/summary.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee -a $OUT/summary.txt
bash tools/collect_profiles.sh r06zzz > $OUT/collect.log 2>&1
echo "collect rc=$?" | tee -a $OUT/summary.txt
rm -rf gpurun_out/r06zzz/raw_*
