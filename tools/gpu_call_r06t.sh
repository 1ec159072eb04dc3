#!/usr/bin/env bash
# final evidence of round 6: rocprofv3 kernel stats + PMC passes + un-profiled lines of the final build
set -u
export PYTHONPATH=$PWD HSA_ENABLE_IPC_MODE_LEGACY=0
bash tools/collect_profiles.sh r06t > gpurun_out/r06t_collect.log 2>&1
echo "collect rc=$?"
ls gpurun_out/r06t | grep -v "^raw_" | head -60
rm -rf gpurun_out/r06t/raw_*
