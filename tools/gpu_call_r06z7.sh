#!/usr/bin/env bash
set -u
export PYTHONPATH=$PWD HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/r06z7
mkdir -p $OUT
timeout 600 python -m pytest "$@" -m gpu -q -x --durations=3 > $OUT/pytest.txt 2>&1; echo "pytest rc=$?: $(tail -1 $OUT/pytest.txt)" | tee -a $OUT/summary.txt
grep -E "^FAILED|^ERROR|Error|^E  " $OUT/pytest.txt | head -20 | cut -c1-300 | tee -a $OUT/summary.txt
