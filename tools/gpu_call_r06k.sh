#!/usr/bin/env bash
set -u
OUT=gpurun_out/r06k
mkdir -p $OUT
export PYTHONPATH=$PWD HSA_ENABLE_IPC_MODE_LEGACY=0
export RR_BENCH_VALIDATE_REPEAT=8
for i in 1 2 3; do
  timeout 900 python -m pytest tests/test_gpu_world8.py -q -m gpu --timeout 900 > $OUT/pytest_w8_$i.txt 2>&1; rc=$?
  echo "session $i rc=$rc: $(tail -1 $OUT/pytest_w8_$i.txt)" | tee -a $OUT/summary.txt
  cp gpurun_out/test_bench_eight_ranks.stderr.txt $OUT/bench8_stderr_$i.txt 2>/dev/null
  grep -h "VALIDATE_REPEAT\|REFERENCE SPLIT" $OUT/bench8_stderr_$i.txt | cut -c1-1200 | tee -a $OUT/summary.txt
  if grep -q "REFERENCE SPLIT" $OUT/bench8_stderr_$i.txt; then break; fi
done
