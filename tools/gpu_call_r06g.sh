#!/usr/bin/env bash
set -u
OUT=gpurun_out/r06g
mkdir -p $OUT
export PYTHONPATH=$PWD HSA_ENABLE_IPC_MODE_LEGACY=0
RR_DEBUG_POISON_ALLOC=1 timeout 1500 python -m pytest tests -m gpu -q --timeout 900 --deselect tests/test_gpu_poison.py > $OUT/pytest_poison.txt 2>&1; echo "pytest (poisoned allocations) rc=$?: $(tail -1 $OUT/pytest_poison.txt)" | tee -a $OUT/summary.txt
grep -E "^FAILED|^ERROR" $OUT/pytest_poison.txt | head -30 | cut -c1-300 | tee -a $OUT/summary.txt
grep -h "VALIDATION MISMATCH\|UNSHARDED REFERENCE" gpurun_out/test_bench_eight_ranks.stderr.txt 2>/dev/null | head -3 | cut -c1-500 | tee -a $OUT/summary.txt
