#!/usr/bin/env bash
# flake hunt: the GPU suite twice more on one box (the second run meets the memory the first one left behind)
set -u
export PYTHONPATH=$PWD HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/r06v
mkdir -p $OUT
for i in 1 2; do
  timeout 1500 python -m pytest tests -m gpu -q --timeout 1200 -p no:cacheprovider > $OUT/pytest_gpu_$i.txt 2>&1; echo "run $i rc=$?: $(tail -1 $OUT/pytest_gpu_$i.txt)" | tee -a $OUT/summary.txt
  grep -E "^FAILED|^ERROR" $OUT/pytest_gpu_$i.txt | head -10 | cut -c1-300 | tee -a $OUT/summary.txt
done
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err; echo "bench rc=$?" | tee -a $OUT/summary.txt
python - <<'PY' | tee -a $OUT/summary.txt
import json
d=json.loads(open('gpurun_out/r06v/bench_driver.json').read().strip().splitlines()[-1])
print({k:d.get(k) for k in ('value','ms_per_step','ms_per_step_cold')}, d.get('roofline'))
PY
