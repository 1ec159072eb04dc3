#!/usr/bin/env bash
set -u
OUT=gpurun_out/r06f
mkdir -p $OUT
export PYTHONPATH=$PWD HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_world8.py tests/test_gpu_p2p.py -q -m gpu --timeout 900 > $OUT/pytest_w8_p2p.txt 2>&1; echo "pytest world8+p2p rc=$?: $(tail -1 $OUT/pytest_w8_p2p.txt)" | tee -a $OUT/summary.txt
grep -h "VALIDATION MISMATCH\|UNSHARDED REFERENCE" gpurun_out/test_bench_eight_ranks.stderr.txt | head -6 | cut -c1-700 | tee -a $OUT/summary.txt
timeout 300 python bench.py --gpus 1 --force-sharded --transport p2p-only --scheme multinomial --no-extra-legs --no-cpu-baseline > $OUT/bench_mn_p2p_world1.json 2> $OUT/bench_mn_p2p_world1.err; python - <<'PY' | tee -a $OUT/summary.txt
import json
try:
    d=json.load(open('bench_legs.json')); print('mn p2p world1 ms', d.get('ms_per_step'), d.get('kernel_ms_avg'))
except Exception as e: print(e)
PY
