#!/usr/bin/env bash
# round 6, late: the whole GPU suite + smoke + the driver's bench command once more on a fresh box (flake watch for the round-end run)
set -u
export PYTHONPATH=$PWD HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/${1:-r06z2}
mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q --timeout 1200 --durations=5 > $OUT/pytest_gpu.txt 2>&1; echo "pytest gpu rc=$?: $(tail -1 $OUT/pytest_gpu.txt)" | tee -a $OUT/summary.txt
grep -E "^FAILED|^ERROR" $OUT/pytest_gpu.txt | head -10 | cut -c1-300 | tee -a $OUT/summary.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee -a $OUT/summary.txt
python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_command.json 2> $OUT/bench_driver.err; echo "bench rc=$?" | tee -a $OUT/summary.txt
tail -n 1 $OUT/bench_driver_command.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('headline', d['ms_per_step'], 'cold', d.get('ms_per_step_cold'), 'kernel_ms', r['avg_kernel_ms'], 'frac', r['frac'], 'binding', r.get('binding_frac'), 'traffic', r['traffic'])" | tee -a $OUT/summary.txt
