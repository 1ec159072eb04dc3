#!/usr/bin/env bash
set -u
OUT=gpurun_out/r06s
mkdir -p $OUT
export PYTHONPATH=$PWD HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_p2p.py tests/test_gpu_world8.py tests/test_gpu_sharded.py -q -m gpu -k "multinomial" --timeout 600 > $OUT/pytest_mn.txt 2>&1; echo "pytest mn rc=$?: $(tail -1 $OUT/pytest_mn.txt)" | tee -a $OUT/summary.txt
RR_MN_GUIDE=0 timeout 900 python -m pytest tests/test_gpu_p2p.py tests/test_gpu_world8.py -q -m gpu -k "multinomial" --timeout 600 > $OUT/pytest_mn_noguide.txt 2>&1; echo "pytest mn (RR_MN_GUIDE=0) rc=$?: $(tail -1 $OUT/pytest_mn_noguide.txt)" | tee -a $OUT/summary.txt
RR_DEBUG_POISON_ALLOC=0x3f timeout 900 python -m pytest tests/test_gpu_p2p.py tests/test_gpu_world8.py -q -m gpu -k "multinomial" --timeout 600 > $OUT/pytest_mn_poison.txt 2>&1; echo "pytest mn (poison 0x3f) rc=$?: $(tail -1 $OUT/pytest_mn_poison.txt)" | tee -a $OUT/summary.txt
for i in 1 2; do
timeout 300 python bench.py --gpus 1 --force-sharded --transport p2p-only --scheme multinomial --no-extra-legs --no-cpu-baseline > $OUT/mn_$i.json 2> $OUT/mn_$i.err
python - <<'PY' | tee -a $OUT/summary.txt
import json
d=json.load(open('bench_legs.json')); k=d.get('kernel_ms_avg') or {}
print('ms/step',round(d.get('ms_per_step'),5), {a:round(b,4) for a,b in k.items()})
PY
done
