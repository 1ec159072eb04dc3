#!/usr/bin/env bash
set -u
OUT=gpurun_out/r06p
mkdir -p $OUT
export PYTHONPATH=$PWD HSA_ENABLE_IPC_MODE_LEGACY=0
echo "probe: $(tools/ubench/memset_sync_probe 20 256)" | tee -a $OUT/summary.txt
timeout 2400 python -m pytest tests -m gpu -q --timeout 1500 -x > $OUT/pytest_gpu.txt 2>&1; echo "pytest gpu rc=$?: $(tail -1 $OUT/pytest_gpu.txt)" | tee -a $OUT/summary.txt
grep -E "^FAILED|^ERROR" $OUT/pytest_gpu.txt | head -10 | cut -c1-300 | tee -a $OUT/summary.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1 | tee -a $OUT/summary.txt
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err; echo "bench rc=$?" | tee -a $OUT/summary.txt
python - <<'PY' | tee -a $OUT/summary.txt
import json
d=json.loads(open('gpurun_out/r06p/bench_driver.json').read().strip().splitlines()[-1])
print({k:d.get(k) for k in ('value','ms_per_step','ms_per_step_cold','roofline','vs_baseline')})
print('sync try_step', d.get('synchronous_try_step'))
print('kernels', d.get('kernel_ms_avg'))
for leg in ('fastslam','fastslam_config4_full','mcl_config5_full','mcl_multinomial','sharded_world1','sharded_world1_multinomial','weak_scaling_ceiling','strong_scaling_ceiling'):
    v=d.get(leg)
    if isinstance(v,dict): print(leg, {k:v.get(k) for k in ('ms_per_step','p2p','rccl','roofline','configs[1] p2p','configs[1] rccl','configs[3] p2p','configs[4] p2p') if k in v})
PY
