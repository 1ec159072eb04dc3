#!/usr/bin/env bash
# second GPU call of round 6
set -u
OUT=gpurun_out/r06b
mkdir -p $OUT
export PYTHONPATH=$PWD HSA_ENABLE_IPC_MODE_LEGACY=0
date > $OUT/start.txt
timeout 1500 python -m pytest tests -m gpu -q -x --timeout 900 > $OUT/pytest_gpu.txt 2>&1; echo "pytest rc=$?" | tee -a $OUT/summary.txt; tail -5 $OUT/pytest_gpu.txt | tee -a $OUT/summary.txt
timeout 600 bash tools/ab_bench.sh mcl r05 default > $OUT/ab_mcl.txt 2>&1; cat $OUT/ab_mcl.txt | tee -a $OUT/summary.txt
timeout 400 bash tools/ab_bench.sh fastslam r05 default > $OUT/ab_fs1.txt 2>&1; cat $OUT/ab_fs1.txt | tee -a $OUT/summary.txt
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extra-legs > $OUT/bench_driver_noextra.json 2> $OUT/bench_driver_noextra.err; echo "bench rc=$?" | tee -a $OUT/summary.txt
tail -1 $OUT/bench_driver_noextra.json | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print({k:d.get(k) for k in ('ms_per_step','ms_per_step_cold','ms_per_step_cold_unwarmed')}, d['roofline'])" | tee -a $OUT/summary.txt
grep -h synchronous_try_step bench_legs.json | head -2
python - <<'PY' | tee -a $OUT/summary.txt
import json
d=json.load(open('bench_legs.json'))
print('sync', d.get('synchronous_try_step'))
print('plain', d.get('plain_async_step'))
print('kernels', d.get('kernel_ms_avg'))
PY
L="python -m torch.distributed.run --nnodes=1 --nproc-per-node=8 --master-addr 127.0.0.1"
OMP_NUM_THREADS=1 RR_WORKER_DUMP_AFTER_S=300 RR_WORKER_GLOO_TIMEOUT_S=330 timeout 400 $L --master-port 29766 tests/_gpu_fs1_p2p_worker.py 125000 4 200 > $OUT/ipc_fs1_config4.out 2> $OUT/ipc_fs1_config4.err; echo "ipc_fs1_config4 rc=$?" | tee -a $OUT/summary.txt
grep -E "fs1 p2p worker|Traceback|File |Error" $OUT/ipc_fs1_config4.err | tail -40 | tee -a $OUT/summary.txt
timeout 1500 bash tools/fs1_tlb_probe.sh r06_fs1_tlb > $OUT/tlb_probe.txt 2>&1; tail -15 $OUT/tlb_probe.txt | tee -a $OUT/summary.txt
date >> $OUT/start.txt
