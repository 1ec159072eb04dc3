#!/usr/bin/env bash
# fourth GPU call of round 6: the GPU suite on the committed build, the world-8 rig again, FastSLAM configs[3] as eight processes
set -u
OUT=gpurun_out/r06d
mkdir -p $OUT
export PYTHONPATH=$PWD HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1800 python -m pytest tests -m gpu -q --timeout 900 > $OUT/pytest_gpu.txt 2>&1; echo "pytest rc=$?" | tee -a $OUT/summary.txt; tail -12 $OUT/pytest_gpu.txt | cut -c1-1500 | tee -a $OUT/summary.txt
L="python -m torch.distributed.run --nnodes=1 --nproc-per-node=8 --master-addr 127.0.0.1"
OMP_NUM_THREADS=1 RR_P2P_TIMEOUT_MS=30000 RR_WORKER_DUMP_AFTER_S=300 RR_WORKER_GLOO_TIMEOUT_S=330 timeout 420 $L --master-port 29766 tests/_gpu_fs1_p2p_worker.py 125000 4 200 > $OUT/ipc_fs1_config4.out 2> $OUT/ipc_fs1_config4.err; echo "ipc_fs1_config4 rc=$? ok=$(grep -o FS1_P2P_OK $OUT/ipc_fs1_config4.out | wc -l)" | tee -a $OUT/summary.txt
grep -h "FS1_P2P_CONFIG4" $OUT/ipc_fs1_config4.out | tee -a $OUT/summary.txt
grep -E "fs1 p2p worker rank 0" $OUT/ipc_fs1_config4.err | tail -12 | tee -a $OUT/summary.txt
timeout 300 python bench.py --gpus 1 --force-sharded --transport p2p-only --scheme multinomial --no-extra-legs --no-cpu-baseline > $OUT/bench_mn_p2p_world1.json 2> $OUT/bench_mn_p2p_world1.err; tail -1 $OUT/bench_mn_p2p_world1.json | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('multinomial p2p world1 ms_per_step', d['ms_per_step'], d.get('kernel_ms_avg'))" | tee -a $OUT/summary.txt
python - <<'PY' | tee -a $OUT/summary.txt
import json
try:
    d=json.load(open('bench_legs.json')); print('mn p2p kernels', d.get('kernel_ms_avg'))
except Exception as e: print(e)
PY
