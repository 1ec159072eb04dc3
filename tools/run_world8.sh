#!/usr/bin/env bash
# n_ranks = 8 executed on ONE device, every wiring, logs under gpurun_out/<tag>/ (copy the ones to keep into profiles/):
#   bash tools/run_world8.sh r06_world8
# (a) eight shards linked in one process, whole-device rules and one eighth of the CUs each (RR_P2P_CU_PARTITION=1)
# (b) eight processes over hipIpc handles: the workers of the tests, and bench.py --gpus 8 with every rank on device 0
# Every case is bounded by `timeout`; every in-kernel wait is bounded by itself (RR_P2P_TIMEOUT_MS).
set -u
TAG=${1:-r06_world8}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export PYTHONPATH=$PWD HSA_ENABLE_IPC_MODE_LEGACY=0
{ echo "host: $(nproc) cpus"; free -g | head -2; /opt/rocm/bin/rocm-smi --showmeminfo vram 2>/dev/null | grep -i total | head -2; } > "$OUT/box.txt" 2>&1
run() {  # name, timeout, env..., -- command
  local name=$1 t=$2; shift 2
  local envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done
  shift
  local t0=$(date +%s)
  env "${envs[@]}" timeout "$t" "$@" > "$OUT/$name.out" 2> "$OUT/$name.err"
  local rc=$?
  echo "$name rc=$rc $(( $(date +%s) - t0 ))s" | tee -a "$OUT/summary.txt"
}
W=tools/world8_one_device.py
run inproc_small_whole_device 600 GPU_MAX_HW_QUEUES=12 RR_P2P_CU_PARTITION=0 -- python $W mcl-small mcl-heavy mcl-lazy-max
run inproc_small_cu_partition 600 GPU_MAX_HW_QUEUES=12 RR_P2P_CU_PARTITION=1 -- python $W mcl-small mcl-heavy mcl-lazy-max
run inproc_small_multi_launch_plan 600 GPU_MAX_HW_QUEUES=12 RR_PF_FUSED_PLAN=0 -- python $W mcl-small mcl-heavy
run inproc_config5_cu_partition 900 GPU_MAX_HW_QUEUES=12 RR_P2P_CU_PARTITION=1 -- python $W mcl-config5 mcl-config5-heavy
run inproc_config5_whole_device 900 GPU_MAX_HW_QUEUES=12 RR_P2P_CU_PARTITION=0 -- python $W mcl-config5 mcl-config5-heavy
run inproc_fs1 1200 GPU_MAX_HW_QUEUES=12 RR_P2P_CU_PARTITION=1 RR_P2P_TIMEOUT_MS=20000 -- python $W fs1-small fs1-config4
L="python -m torch.distributed.run --nnodes=1 --nproc-per-node=8 --master-addr 127.0.0.1"
run ipc_mcl_8000 600 OMP_NUM_THREADS=1 -- $L --master-port 29751 tests/_gpu_p2p_worker.py 8000 8
run ipc_mcl_8000_wmax_early 600 OMP_NUM_THREADS=1 RR_P2P_WMAX_EARLY=1 -- $L --master-port 29752 tests/_gpu_p2p_worker.py 8000 8
run ipc_mcl_config5_cu_partition 1200 OMP_NUM_THREADS=1 RR_P2P_CU_PARTITION=1 RR_WORKER_PEAKED=64 -- $L --master-port 29753 tests/_gpu_p2p_worker.py 2000000 5
run ipc_mcl_config5_whole_device 1200 OMP_NUM_THREADS=1 RR_WORKER_PEAKED=64 -- $L --master-port 29754 tests/_gpu_p2p_worker.py 2000000 5
run ipc_fs1_small 600 OMP_NUM_THREADS=1 -- $L --master-port 29755 tests/_gpu_fs1_p2p_worker.py 3000 8 7
run ipc_fs1_config4 1500 OMP_NUM_THREADS=1 -- $L --master-port 29756 tests/_gpu_fs1_p2p_worker.py 125000 4 200
run bench_gpus8_shared_device 900 RR_BENCH_SHARE_DEVICE=1 RR_BENCH_DEADLINE_S=600 RR_P2P_CU_PARTITION=1 RR_BENCH_LEGS_FILE=$OUT/bench_gpus8_shared_device_legs.json -- python bench.py --gpus 8 --steps 20 --warmup 5 --no-extra-legs --no-cpu-baseline
cat "$OUT/summary.txt"
