#!/usr/bin/env bash
# the hunt for "the unsharded reference differs between ranks": 8 processes on one device, the same unsharded filter in each
set -u
OUT=gpurun_out/r06h
mkdir -p $OUT
export PYTHONPATH=$PWD HSA_ENABLE_IPC_MODE_LEGACY=0
run() {  # name, env..., -- args
  local name=$1; shift
  local envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 420 python tools/contention_soak.py "$@" > $OUT/$name.jsonl 2> $OUT/$name.err
  echo "$name rc=$? rounds=$(wc -l < $OUT/$name.jsonl) unequal=$(grep -c '"equal": false' $OUT/$name.jsonl)" | tee -a $OUT/summary.txt
  grep '"equal": false' $OUT/$name.jsonl | head -3 | cut -c1-900 | tee -a $OUT/summary.txt
}
# some earlier tenants of the device's memory, as in the failing pytest sessions
timeout 300 python -m pytest tests/test_gpu_world8.py -q -m gpu -k "not bench_eight" --timeout 300 > $OUT/pre_pytest.txt 2>&1; echo "pre pytest: $(tail -1 $OUT/pre_pytest.txt)" | tee -a $OUT/summary.txt
run shards_cu RR_P2P_CU_PARTITION=1 RR_P2P_TIMEOUT_MS=30000 -- --procs 8 --rounds 12 --particles 2000000 --steps 12 --shards --port 29701
run plain X=1 -- --procs 8 --rounds 12 --particles 2000000 --steps 12 --port 29702
run shards_cu_poison RR_P2P_CU_PARTITION=1 RR_P2P_TIMEOUT_MS=30000 RR_DEBUG_POISON_ALLOC=1 -- --procs 8 --rounds 8 --particles 2000000 --steps 12 --shards --port 29703
run shards_cu_multilaunch RR_P2P_CU_PARTITION=1 RR_P2P_TIMEOUT_MS=30000 RR_PF_FUSED_PLAN=0 -- --procs 8 --rounds 8 --particles 2000000 --steps 12 --shards --port 29704
run shards_cu_eachstep RR_P2P_CU_PARTITION=1 RR_P2P_TIMEOUT_MS=30000 -- --procs 8 --rounds 8 --particles 2000000 --steps 12 --shards --each-step --port 29705
