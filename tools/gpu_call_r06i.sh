#!/usr/bin/env bash
set -u
OUT=gpurun_out/r06i
mkdir -p $OUT
export PYTHONPATH=$PWD HSA_ENABLE_IPC_MODE_LEGACY=0
# (1) the failing session itself, up to 4 times, stop at the first failure and keep what every rank saw
for i in 1 2 3 4; do
  timeout 600 python -m pytest tests/test_gpu_world8.py tests/test_gpu_p2p.py -q -m gpu --timeout 600 > $OUT/pytest_w8_p2p_$i.txt 2>&1; rc=$?
  echo "session $i rc=$rc: $(tail -1 $OUT/pytest_w8_p2p_$i.txt)" | tee -a $OUT/summary.txt
  cp gpurun_out/test_bench_eight_ranks.stderr.txt $OUT/bench8_stderr_$i.txt 2>/dev/null
  if [ $rc -ne 0 ]; then grep -h "VALIDATION MISMATCH\|UNSHARDED REFERENCE" $OUT/bench8_stderr_$i.txt | head -12 | cut -c1-900 | tee -a $OUT/summary.txt; break; fi
done
run() {
  local name=$1; shift
  local envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 420 python tools/contention_soak.py "$@" > $OUT/$name.jsonl 2> $OUT/$name.err
  echo "$name rc=$? rounds=$(grep -c '"round"' $OUT/$name.jsonl) unequal=$(grep -c '"equal": false' $OUT/$name.jsonl)" | tee -a $OUT/summary.txt
  grep '"equal": false' $OUT/$name.jsonl | head -3 | cut -c1-1200 | tee -a $OUT/summary.txt
}
run tenant_shards RR_P2P_CU_PARTITION=1 RR_P2P_TIMEOUT_MS=30000 -- --procs 8 --rounds 12 --particles 2000000 --steps 12 --shards --tenant --port 29711
run tenant_plain X=1 -- --procs 8 --rounds 12 --particles 2000000 --steps 12 --tenant --port 29712
# (3) the in-process worlds against poisoned allocations of several kinds (the null-stream deadlock of the first poison run is gone)
for pat in 1 0x3f 0xff 0x01; do
  RR_DEBUG_POISON_ALLOC=$pat timeout 900 python -m pytest tests/test_gpu_p2p.py tests/test_gpu_fs1_sharded.py tests/test_gpu_pf_parity.py tests/test_gpu_multinomial_lazy.py tests/test_gpu_edge_sizes.py -q -m gpu --timeout 600 -x > $OUT/poison_$pat.txt 2>&1
  echo "poison $pat rc=$?: $(tail -1 $OUT/poison_$pat.txt)" | tee -a $OUT/summary.txt
  grep -E "^FAILED|^ERROR" $OUT/poison_$pat.txt | head -5 | cut -c1-300 | tee -a $OUT/summary.txt
done
