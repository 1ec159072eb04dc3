#!/usr/bin/env bash
set -u
OUT=gpurun_out/r06j
mkdir -p $OUT
export PYTHONPATH=$PWD HSA_ENABLE_IPC_MODE_LEGACY=0
run() {
  local name=$1; shift
  local envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 300 python tools/contention_soak.py "$@" > $OUT/$name.jsonl 2> $OUT/$name.err
  echo "$name rc=$? rounds=$(grep -c '"round"' $OUT/$name.jsonl) unequal=$(grep -c '"equal": false' $OUT/$name.jsonl)" | tee -a $OUT/summary.txt
  grep '"equal": false' $OUT/$name.jsonl | head -2 | cut -c1-1500 | tee -a $OUT/summary.txt
}
E="RR_P2P_CU_PARTITION=1 RR_P2P_TIMEOUT_MS=30000"
run ladder_both $E -- --procs 8 --rounds 10 --particles 2000000 --steps 12 --shards --ladder native,torch --port 29721
run ladder_both_every $E -- --procs 8 --rounds 6 --particles 2000000 --steps 12 --shards --ladder native,torch --ladder-every-round --port 29722
run ladder_native $E -- --procs 8 --rounds 10 --particles 2000000 --steps 12 --shards --ladder native --port 29723
run ladder_torch $E -- --procs 8 --rounds 10 --particles 2000000 --steps 12 --shards --ladder torch --port 29724
run ladder_both_noshards X=1 -- --procs 8 --rounds 10 --particles 2000000 --steps 12 --ladder native,torch --port 29725
