#!/usr/bin/env bash
set -u
OUT=gpurun_out/r06l
mkdir -p $OUT
export PYTHONPATH=$PWD HSA_ENABLE_IPC_MODE_LEGACY=0
bench() {  # name, extra env..., -- extra args
  local name=$1; shift
  local envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env RR_BENCH_SHARE_DEVICE=1 RR_BENCH_DEADLINE_S=600 RR_P2P_CU_PARTITION=1 RR_P2P_TIMEOUT_MS=30000 RR_BENCH_VALIDATE_REPEAT=3 "${envs[@]}" \
    timeout 300 python bench.py --gpus 8 --steps 20 --warmup 5 --particles 250000 --no-extra-legs --no-cpu-baseline "$@" > $OUT/$name.out 2> $OUT/$name.err
  local rc=$?
  local split=$(grep -c "DIFFERS BETWEEN RANKS" $OUT/$name.err)
  echo "bench $name rc=$rc split_lines=$split $(grep -h 'DIFFERS BETWEEN RANKS' $OUT/$name.err | head -1 | grep -o 'ranks \[[0-9, ]*\]') validated=$(grep -c 'transport validated' $OUT/$name.err)" | tee -a $OUT/summary.txt
}
pre() {
  timeout 600 python -m pytest tests/test_gpu_world8.py -q -m gpu --timeout 600 -k "$1" > $OUT/pre.txt 2>&1
  echo "pre [$1]: $(tail -1 $OUT/pre.txt)" | tee -a $OUT/summary.txt
}
bench fresh X=1 --
pre "ipc_handles_fastslam"; bench after_ipc_fs X=1 --
pre "ipc_handles_mcl"; bench after_ipc_mcl X=1 --
pre "config4"; bench after_config4 X=1 --
pre "config5"; bench after_config5 X=1 --
pre "small_and_worst or multinomial or multi_launch"; bench after_small X=1 --
bench again1 X=1 --
bench again2 X=1 --
PRE="not bench_eight"
pre "$PRE"; bench full_asis X=1 --
pre "$PRE"; bench full_p2ponly X=1 -- --transport p2p-only
pre "$PRE"; bench full_nopart RR_P2P_CU_PARTITION=0 --
pre "$PRE"; bench full_poison RR_DEBUG_POISON_ALLOC=1 --
pre "$PRE"; bench full_nop2psteps RR_BENCH_HUNT_NO_P2P_STEPS=1 --
