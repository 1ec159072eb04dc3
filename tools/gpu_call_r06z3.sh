#!/usr/bin/env bash
# round 6, late: what 10 Philox rounds instead of 7 would cost, and three scheduler strategies of the code generator, against the in-tree build
set -u
export PYTHONPATH=$PWD HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/r06z3
mkdir -p $OUT
for i in 1 2; do
  for lib in rust_robotics_amd/librust_robotics_amd.so build_ab/lib_philox10.so build_ab/lib_sched_max-ilp.so build_ab/lib_sched_iterative-ilp.so build_ab/lib_sched_max-memory-clause.so; do
    RR_AMD_LIBRARY=$PWD/$lib timeout 300 python bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline --no-extra-legs --no-sharded-world1 2>/dev/null | tail -n 1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$lib', round(d['ms_per_step']*1e3,2), 'us/step; k_step_lazy<EST>', round(d['roofline']['avg_kernel_ms']*1e3,2), '; plain step', d.get('legs',{}).get('plain_async_step',[None])[0])" | tee -a $OUT/ab.txt
  done
done
