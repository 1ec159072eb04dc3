#!/usr/bin/env bash
# round 6, late: feasibility of drawing the NEXT step's motion noise in the shadow of the (latency-bound) plan kernel:
# (a) the plan kernel held to 64 VGPRs (so that something else fits beside it), (b) k_step_lazy with its noise loaded instead of drawn (timing only)
set -u
export PYTHONPATH=$PWD HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/r06z4
mkdir -p $OUT
for i in 1 2; do
  for lib in rust_robotics_amd/librust_robotics_amd.so build_ab/lib_plan64.so build_ab/lib_nonoise.so; do
    RR_AMD_LIBRARY=$PWD/$lib timeout 300 python bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline --no-extra-legs --no-sharded-world1 2>$OUT/err.txt | tail -n 1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$lib', round(d['ms_per_step']*1e3,2), 'us/step; k_step_lazy<EST>', round(d['roofline']['avg_kernel_ms']*1e3,2), '; plain step', d.get('legs',{}).get('plain_async_step',[None])[0])" | tee -a $OUT/ab.txt
  done
done
