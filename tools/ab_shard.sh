#!/bin/bash
# A/B of two builds of the library on the sharded world-1 p2p step (bench.py --force-sharded), alternating, same box
cd /root/repo
for i in 1 2 3; do
  for lib in build_ab/librust_robotics_amd_prev.so rust_robotics_amd/librust_robotics_amd.so; do
    RR_AMD_LIBRARY=$PWD/$lib python bench.py --gpus 1 --steps 200 --warmup 20 --force-sharded --transport p2p-only --no-cpu-baseline --no-extra-legs 2>/dev/null | tail -n 1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$lib', round(d['ms_per_step']*1e3,2), 'us/step')"
  done
done
