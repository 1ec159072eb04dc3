#!/bin/bash
# A/B of several builds of the library on one box, alternating: tools/ab_libs.sh "<bench.py arguments>" lib1.so lib2.so ...
cd /root/repo
ARGS=$1; shift
for i in 1 2 3; do
  for lib in "$@"; do
    RR_AMD_LIBRARY=$PWD/$lib python bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline --no-extra-legs $ARGS 2>/dev/null | tail -n 1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$lib', round(d['ms_per_step']*1e3,2), 'us/step; dominant kernel', round(d['roofline']['avg_kernel_ms']*1e3,2), '; plain', d.get('legs',{}).get('plain_async_step',[None])[0])"
  done
done
