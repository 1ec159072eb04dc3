#!/bin/bash
# A/B of two builds of the library on the headline step, alternating, on one box: build_ab/librust_robotics_amd_prev.so (the build of an
# earlier commit: git archive <rev> rust_robotics_amd/csrc include | tar -x -C /tmp/prev; make there; copy the .so) against the in-tree one
cd /root/repo
for i in 1 2 3; do
  for lib in build_ab/librust_robotics_amd_prev.so rust_robotics_amd/librust_robotics_amd.so; do
    RR_AMD_LIBRARY=$PWD/$lib python bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline --no-extra-legs 2>/dev/null | tail -n 1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$lib', 'headline', round(d['ms_per_step']*1e3,2), 'us/step; k_step_lazy', round(d['roofline']['avg_kernel_ms']*1e3,2))"
  done
done
