#!/bin/bash
# A/B of two engine builds on the MCL legs, alternating, one box:  tools/ab_mcl.sh "<bench args>" default head
args=$1; shift
for i in 1 2; do for v in "$@"; do
  if [ $v = default ]; then unset RR_AMD_LIBRARY; else export RR_AMD_LIBRARY=$PWD/build_ab/lib_$v.so; fi
  python bench.py --no-cpu-baseline --no-extra-legs $args 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], round(d['ms_per_step']*1000,2), 'us/step', {k: round(v*1e3,2) for k,v in d['kernel_ms_avg'].items()})" $v
done; done
