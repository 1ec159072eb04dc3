#!/bin/bash
# A/B of engine builds on one box, in one gpurun call (box-to-box variance is larger than most kernel changes):
#   tools/ab_bench.sh <mcl|fastslam|fastslam2> <variant>...    variant = default | name of build_ab/lib_<name>.so
wl=$1; shift
run(){ RR_BENCH_NO_COLD=1 python bench.py --workload $wl --no-cpu-baseline --no-extra-legs 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], round(d['ms_per_step']*1000,2), 'kernel', round(d['roofline']['avg_kernel_ms']*1000,2), 'frac', round(d['roofline']['frac'],3), 'est', round(d.get('estimate_every_step',{}).get('ms_per_step',0)*1000,2))" "$1"; }
for i in 1 2; do
for v in "$@"; do
  if [ $v = default ]; then unset RR_AMD_LIBRARY; else export RR_AMD_LIBRARY=$PWD/build_ab/lib_$v.so; fi
  run $v
done; done
