#!/bin/bash
# A/B of the multinomial search on one box: guide table over the target space (default) against the coarse-table search
# of the CDF (RR_MN_GUIDE=0).  Same library, the switch is read when a filter first resamples.
run(){ python bench.py --workload mcl --scheme multinomial --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], round(d['ms_per_step']*1000,2), {k: round(v*1000,2) for k,v in d.get('kernel_ms_avg',{}).items()})" "$1"; }
for i in 1 2; do
  for v in "$@"; do
    case $v in
      guide*) export RR_MN_GUIDE=1; lg=${v#guide}; if [ -n "$lg" ]; then export RR_MN_GUIDE_LOG2=$lg; else unset RR_MN_GUIDE_LOG2; fi;;
      coarse) export RR_MN_GUIDE=0;;
    esac
    run $v
  done
done
