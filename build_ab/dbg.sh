for v in "$@"; do
  if [ $v = default ]; then unset RR_AMD_LIBRARY; else export RR_AMD_LIBRARY=$PWD/build_ab/lib_$v.so; fi
  for L in 1 32; do python bench.py --workload mcl --landmarks $L --no-cpu-baseline --no-extra-legs 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], sys.argv[2], round(d['ms_per_step']*1000,2), round(d['roofline']['avg_kernel_ms']*1000,2))" $v $L; done; done
