run(){ python bench.py --workload mcl --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], round(d['ms_per_step']*1000,2), 'k1', round(d['roofline']['avg_kernel_ms']*1000,2), 'est', round(d.get('estimate_every_step',{}).get('ms_per_step',0)*1000,2))" "$1"; }
for i in 1 2; do
for v in "$@"; do
  if [ $v = default ]; then unset RR_AMD_LIBRARY; else export RR_AMD_LIBRARY=$PWD/build_ab/lib_$v.so; fi
  run $v
done; done
