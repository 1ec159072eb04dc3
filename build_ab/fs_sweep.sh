run(){ python bench.py --workload $WL --no-cpu-baseline --no-breakdown 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], round(d['ms_per_step']*1000,1), 'kernel', round(d['roofline']['avg_kernel_ms']*1000,1), 'chunks', d.get('obs_chunks'))" "$1"; }
for WL in fastslam fastslam2; do
for tw in 20000 30000 45000 60000 90000; do RR_FS1_TARGET_WAVES=$tw run "$WL tw$tw"; done
for v in 6 1; do RR_FS1_VARIANT=$v run "$WL var$v"; done
done
