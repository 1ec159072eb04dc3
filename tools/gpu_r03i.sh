cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python -m pytest tests/test_gpu_fs1_parity.py tests/test_gpu_fs2_parity.py tests/test_gpu_fs1_sharded.py -q -x 2>&1 | tail -2
RR_FS1_BANDS=3 python -m pytest tests/test_gpu_fs1_parity.py -q -x 2>&1 | tail -2
for rep in 1 2; do
for cfg in "build_ab/librust_robotics_amd_head.so 1" "rust_robotics_amd/librust_robotics_amd.so 1" "rust_robotics_amd/librust_robotics_amd.so 2" "rust_robotics_amd/librust_robotics_amd.so 4" "rust_robotics_amd/librust_robotics_amd.so 8"; do
set -- $cfg
RR_FS1_BANDS=$2 RR_AMD_LIBRARY=$PWD/$1 python bench.py --workload fastslam --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().splitlines()[-1]); print('$1 bands=$2', round(d['ms_per_step'],4), round(d['roofline']['frac'],4), {k:round(v,4) for k,v in d.get('kernel_ms_avg',{}).items()})"
done
done
