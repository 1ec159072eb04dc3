cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=8 TMPDIR=/tmp
R=$PWD
export PYTHONPATH=$R
for rep in 1 2 3; do
for pad in 0 57344; do
RR_SHARD_PLAN_LDS_PAD=$pad python bench.py --gpus 1 --force-sharded --transport p2p-only --no-extra-legs --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().splitlines()[-1]); print('pad=$pad p2p world1', round(d['ms_per_step']*1e3,1), {k:round(v*1e3,1) for k,v in d['kernel_ms_avg'].items()})"
done
done
