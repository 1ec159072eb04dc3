cd $GRAFT_REPO_ROOT
for i in $(seq 1 12); do
python bench.py --gpus 1 --steps 20 --warmup 5 --no-extra-legs --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().splitlines()[-1]); print('headline', round(d['ms_per_step']*1e3,2), 'plain', round(d['plain_async_step']['ms_per_step']*1e3,2), 'k1', round(d['roofline']['avg_kernel_ms']*1e3,2))"
done
