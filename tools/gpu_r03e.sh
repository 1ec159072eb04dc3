cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03e
python -m pytest tests/test_gpu_sharded.py tests/test_gpu_p2p.py -q 2>&1 | tail -30
for tr in rccl p2p-only; do
python bench.py --gpus 1 --force-sharded --transport $tr --no-extra-legs --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().splitlines()[-1]); print('$tr', d['ms_per_step'], d['kernel_ms_avg'])"
done
