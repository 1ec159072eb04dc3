set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03g
python -m pytest tests -m gpu -q --maxfail=40 > gpurun_out/r03g/pytest_main.log 2>&1; echo "rc=$?" >> gpurun_out/r03g/pytest_main.log
tail -30 gpurun_out/r03g/pytest_main.log
timeout 600 python bench.py > gpurun_out/r03g/bench_default.json 2> gpurun_out/r03g/bench_default.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r03g/bench_default.json'))
def g(*k):
    x=d
    for a in k:
        x=x.get(a) if isinstance(x,dict) else None
        if x is None: return None
    return x
print('headline ms', d['ms_per_step'], d['value'])
print('plain', g('plain_async_step','ms_per_step'))
print('roofline', g('roofline','frac'), g('roofline','avg_kernel_ms'))
print('kernel_ms', d.get('kernel_ms_avg'))
print('sh1', json.dumps(d.get('sharded_world1'))[:2500])
PY
