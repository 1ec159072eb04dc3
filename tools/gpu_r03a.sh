set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03a
python -m pytest tests -m gpu -q --maxfail=40 -x --deselect tests/test_gpu_plan_degrade.py > gpurun_out/r03a/pytest_main.log 2>&1; echo "rc=$?" >> gpurun_out/r03a/pytest_main.log
tail -30 gpurun_out/r03a/pytest_main.log
python -m pytest tests/test_gpu_plan_degrade.py -q > gpurun_out/r03a/pytest_degrade.log 2>&1; echo "rc=$?" >> gpurun_out/r03a/pytest_degrade.log
tail -40 gpurun_out/r03a/pytest_degrade.log
timeout 600 python bench.py > gpurun_out/r03a/bench_default.json 2> gpurun_out/r03a/bench_default.err; echo "bench rc=$?"
tail -5 gpurun_out/r03a/bench_default.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r03a/bench_default.json'))
def g(*k):
    x=d
    for a in k:
        x=x.get(a) if isinstance(x,dict) else None
        if x is None: return None
    return x
print('headline ms', d['ms_per_step'], d['value'], g('headline_step'))
print('plain', g('plain_async_step','ms_per_step'))
print('roofline', g('roofline','frac'), g('roofline','avg_kernel_ms'))
print('kernel_ms', d.get('kernel_ms_avg'))
print('index_parity', g('index_parity','differing_slots_vs_literal_float_walk'))
print('fs', g('fastslam','ms_per_step'), g('fastslam','roofline','frac'), g('fastslam','cpu_baseline','value'))
print('mn', g('mcl_multinomial','ms_per_step'), g('mcl_multinomial','roofline','frac'), g('mcl_multinomial','error'), g('mcl_multinomial','index_parity','differing_slots_vs_literal_float_walk'))
print('sh1', json.dumps(d.get('sharded_world1'))[:1500])
print('cpu', g('cpu_baseline','value'), g('cpu_baseline','other_resampler','value'))
PY
