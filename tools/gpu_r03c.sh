cd $GRAFT_REPO_ROOT
bench2() {
for i in $(seq 1 $1); do
RR_BENCH_SHARE_DEVICE=1 RR_BENCH_DEADLINE_S=240 timeout 300 python bench.py --gpus 2 --steps 20 --warmup 5 --particles 100000 --no-extra-legs --no-cpu-baseline > /tmp/o.json 2> /tmp/e.log
python - <<'PY'
import json
try:
    d=json.loads(open('/tmp/o.json').read().splitlines()[-1])
    s=d['config']['sharding']
    print('OK' if 'peer-to-peer transport validated' in s else 'FAIL '+s[-300:], d['ms_per_step'])
except Exception as e:
    print('ERR', e); import re; t=open('/tmp/e.log').read(); print([l for l in t.splitlines() if 'fault' in l or 'Error' in l or 'error' in l][:8])
PY
done
}
echo "==== S1 two hogs, one-launch"
(python tools/hog.py 15 A > /tmp/hA.log 2>&1 &  python tools/hog.py 15 B > /tmp/hB.log 2>&1 & wait); tail -n 3 /tmp/hA.log; tail -n 3 /tmp/hB.log
echo "==== S2 hog 2-launch + bench x8"
RR_PF_FUSED_PLAN=0 python tools/hog.py 50 C > /tmp/hC.log 2>&1 &
sleep 5; bench2 8; wait; tail -n 3 /tmp/hC.log
