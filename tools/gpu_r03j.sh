cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
RR_BENCH_SHARE_DEVICE=1 GPU_MAX_HW_QUEUES=8 timeout 200 python bench.py --gpus 2 --steps 100 --warmup 20 --particles 1000000 --no-cpu-baseline --no-extra-legs > /tmp/o.log 2>/tmp/e.log; echo rc=$?; tail -1 /tmp/o.log | cut -c1-250; grep -n "Error\|error\|Traceback" -A3 /tmp/e.log | grep -v "^--" | head -40
