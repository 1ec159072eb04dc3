cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp GPU_MAX_HW_QUEUES=8 RR_P2P_TIMEOUT_MS=300 RR_BENCH_SHARE_DEVICE=1
timeout 200 python bench.py --gpus 2 --steps 20 --warmup 5 --particles 1000000 --no-cpu-baseline --no-extra-legs > /tmp/o.log 2>/tmp/e.log; echo "rc=$?"
grep "^\[sharded\|^\[bench" /tmp/e.log | head -40
