cd $GRAFT_REPO_ROOT
for i in 1 2 3 4; do
python -m pytest tests/test_gpu_p2p.py -q 2>&1 | grep -E "passed|failed|MISMATCH|differ" | cut -c1-1500 | head -8
done
