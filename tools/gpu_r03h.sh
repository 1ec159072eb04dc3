cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_p2p.py -q 2>&1 | tail -8
