#!/usr/bin/env bash
# round 6, late: the two largest-size tests and the FastSLAM size probe after the fix of rr_fs1_create's fill launch
set -u
export PYTHONPATH=$PWD HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/r06z6
mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_edge_sizes.py::test_a_billion_particles_on_one_gpu tests/test_gpu_fs1_parity.py::test_four_million_particles_times_200_landmarks -m gpu -q --durations=3 > $OUT/pytest.txt 2>&1; echo "pytest rc=$?: $(tail -1 $OUT/pytest.txt)" | tee -a $OUT/summary.txt
grep -E "^FAILED|^ERROR|Error|assert " $OUT/pytest.txt | head -10 | cut -c1-300 | tee -a $OUT/summary.txt
timeout 600 python tools/max_size_probe_fastslam.py 14e6 > $OUT/r06z6_max_size_probe_fastslam.jsonl 2> $OUT/probe_fs.err; echo "fastslam probe rc=$?" | tee -a $OUT/summary.txt
cat $OUT/r06z6_max_size_probe_fastslam.jsonl | cut -c1-500 | tee -a $OUT/summary.txt
tail -3 $OUT/probe_fs.err
