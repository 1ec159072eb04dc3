#!/usr/bin/env bash
# round 6, late: FastSLAM 2.0 at 4e6 / 8e6 x 200 on one GPU; FastSLAM 2.0 as eight linked shards at configs[3]'s size; FastSLAM 1.0 as 8 x 250 000 x 200
set -u
export PYTHONPATH=$PWD HSA_ENABLE_IPC_MODE_LEGACY=0 GPU_MAX_HW_QUEUES=12
OUT=gpurun_out/r06z10
mkdir -p $OUT
timeout 300 python tools/max_size_probe_fastslam.py 8e6 3e6 2 > $OUT/r06z10_max_size_probe_fastslam2.jsonl 2> $OUT/fs2.err; echo "fs2 probe rc=$?" | tee -a $OUT/summary.txt
cut -c1-400 $OUT/r06z10_max_size_probe_fastslam2.jsonl | tee -a $OUT/summary.txt
timeout 600 python tools/world8_one_device.py fs2-config4 fs1-2e6 > $OUT/r06z10_world8_fastslam.jsonl 2> $OUT/w8.err; echo "world8 rc=$?" | tee -a $OUT/summary.txt
cut -c1-500 $OUT/r06z10_world8_fastslam.jsonl | tee -a $OUT/summary.txt
tail -n 4 $OUT/fs2.err; tail -n 4 $OUT/w8.err
