#!/usr/bin/env bash
# round 6, late: eight linked shards of 12.5e6 particles each (1e8 in one global slot space) on one device, eager and CU-partitioned lazy step
set -u
export PYTHONPATH=$PWD HSA_ENABLE_IPC_MODE_LEGACY=0 GPU_MAX_HW_QUEUES=12
OUT=gpurun_out/r06z8
mkdir -p $OUT
free -g | head -2 | tee -a $OUT/summary.txt
timeout 600 python tools/world8_one_device.py mcl-1e8 > $OUT/world8_1e8_eager.jsonl 2> $OUT/eager.err; echo "eager rc=$?" | tee -a $OUT/summary.txt
RR_P2P_CU_PARTITION=1 RR_P2P_TIMEOUT_MS=30000 timeout 600 python tools/world8_one_device.py mcl-1e8 > $OUT/world8_1e8_cu_partition.jsonl 2> $OUT/cu.err; echo "cu-partition rc=$?" | tee -a $OUT/summary.txt
cat $OUT/world8_1e8_eager.jsonl $OUT/world8_1e8_cu_partition.jsonl | cut -c1-700 | tee -a $OUT/summary.txt
tail -n 3 $OUT/eager.err; tail -n 3 $OUT/cu.err
