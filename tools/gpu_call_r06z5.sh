#!/usr/bin/env bash
# round 6, late: how far one 288 GB GPU takes the unsharded MCL filter (tools/max_size_probe.py)
set -u
export PYTHONPATH=$PWD HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/r06z5
mkdir -p $OUT
timeout 420 python tools/max_size_probe.py ${1:-1e9} ${2:-0} ${5:-1} > $OUT/r06z5_max_size_probe.jsonl 2> $OUT/probe.err; echo "probe rc=$?" | tee -a $OUT/summary.txt
cat $OUT/r06z5_max_size_probe.jsonl | cut -c1-400 | tee -a $OUT/summary.txt
tail -3 $OUT/probe.err
rocm-smi --showmeminfo vram 2>/dev/null | grep -i "total\|used" | head -4 | tee -a $OUT/summary.txt
if [ "${3:-}" != "" ]; then
  timeout 420 python tools/max_size_probe_fastslam.py $3 ${4:-0} > $OUT/r06z5_max_size_probe_fastslam.jsonl 2> $OUT/probe_fs.err; echo "fastslam probe rc=$?" | tee -a $OUT/summary.txt
  cat $OUT/r06z5_max_size_probe_fastslam.jsonl | cut -c1-500 | tee -a $OUT/summary.txt
  tail -3 $OUT/probe_fs.err
fi
