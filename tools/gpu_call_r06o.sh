#!/usr/bin/env bash
set -u
OUT=gpurun_out/r06o
mkdir -p $OUT
P=tools/ubench/memset_sync_probe
echo "alone: $($P 20 1024)" | tee -a $OUT/summary.txt
echo "alone 64 MiB: $($P 50 64)" | tee -a $OUT/summary.txt
for i in 1 2 3 4 5 6 7 8; do $P 20 512 > $OUT/conc_$i.json & done; wait
echo "8 at once:" | tee -a $OUT/summary.txt; cat $OUT/conc_*.json | tee -a $OUT/summary.txt
