#!/usr/bin/env bash
set -u
OUT=gpurun_out/r06q
mkdir -p $OUT
export PYTHONPATH=$PWD HSA_ENABLE_IPC_MODE_LEGACY=0
for cl in 6 7 8 9; do for grid in default 512 1024; do
  E="RR_MN_COARSE_LOG2=$cl"; [ $grid != default ] && E="$E RR_MN_GRID=$grid"
  env $E timeout 300 python bench.py --gpus 1 --force-sharded --transport p2p-only --scheme multinomial --no-extra-legs --no-cpu-baseline > $OUT/mn_${cl}_$grid.json 2> $OUT/mn_${cl}_$grid.err
  python - "$cl" "$grid" <<'PY' | tee -a $OUT/summary.txt
import json,sys
try:
    d=json.load(open('bench_legs.json')); k=d.get('kernel_ms_avg') or {}
    print('coarse_log2',sys.argv[1],'grid',sys.argv[2],'ms/step',round(d.get('ms_per_step'),5),'push',round(k.get('k_resample_gather',0),5), {a:round(b,4) for a,b in k.items()})
except Exception as e: print(sys.argv[1:],e)
PY
done; done
