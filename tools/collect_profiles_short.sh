#!/bin/bash
# The part of tools/collect_profiles.sh that a change to the PLAN kernels invalidates -- kernel timings of the headline and of the
# sharded step, the un-profiled bench lines -- without the PMC passes and the FastSLAM / multinomial / L-sweep legs (a few minutes):
#   tools/collect_profiles_short.sh r05p
set -u
TAG=${1:-r05p}
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
MCL="python $REPO/bench.py --no-cpu-baseline --no-breakdown --no-extra-legs"
run() {  # name, rocprof args..., -- command
  local name=$1; shift
  (cd /tmp && rocprofv3 -d "$OUT/raw_$name" -o p --output-format csv "$@" > "$OUT/$name.bench.json" 2> "$OUT/$name.err")
}
find_csv() { find "$OUT/raw_$1" -name "p_$2.csv" | head -1; }
run mcl_trace --kernel-trace --stats -- $MCL
python tools/summarize_rocprof.py stats "$(find_csv mcl_trace kernel_trace)" > "$OUT/${TAG}_mcl_1e6x32_kernel_stats.csv"
grep '^{' "$OUT/mcl_trace.bench.json" | tail -1 > "$OUT/${TAG}_mcl_1e6x32_bench_under_rocprof.json"
run p2p_trace --kernel-trace --stats -- python $REPO/bench.py --gpus 1 --force-sharded --transport p2p-only --no-extra-legs --no-cpu-baseline --no-breakdown
python tools/summarize_rocprof.py stats "$(find_csv p2p_trace kernel_trace)" > "$OUT/${TAG}_mcl_sharded_p2p_world1_kernel_stats.csv"
rm -rf "$OUT"/raw_*
python bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/${TAG}_bench_driver_command.json" 2> "$OUT/bench_driver.err"
python bench.py > "$OUT/${TAG}_bench_default.json" 2> "$OUT/bench_default.err"
{ for tr in p2p-only rccl; do python bench.py --gpus 1 --force-sharded --transport $tr --no-extra-legs --no-cpu-baseline 2>/dev/null | tail -1; done; } > "$OUT/${TAG}_bench_sharded_world1_all_legs.json"
ls -la "$OUT"
