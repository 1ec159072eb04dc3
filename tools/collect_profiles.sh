#!/bin/bash
# Collect the rocprofv3 evidence of one round on a gpurun MI355X box:
#   tools/collect_profiles.sh r02a        (from the repo root; writes gpurun_out/<tag>/, copy the summaries to profiles/)
# Kernel timings (--kernel-trace) and every PMC pass run separately: FETCH_SIZE and WRITE_SIZE do not fit one pass
# (TCC slots, MI355X_MICROARCH.md "rocprofv3 PMC slots"), and no trace domain other than the kernel trace is combined
# with --pmc.
set -u
TAG=${1:-r02}
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
MCL="python $REPO/bench.py --no-cpu-baseline --no-breakdown --no-extra-legs"
FS1="python $REPO/bench.py --workload fastslam --no-cpu-baseline --no-breakdown"
run() {  # name, rocprof args..., -- command
  local name=$1; shift
  (cd /tmp && rocprofv3 -d "$OUT/raw_$name" -o p --output-format csv "$@" > "$OUT/$name.bench.json" 2> "$OUT/$name.err")
}
find_csv() { find "$OUT/raw_$1" -name "p_$2.csv" | head -1; }

run mcl_trace --kernel-trace --stats -- $MCL
run fs1_trace --kernel-trace --stats -- $FS1
python tools/summarize_rocprof.py stats "$(find_csv mcl_trace kernel_trace)" > "$OUT/${TAG}_mcl_1e6x32_kernel_stats.csv"
python tools/summarize_rocprof.py stats "$(find_csv fs1_trace kernel_trace)" > "$OUT/${TAG}_fastslam_1e5x200_kernel_stats.csv"
grep '^{' "$OUT/mcl_trace.bench.json" | tail -1 > "$OUT/${TAG}_mcl_1e6x32_bench_under_rocprof.json"
grep '^{' "$OUT/fs1_trace.bench.json" | tail -1 > "$OUT/${TAG}_fastslam_1e5x200_bench_under_rocprof.json"
run mn_trace --kernel-trace --stats -- $MCL --scheme multinomial
python tools/summarize_rocprof.py stats "$(find_csv mn_trace kernel_trace)" > "$OUT/${TAG}_mcl_1e6x32_multinomial_kernel_stats.csv"
[ -n "${ONLY_TRACE:-}" ] && exit 0  # kernel timings only (the PMC passes take several minutes)

# workload keys are the ones bench.py's measured_traffic() looks up
FS2="python $REPO/bench.py --workload fastslam2 --no-cpu-baseline --no-breakdown"
MCL5="$MCL --particles 16000000 --landmarks 64"
for wl in mcl fs1 fs2 mcl_1000000x32_multinomial mcl_16000000x64_systematic; do
  cmd="$MCL --steps 40 --warmup 5"
  [ $wl = fs1 ] && cmd="$FS1 --steps 40 --warmup 5"
  [ $wl = fs2 ] && cmd="$FS2 --steps 40 --warmup 5"
  [ $wl = mcl_1000000x32_multinomial ] && cmd="$MCL --scheme multinomial --steps 40 --warmup 5"
  [ $wl = mcl_16000000x64_systematic ] && cmd="$MCL5 --steps 10 --warmup 3"
  run ${wl}_fetch --kernel-trace --pmc FETCH_SIZE -- $cmd
  run ${wl}_write --kernel-trace --pmc WRITE_SIZE -- $cmd
  python tools/summarize_rocprof.py hbm $wl "$(find_csv ${wl}_fetch counter_collection)" "$(find_csv ${wl}_write counter_collection)" > "$OUT/hbm_$wl.csv"
done
{ cat "$OUT/hbm_mcl.csv"; for wl in fs1 fs2 mcl_1000000x32_multinomial mcl_16000000x64_systematic; do tail -n +2 "$OUT/hbm_$wl.csv"; done; } > "$OUT/pmc_hbm_traffic_nosha.csv"
# every row carries the hash of the library it was measured on: bench.py's measured_traffic() reports the bytes only when that is the
# library it has loaded (library_sha16 column = first 16 hex digits of the .so's SHA-256)
# (library_sha16 = the hash of the library's SOURCES as the library reports it, benchlib.common.library_sha16: builds are not
# bit-reproducible, sources are)
PYTHONPATH="$REPO" python - "$OUT/pmc_hbm_traffic_nosha.csv" > "$OUT/${TAG}_pmc_hbm_traffic.csv" <<'PY'
import sys
from benchlib.common import library_sha16
sha = library_sha16()
for i, ln in enumerate(open(sys.argv[1]).read().splitlines()):
    print(ln + ("," + ("library_sha16" if i == 0 else sha)))
PY
rm -f "$OUT/pmc_hbm_traffic_nosha.csv"
run mcl5_trace --kernel-trace --stats -- $MCL5 --steps 20 --warmup 3
python tools/summarize_rocprof.py stats "$(find_csv mcl5_trace kernel_trace)" > "$OUT/${TAG}_mcl_1.6e7x64_kernel_stats.csv"
run fs2_trace --kernel-trace --stats -- $FS2
python tools/summarize_rocprof.py stats "$(find_csv fs2_trace kernel_trace)" > "$OUT/${TAG}_fastslam2_1e5x200_kernel_stats.csv"

run mcl_sq --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_LDS -- $MCL --steps 40 --warmup 5
python tools/summarize_rocprof.py sq "$(find_csv mcl_sq counter_collection)" > "$OUT/${TAG}_mcl_pmc_sq_summary.csv"
run fs1_sq --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_LDS -- $FS1 --steps 20 --warmup 5
python tools/summarize_rocprof.py sq "$(find_csv fs1_sq counter_collection)" > "$OUT/${TAG}_fastslam_pmc_sq_summary.csv"

# in-kernel timeline of the one-launch resample plan (instrumented build: make -C rust_robotics_amd/csrc timeline)
if [ -f "$REPO/rust_robotics_amd/librust_robotics_amd_timeline.so" ]; then
  RR_AMD_LIBRARY="$REPO/rust_robotics_amd/librust_robotics_amd_timeline.so" python tools/plan_timeline.py "$OUT/${TAG}_plan_kernel_timeline.json" > /dev/null 2> "$OUT/plan_timeline.err"
fi
if [ -f "$REPO/rust_robotics_amd/librust_robotics_amd_timeline.so" ]; then  # ... and of a shard's (k_shard_plan_mark, world size 1)
  RR_AMD_LIBRARY="$REPO/rust_robotics_amd/librust_robotics_amd_timeline.so" python tools/shard_plan_timeline.py "$OUT/${TAG}_shard_plan_timeline.json" > /dev/null 2> "$OUT/shard_plan_timeline.err"
fi
if [ -f "$REPO/rust_robotics_amd/librust_robotics_amd_timeline.so" ]; then
  RR_AMD_LIBRARY="$REPO/rust_robotics_amd/librust_robotics_amd_timeline.so" python tools/resident_timeline.py > "$OUT/${TAG}_resident_step_timeline.json" 2> "$OUT/resident_timeline.err"
fi
python tools/reference_size_loops.py > "$OUT/${TAG}_reference_size_loops.json" 2> "$OUT/reference_size_loops.err"
python tools/l_sweep.py > "$OUT/${TAG}_mcl_L_sweep.json" 2> "$OUT/l_sweep.err"
./tools/ubench/host_link > "$OUT/${TAG}_host_link_pingpong.json" 2>/dev/null
# un-profiled lines of the same build: the default line, the driver's own command, the sharded legs at world size 1
python bench.py > "$OUT/${TAG}_bench_default.json" 2> "$OUT/bench_default.err"
python bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/${TAG}_bench_driver_command.json" 2> "$OUT/bench_driver.err"
python bench.py --scheme multinomial --no-cpu-baseline --no-extra-legs > "$OUT/${TAG}_mcl_1e6x32_multinomial_bench.json" 2>/dev/null
python bench.py --workload fastslam2 > "$OUT/${TAG}_fastslam2_1e5x200_bench.json" 2>/dev/null
# the sharded step at world size 1, both transports, as their own lines (the default line carries the same two as `sharded_world1`)
{ for tr in p2p-only rccl; do python bench.py --gpus 1 --force-sharded --transport $tr --no-extra-legs --no-cpu-baseline 2>/dev/null | tail -1; done; } > "$OUT/${TAG}_bench_sharded_world1_all_legs.json"
# two ranks of one filter on this one device (development knob RR_BENCH_SHARE_DEVICE: IPC transport, real overhang traffic
# between the ranks; both ranks' kernels share the GPU, so this is a protocol check with a time attached, not a scaling number)
RR_BENCH_SHARE_DEVICE=1 GPU_MAX_HW_QUEUES=8 timeout 300 python bench.py --gpus 2 --steps 300 --warmup 50 --particles 100000 --no-cpu-baseline --no-extra-legs 2>/dev/null | tail -1 > "$OUT/${TAG}_bench_2ranks_shared_device.json"
run p2p_trace --kernel-trace --stats -- python $REPO/bench.py --gpus 1 --force-sharded --transport p2p-only --no-extra-legs --no-cpu-baseline --no-breakdown
python tools/summarize_rocprof.py stats "$(find_csv p2p_trace kernel_trace)" > "$OUT/${TAG}_mcl_sharded_p2p_world1_kernel_stats.csv"
rm -rf "$OUT"/raw_*   # the raw traces are large; the summaries above are what gets committed
ls -la "$OUT"
