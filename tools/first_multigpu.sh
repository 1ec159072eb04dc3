#!/usr/bin/env bash
# FIRST CONTACT with a node that has more than one MI355X: one command from "unmeasured" to a scaling table.
#     bash tools/first_multigpu.sh [tag]          (from the repo root; writes gpurun_out/<tag>/, default tag first_multigpu)
# 1. the two-device tests (tests/test_gpu_two_devices.py: peer-to-peer and RCCL, MCL and FastSLAM, bit-identical to the unsharded
#    filter ACROSS PHYSICAL DEVICES -- skipped on every box this repository has seen so far)
# 2. the world-8 tests on device 0 again (tests/test_gpu_world8.py: same code, now next to real peers' results)
# 3. bench.py --gpus 1 / 2 / 4 / 8 (as many as the node has), the driver's own command line, with NCCL_DEBUG=INFO so that RCCL's
#    view of the communicator can be read back
# 4. tools/first_multigpu.sh's table: per N the ms_per_step, value, speed-up over N = 1 (weak scaling: value_N / value_1), the
#    transport the ladder chose, ranks seen / peers mapped / distinct devices (from the line: `sharded.ranks_seen`), RCCL ranks
#    (parsed from NCCL_DEBUG), give-ups (`p2p_timed_out`, transport_note) -- as markdown and JSON
set -u
TAG=${1:-first_multigpu}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export PYTHONPATH=$PWD HSA_ENABLE_IPC_MODE_LEGACY=0
NDEV=$(python -c "from rust_robotics_amd import _ffi; print(_ffi.lib().rr_device_count())")
echo "devices visible: $NDEV" | tee "$OUT/summary.txt"
{ /opt/rocm/bin/rocm-smi --showtopo 2>/dev/null || true; } > "$OUT/topology.txt"
if [ "$NDEV" -ge 2 ]; then
  timeout 3600 python -m pytest tests/test_gpu_two_devices.py -q -m gpu -x > "$OUT/two_device_tests.txt" 2>&1
  echo "two-device tests rc=$?: $(tail -1 "$OUT/two_device_tests.txt")" | tee -a "$OUT/summary.txt"
else
  echo "two-device tests: SKIPPED (one device)" | tee -a "$OUT/summary.txt"
fi
timeout 5400 python -m pytest tests/test_gpu_world8.py -q -m gpu > "$OUT/world8_tests.txt" 2>&1
echo "world-8-on-one-device tests rc=$?: $(tail -1 "$OUT/world8_tests.txt")" | tee -a "$OUT/summary.txt"
for N in 1 2 4 8; do
  [ "$N" -le "$NDEV" ] || continue
  RR_BENCH_LEGS_FILE="$OUT/bench_gpus${N}_legs.json" NCCL_DEBUG=INFO NCCL_DEBUG_FILE="$OUT/nccl_gpus${N}.%h.%p.log" \
    timeout 2400 python bench.py --gpus "$N" --steps 20 --warmup 5 > "$OUT/bench_gpus${N}.out" 2> "$OUT/bench_gpus${N}.err"
  echo "bench --gpus $N rc=$?" | tee -a "$OUT/summary.txt"
done
python tools/first_multigpu_table.py "$OUT" | tee -a "$OUT/summary.txt"
