#!/usr/bin/env bash
# Why is k_fs1_observe at 0.66 of the HBM peak at 1e5 x 200 and at 0.61 at 1e6 x 200?  Collects, at both sizes:
#   * the counters this rocprofv3 offers for address translation and memory-side requests (looked up in `rocprofv3 -L`, one
#     pass per small group, kernel trace only beside them)
#   * the access pattern alone (tools/ubench/plane_layout_tlb): planes vs particle-blocked layouts at both sizes
# into gpurun_out/<tag>/; the write-up is profiles/r06_fs1_tlb.md.     bash tools/fs1_tlb_probe.sh r06_fs1_tlb
set -u
TAG=${1:-r06_fs1_tlb}
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp PYTHONPATH=$REPO
(cd /tmp && rocprofv3 -L > "$OUT/counters_available.txt" 2>&1)
grep -o -i -E "\b(TCP_UTCL1[A-Z0-9_]*|TCP_[A-Z0-9_]*STALL[A-Z0-9_]*|TCC_EA0?_RDREQ[A-Z0-9_]*|TCC_EA0?_WRREQ[A-Z0-9_]*|TCC_(HIT|MISS|REQ|READ|WRITE)[A-Z0-9_]*|TCC_TAG_STALL[A-Z0-9_]*|TCC_EA0?_[A-Z0-9_]*STALL[A-Z0-9_]*|TCP_TCC_READ_REQ[A-Z0-9_]*|TCP_PENDING_STALL_CYCLES|TCP_GATE_EN[12]|TCP_TOTAL_CACHE_ACCESSES|TCP_TA_TCP_STATE_READ|GRBM_GUI_ACTIVE|UTCL2[A-Z0-9_]*|VML2[A-Z0-9_]*)\b" \
  "$OUT/counters_available.txt" | sort -u > "$OUT/counters_of_interest.txt"
echo "counters of interest: $(wc -l < "$OUT/counters_of_interest.txt")"
# the access pattern by itself
./tools/ubench/plane_layout_tlb 100000 25 > "$OUT/${TAG}_layout_1e5.json" 2> "$OUT/layout_1e5.err"
./tools/ubench/plane_layout_tlb 1000000 3 25 > "$OUT/${TAG}_layout_1e6.json" 2> "$OUT/layout_1e6.err"
run() {  # name, rocprof args..., -- command
  local name=$1; shift
  (cd /tmp && timeout 900 rocprofv3 -d "$OUT/raw_$name" -o p --output-format csv "$@" > "$OUT/$name.out" 2> "$OUT/$name.err")
}
SMALL="python $REPO/bench.py --workload fastslam --no-cpu-baseline --no-breakdown --steps 10 --warmup 3"
BIG="python $REPO/bench.py --workload fastslam --particles 1000000 --no-cpu-baseline --no-breakdown --steps 6 --warmup 2"
pass() {  # group name, counters...
  local g=$1; shift
  local have=()
  for c in "$@"; do grep -q -x "$c" "$OUT/counters_of_interest.txt" && have+=("$c"); done
  [ ${#have[@]} -eq 0 ] && { echo "group $g: none of [$*] offered"; return; }
  for size in small big; do
    local cmd=$SMALL; [ $size = big ] && cmd=$BIG
    run ${g}_$size --kernel-trace --pmc "${have[@]}" -- $cmd
    local csv=$(find "$OUT/raw_${g}_$size" -name "p_counter_collection.csv" | head -1)
    if [ -n "$csv" ]; then python tools/pmc_avg.py "$csv" k_fs1_observe | sed "s/^/$size,/" >> "$OUT/${TAG}_counters.csv"; else echo "group $g $size: no counter file ($(tail -1 "$OUT/${g}_$size.err"))"; fi
  done
  echo "group $g: ${have[*]}"
}
echo "size,kernel,counter,avg_per_dispatch,dispatches" > "$OUT/${TAG}_counters.csv"
pass utcl1 TCP_UTCL1_TRANSLATION_MISS TCP_UTCL1_TRANSLATION_HIT TCP_UTCL1_REQUEST TCP_UTCL1_PERMISSION_MISS
pass utcl1_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum
pass tcpstall TCP_PENDING_STALL_CYCLES TCP_TCR_TCP_STALL_CYCLES TCP_UTCL1_STALL_INFLIGHT_MAX TCP_UTCL1_STALL_LRU_INFLIGHT TCP_UTCL1_STALL_MULTI_MISS
pass tcpstall_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum
pass tccreq TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum
pass tccstall TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_STALL_sum TCC_TAG_STALL_sum
pass grbm GRBM_GUI_ACTIVE
rm -rf "$OUT"/raw_*
ls "$OUT"
