#!/usr/bin/env python3
"""Turn rocprofv3 CSV output into the per-kernel summaries committed under profiles/.

  stats   <kernel_trace.csv>                       per-kernel launches / total / average / min / max duration (what
                                                   `rocprofv3 --kernel-trace --stats` prints, from the trace itself)
  hbm     <workload> <fetch_counter.csv> <write_counter.csv>
                                                   HBM traffic per launch from two separate --pmc passes (FETCH_SIZE and
                                                   WRITE_SIZE, KB per dispatch); the read side is doubled as
                                                   MI355X_MICROARCH.md "HBM" prescribes for gfx950
  sq      <counter.csv>                            per-kernel averages of an SQ counter pass + derived ratios

Kernel names are shortened to `name<template args>`.
"""
import csv
import re
import sys
from collections import defaultdict


def short(name):
    name = name.strip().strip('"')
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"^((?:rr::)?[A-Za-z_0-9]+)(<[^(]*>)?\(", name)
    if m:
        return (m.group(1).replace("rr::", "") + (m.group(2) or "")).replace(" ", "")
    return name.split("(")[0][:60]


def stats(path):
    acc = defaultdict(list)
    for r in csv.DictReader(open(path)):
        acc[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    total = sum(sum(v) for v in acc.values())
    w = csv.writer(sys.stdout)
    w.writerow(["kernel", "launches", "total_us", "avg_us", "min_us", "max_us", "pct_of_gpu_time"])
    for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
        w.writerow([k, len(v), f"{sum(v):.1f}", f"{sum(v) / len(v):.2f}", f"{min(v):.2f}", f"{max(v):.2f}", f"{100 * sum(v) / total:.1f}"])


def counters(path):
    acc = defaultdict(lambda: defaultdict(list))
    for r in csv.DictReader(open(path)):
        acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return acc


def hbm(workload, fetch_csv, write_csv):
    f, wr = counters(fetch_csv), counters(write_csv)
    w = csv.writer(sys.stdout)
    w.writerow(["workload", "kernel", "dispatches", "FETCH_SIZE_KB_avg_raw", "WRITE_SIZE_KB_avg_raw", "read_MB_corrected_x2", "write_MB"])
    for k in f:
        fs = f[k].get("FETCH_SIZE", [])
        ws = wr.get(k, {}).get("WRITE_SIZE", [])
        if not fs:
            continue
        fa = sum(fs) / len(fs)
        wa = sum(ws) / len(ws) if ws else 0.0
        w.writerow([workload, k, len(fs), f"{fa:.1f}", f"{wa:.1f}", f"{2 * fa * 1024 / 1e6:.2f}", f"{wa * 1024 / 1e6:.2f}"])


def sq(path):
    acc = counters(path)
    w = csv.writer(sys.stdout)
    w.writerow(["kernel", "launches", "waves", "valu_insts_per_wave", "salu_insts_per_wave", "lds_insts_per_wave",
                "valu_active_over_wave_cycles", "wait_inst_any_over_wave_cycles", "wave_cycles_over_busy_cycles"])
    for k, c in acc.items():
        def avg(name):
            v = c.get(name, [])
            return sum(v) / len(v) if v else 0.0

        waves = avg("SQ_WAVES")
        if waves <= 0:
            continue
        wc, busy = avg("SQ_WAVE_CYCLES"), avg("SQ_BUSY_CYCLES")
        w.writerow([k, len(c.get("SQ_WAVES", [])), f"{waves:.0f}", f"{avg('SQ_INSTS_VALU') / waves:.0f}", f"{avg('SQ_INSTS_SALU') / waves:.0f}",
                    f"{avg('SQ_INSTS_LDS') / waves:.0f}", f"{avg('SQ_ACTIVE_INST_VALU') / wc:.3f}" if wc else "",
                    f"{avg('SQ_WAIT_INST_ANY') / wc:.3f}" if wc else "", f"{wc / busy:.2f}" if busy else ""])


if __name__ == "__main__":
    cmd = sys.argv[1] if len(sys.argv) > 1 else ""
    if cmd == "stats":
        stats(sys.argv[2])
    elif cmd == "hbm":
        hbm(sys.argv[2], sys.argv[3], sys.argv[4])
    elif cmd == "sq":
        sq(sys.argv[2])
    else:
        sys.exit(__doc__)
