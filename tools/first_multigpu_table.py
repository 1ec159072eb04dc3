#!/usr/bin/env python3
"""The table of tools/first_multigpu.sh: reads <dir>/bench_gpus<N>.out (bench.py's stdout: legs, then the compact line),
<dir>/bench_gpus<N>.err and <dir>/nccl_gpus<N>.*.log (NCCL_DEBUG=INFO) and prints, per N: ms_per_step, whole-job value, speed-up over
N = 1 (weak scaling: value_N / value_1; the driver computes efficiency itself), transport, ranks seen / peers connected / distinct
devices, RCCL ranks, give-ups.  Writes <dir>/scaling_table.{md,json}.     python tools/first_multigpu_table.py gpurun_out/first_multigpu"""
import glob
import json
import os
import re
import sys


def read_bench(path):
    if not os.path.exists(path):
        return None, {}
    lines = [ln for ln in open(path).read().splitlines() if ln.startswith("{")]
    if not lines:
        return None, {}
    legs = {}
    for ln in lines[:-1]:
        try:
            d = json.loads(ln)
            legs[d.get("leg")] = d
        except ValueError:
            pass
    return json.loads(lines[-1]), legs


def rccl_ranks(d, n):
    """ranks RCCL reports having initialised (NCCL_DEBUG=INFO: 'comm 0x... rank R nranks N ... - Init COMPLETE')"""
    seen = set()
    for p in glob.glob(os.path.join(d, f"nccl_gpus{n}.*.log")) + [os.path.join(d, f"bench_gpus{n}.err")]:
        if os.path.exists(p):
            for m in re.finditer(r"rank (\d+) nranks (\d+)[^\n]*Init COMPLETE", open(p, errors="replace").read()):
                if int(m.group(2)) == n:
                    seen.add(int(m.group(1)))
    return len(seen)


def main():
    d = sys.argv[1]
    rows, base = [], None
    for n in (1, 2, 4, 8):
        line, legs = read_bench(os.path.join(d, f"bench_gpus{n}.out"))
        if line is None:
            continue
        sh = (legs.get("sharded") or line.get("sharded") or {})
        seen = sh.get("ranks_seen") or {}
        if n == 1:
            base = line.get("value")
        note = str(sh.get("transport_note", ""))
        giveups = int(bool(sh.get("p2p_timed_out"))) + len(re.findall(r"gave up|timed out|FAILED validation", note))
        rows.append(dict(n_gpus=n, ms_per_step=line.get("ms_per_step"), value=line.get("value"), unit=line.get("unit"),
                         speedup_vs_1=(line["value"] / base if base and line.get("value") else None),
                         transport=sh.get("transport", "none (unsharded)" if n == 1 else "?"),
                         ranks_seen=seen.get("ranks", 1 if n == 1 else None), peers_connected_min=seen.get("peers_connected_min"),
                         distinct_devices=seen.get("distinct_devices", 1 if n == 1 else None), rccl_ranks_init_complete=rccl_ranks(d, n),
                         give_ups=giveups, shared_device=line.get("shared_device"), note=note[:300]))
    md = ["| N | ms/step | updates/s (whole job) | x over N=1 | transport | ranks seen | peers connected | devices | RCCL ranks | give-ups |",
          "|---|---|---|---|---|---|---|---|---|---|"]
    for r in rows:
        md.append(f"| {r['n_gpus']} | {r['ms_per_step']:.4f} | {r['value']:.4g} | " + (f"{r['speedup_vs_1']:.2f}" if r["speedup_vs_1"] else "-") +
                  f" | {r['transport']} | {r['ranks_seen']} | {r['peers_connected_min']} | {r['distinct_devices']} | {r['rccl_ranks_init_complete']} | {r['give_ups']} |")
    text = "\n".join(md)
    print(text)
    open(os.path.join(d, "scaling_table.md"), "w").write(text + "\n")
    json.dump(rows, open(os.path.join(d, "scaling_table.json"), "w"), indent=1)
    bad = [r for r in rows if r["n_gpus"] > 1 and (r["ranks_seen"] != r["n_gpus"] or r["give_ups"] or (r["distinct_devices"] or 0) < r["n_gpus"])]
    if bad:
        print("ATTENTION: " + "; ".join(f"N={r['n_gpus']}: ranks_seen {r['ranks_seen']}, devices {r['distinct_devices']}, give-ups {r['give_ups']}" for r in bad))


if __name__ == "__main__":
    main()
