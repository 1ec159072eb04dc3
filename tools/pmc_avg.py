#!/usr/bin/env python3
"""Average rocprofv3 counter values per dispatch, per kernel:  pmc_avg.py <counter_collection.csv> [kernel substring]"""
import csv, sys, collections
rows = csv.DictReader(open(sys.argv[1]))
sub = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: [0.0, 0])
for r in rows:
    k = r["Kernel_Name"]
    if sub not in k: continue
    key = (k.split("(")[0][:60], r["Counter_Name"])
    acc[key][0] += float(r["Counter_Value"]); acc[key][1] += 1
for (k, c), (v, n) in sorted(acc.items()):
    print(f"{k},{c},{v / n:.1f},{n}")
