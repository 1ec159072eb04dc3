"""debug: merge the kernel traces of the ranks of one run (rocprofv3 --kernel-trace, one csv per process) into one timeline"""
import csv, glob, sys
sys.path.insert(0, 'tools')
from summarize_rocprof import short
rows = []
for k, path in enumerate(sorted(glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True))):
    for r in csv.DictReader(open(path)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), k, short(r['Kernel_Name'])))
rows.sort()
names = [r for r in rows if r[3].startswith('k_shard_plan_mark')]
if not names:
    print('no plan kernels', len(rows)); sys.exit(0)
mid = names[len(names) // 2][0]
t0 = None
n = 0
for s, e, k, name in rows:
    if s < mid: continue
    if t0 is None: t0 = s
    print(f"{(s - t0) / 1e3:9.2f} {(e - t0) / 1e3:9.2f} {(e - s) / 1e3:8.2f}  p{k} {name}")
    n += 1
    if n > int(sys.argv[2]): break
