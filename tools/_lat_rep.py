import sys, json
sys.path.insert(0, "tools"); sys.path.insert(0, ".")
import resident_latency as R
for rep in range(4):
    for n, L in ((100, 3), (1000, 4)):
        r = R.run(n, L, 5000.0)
        print(rep, n, L, r["mean_us"], r["p50_us"], r["min_us"], r["p99_us"], flush=True)
