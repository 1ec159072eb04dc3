import sys, json, os
sys.path.insert(0, "tools"); sys.path.insert(0, ".")
import resident_latency as R
import bench
print("pinned", bench.pin_to_gpu_numa_node(0))
for n, L in ((100, 3), (150, 5), (250, 5)):
    for res in (0.0, 5000.0):
        r = R.run(n, L, res)
        print(os.environ.get("RR_PF_SMALL_BLOCK"), n, L, res, r["mean_us"], r["p50_us"], r["min_us"], r["p99_us"], flush=True)
