import sys, time, numpy as np
sys.path.insert(0, ".")
from rust_robotics_amd.slam import fastslam1 as fs
from tests import helpers as H
import bench
print("pinned", bench.pin_to_gpu_numa_node(0))
for n, L in ((100, 8), (1000, 8)):
    lms = np.random.default_rng(3).uniform(-13, 13, size=(L, 2))
    zs = [np.ascontiguousarray(np.array(fs.get_observations(H.true_pose(t + 1, v=0.5), [tuple(p) for p in lms], seed=5, step=t)).reshape(-1, 3)) for t in range(64)]
    for res in (0.0, 5000.0):
        f = fs.FastSlam1(n, L, seed=5)
        if res: f.set_resident(res)
        for t in range(300):
            f.update([0.5, 0.1], zs[t % 64]); f.best_particle()
        t0 = time.perf_counter()
        for t in range(2000):
            f.update([0.5, 0.1], zs[t % 64]); f.best_particle()
        dt = (time.perf_counter() - t0) / 2000 * 1e6
        print(n, L, "resident" if res else "launched", "update + best_particle:", round(dt, 2), "us", f.resident_stats())
