"""One step in several hundred synchronous steps takes ~0.45 ms instead of ~66 us: a one-off stall of the HIP runtime at some launch count
of the process (its position moves with the launches before it -- `cold`: another filter created, stepped and destroyed first).  With
bench.py --steps 20 that single step is 22 us of the synchronous leg's mean when it falls into the window (round 5: 85-93 us instead of
65-67), which is why that leg reports median and maximum beside the mean.   python tools/stall_probe.py [cold]"""
import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
import bench
import rust_robotics_amd.localization as loc
n, L = 1_000_000, 32
obs = bench.make_scene(L, 3000, seed=1)
u = [1.0, 0.1]
cfg = loc.MonteCarloLocalizationConfig(min_particles=n, max_particles=n)
mk = lambda: loc.MonteCarloLocalizer.with_initial_state([0.0, 0.0, 0.0, 1.0], cfg, seed=1, device=0, resample_scheme=1, likelihood_mode=0)
pf = mk()
if len(sys.argv) > 1 and sys.argv[1] == "cold":
    pc = mk()
    for t in range(25): pc.step_async_estimate(u, obs[t])
    pc.synchronize()
    del pc
for t in range(1000):
    pf.step_async_estimate(u, obs[t])
    if t % 50 == 49: pf.synchronize()
ts = []
for t in range(600):
    a = time.perf_counter(); pf.step(u, obs[1000 + t]); ts.append((time.perf_counter() - a) * 1e6)
ts = np.array(ts)
slow = [(i, round(v)) for i, v in enumerate(ts) if v > 150]
print(sys.argv[1:], "median", round(float(np.median(ts)), 1), "mean first 25", round(float(ts[:25].mean()), 1), "slow steps (index, us):", slow[:20])
