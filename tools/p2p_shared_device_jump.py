#!/usr/bin/env python3
"""Shards of one filter on ONE device, fed observations that jump back in time after 60 steps: the weights collapse onto a few
particles of one shard and the next resample moves most of the other shard across the boundary (~10^6 deliveries at 10^6
particles per rank).  This is the scenario behind the round-3 "give-up at 10^6 particles per rank" of two ranks sharing a device
(bench.py replays warm-up observations): the consuming kernels of the sharers filled the device and waited for a push kernel that
had not been dispatched yet.  Since round 4 such sharers take the eager step (rr_pf_shard_step_p2p).  Every shard must equal its
block of the unsharded filter bit for bit and no wait may time out.
    GPU_MAX_HW_QUEUES=8 python tools/p2p_shared_device_jump.py 2,1000000,40 2,800000,40      (world, particles per rank, jump)"""
import math, sys, os
import numpy as np
sys.path.insert(0, os.getcwd())
from tests import helpers as H
import rust_robotics_amd.localization as loc
from rust_robotics_amd import _ffi
from rust_robotics_amd.sharded import P2PShard

def run(world, n_local, sigma, jump_back, unfused=False):
    lms = H.landmarks_grid(32, 1)
    shards = [P2PShard(g, world, 0, n_local, seed=42, range_noise=sigma, velocity_noise=2.0, yaw_rate_noise=math.radians(40.0)) for g in range(world)]
    P2PShard.link_local(shards)
    cfg = loc.MonteCarloLocalizationConfig(min_particles=n_local * world, max_particles=n_local * world, range_noise=sigma, velocity_noise=2.0, yaw_rate_noise=math.radians(40.0))
    ref = loc.MonteCarloLocalizer(cfg, seed=42, resample_scheme=_ffi.RR_RESAMPLE_SYSTEMATIC)
    rng = np.random.default_rng(43)
    obs = [H.observations(lms, H.true_pose(t + 1), sigma, rng) for t in range(80)]
    order = list(range(60)) + list(range(60 - jump_back, 60)) * 2
    for k, t in enumerate(order):
        for s in shards:
            (s.step_unfused if unfused else s.step)([1.0, 0.1], obs[t])
        ref.step_async([1.0, 0.1], obs[t])
    exp = ref.get_particles_array()
    bad = False
    for g, s in enumerate(shards):
        to = s.timed_out()
        try:
            got = s.particles()
            eq = np.array_equal(got.view(np.uint64), exp[g * n_local:(g + 1) * n_local].view(np.uint64))
        except Exception as e:
            eq = f"ERR {e}"
        print(f"world {world} n_local {n_local} sigma {sigma} jump {jump_back} unfused {unfused}: rank {g} timed_out={to} equal={eq} migrated?", flush=True)
    for s in shards:
        s.close()

import sys
args = [a for a in sys.argv[1:]]
for spec in args:
    parts = spec.split(',')
    run(int(parts[0]), int(parts[1]), 0.2, int(parts[2]), unfused=len(parts) > 3)
