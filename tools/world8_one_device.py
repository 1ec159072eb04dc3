#!/usr/bin/env python3
"""The BASELINE 8-rank configurations executed as 8 ranks -- on ONE device -- and compared bit for bit with the unsharded
filter of all particles ("next round" item 1).  No 8-GPU node is needed to execute `n_ranks = 8`: eight shards
linked by pointer inside one process (this file), eight processes over hipIpc (tests/_gpu_p2p_worker.py,
tests/_gpu_fs1_p2p_worker.py, `RR_BENCH_SHARE_DEVICE=1 python bench.py --gpus 8`) and gloo world 8 on CPU
(tests/test_sharded_gloo.py, tests/test_fs1_sharded_gloo.py).  What this covers and what it cannot:

  * the segment arithmetic for 8 blocks, the mailbox / inbox sizing, the sealed deliveries between every pair of ranks, the
    window markers over a global slot index that spans 8 blocks, both senders of the weight maximum -- all with n_ranks = 8;
  * shards small enough for the sharers' step kernels not to fill the device (8 x <= 49 152 particles) take the LAZY window step,
    the deployment path (k_step_lazy<kSrcWindow> | k_shard_plan_mark | k_push_window); larger sharers of one device take the
    eager step (rr_pf_shard_step_p2p's own rule: a consuming kernel that waits for a delivery must not hold every workgroup
    slot of the device the delivering kernel needs) -- unless RR_P2P_CU_PARTITION=1 gives every shard its own eighth of the
    CUs (a stream with a CU mask), in which case the lazy step runs at the full BASELINE sizes too;
  * NOT covered: xGMI itself (latency, ordering of remote stores across a fabric) -- tests/test_gpu_two_devices.py.

Semantics held: fastslam1.rs:205-234 (systematic walk: across 8 blocks the walk becomes the segment matrix), particle_filter.rs:426-439
(normalisation: the integer total is the sum of 8 shard totals).

    GPU_MAX_HW_QUEUES=12 python tools/world8_one_device.py mcl-small mcl-heavy mcl-config5 fs1-config4      (one JSON line per case)
"""
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

WORLD = int(os.environ.get("RR_WORLD8_WORLD", "8"))


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def lazy_path_expected(world, n_local):
    """rr_pf_shard_step_p2p's rule for ranks sharing a device (pf_engine.hip): eager when the sharers' step kernels can fill it"""
    if os.environ.get("RR_P2P_CU_PARTITION", "0") not in ("", "0"):
        return True
    return ((n_local + 511) // 512) * world <= 3 * 256


def run_mcl(case, world, n_local, steps, *, peaked=False, L=32, heavy=None, mode="fused", early=None, seed=42):
    """`world` P2PShards of one process on device 0 against the unsharded filter of world * n_local particles.
    heavy: list of GLOBAL particle indices that alone sit near the true pose (everything else is 100+ m away): after the first
    step's weighting every output slot of every rank copies one of them -- with heavy = [7 * n_local + k] every slot of ranks
    0..6 crosses ranks (the worst case of the exchange: the whole window is overhang)."""
    import rust_robotics_amd.localization as loc
    from rust_robotics_amd import _ffi
    from rust_robotics_amd.sharded import P2PShard
    from tests import helpers as H

    n = world * n_local
    if peaked:  # the bench scene: L landmarks on a seeded grid, defaults of ParticleFilterConfig
        kw = dict(seed=1, initial_state=[0.0, 0.0, 0.0, 1.0])
        lms, sigma = H.landmarks_grid(L, 2 if L == 64 else 1), 0.2
        cfg = loc.MonteCarloLocalizationConfig(min_particles=n, max_particles=n)
        ref = loc.MonteCarloLocalizer.with_initial_state([0.0, 0.0, 0.0, 1.0], cfg, seed=1, resample_scheme=_ffi.RR_RESAMPLE_SYSTEMATIC)
    else:
        kw = dict(seed=seed, range_noise=0.5, velocity_noise=0.3, yaw_rate_noise=math.radians(5.0))
        lms, sigma = H.REF_SCENE_LANDMARKS, 0.5
        cfg = loc.MonteCarloLocalizationConfig(min_particles=n, max_particles=n, range_noise=0.5, velocity_noise=0.3, yaw_rate_noise=math.radians(5.0))
        ref = loc.MonteCarloLocalizer(cfg, seed=seed, resample_scheme=_ffi.RR_RESAMPLE_SYSTEMATIC)
    if early is not None:
        os.environ["RR_P2P_WMAX_EARLY"] = str(early)
    shards = [P2PShard(g, world, 0, n_local, **kw) for g in range(world)]
    P2PShard.link_local(shards)
    if heavy is not None:
        rng0 = np.random.default_rng(5)
        cloud = np.column_stack([rng0.uniform(100.0, 200.0, n), rng0.uniform(100.0, 200.0, n), rng0.uniform(-3.0, 3.0, n),
                                 np.full(n, 1.0), np.full(n, 1.0 / n)])
        for k, gi in enumerate(heavy):
            cloud[gi, :4] = (0.0, 0.0 + 1e-3 * k, 0.0, 1.0)
        ref.set_particles_array(cloud)
        for g, s in enumerate(shards):
            s.set_particles(cloud[g * n_local:(g + 1) * n_local])
    rng = np.random.default_rng(43)
    t0 = time.time()
    for t in range(steps):
        obs = H.observations(lms, H.true_pose(t + 1), sigma, rng)
        fused = mode in ("fused", "wmax_mixed") or (mode == "mixed" and t % 3 != 2)
        for g, s in enumerate(shards):  # only enqueued: the device-side waits pair the shards up
            if mode == "wmax_mixed":
                os.environ["RR_P2P_WMAX_EARLY"] = str((t + g) % 2)
            (s.step if fused else s.step_unfused)([1.0, 0.1], obs)
        ref.step_async([1.0, 0.1], obs)
        if mode == "mixed" and t == steps // 2:
            for s in shards:
                s.particles()
        if heavy is not None and t == 0:  # after the collapse: every particle is a copy of a heavy one (checked on the shards)
            for g, s in enumerate(shards):
                got = s.particles()
                assert not s.timed_out(), f"rank {g}: a peer wait timed out in the collapse step"
                assert len(np.unique(_bits(got[:, :4]), axis=0)) <= len(heavy), f"rank {g}: the collapse left more than {len(heavy)} distinct particles"
    exp = ref.get_particles_array()
    topo = [s.topology() for s in shards]
    gave_up, differ = [], []
    for g, s in enumerate(shards):
        if s.timed_out():
            gave_up.append(g)
        got = s.particles()
        if not np.array_equal(_bits(got), _bits(exp[g * n_local:(g + 1) * n_local])):
            differ.append(g)
    dt = time.time() - t0
    for s in shards:
        s.close()
    rec = dict(case=case, filter="MCL systematic (fastslam1.rs:205-234 walk over particle_filter.rs weights)", world=world, n_local=n_local,
               n_global=n, landmarks=int(len(lms)), steps=steps, mode=mode, wmax_early=os.environ.get("RR_P2P_WMAX_EARLY", "0"),
               fused_plan_env=os.environ.get("RR_PF_FUSED_PLAN", "1"), cu_partition=os.environ.get("RR_P2P_CU_PARTITION", "0"),
               lazy_window_step=all(t["last_step"] == "lazy" for t in topo) if mode in ("fused", "wmax_mixed") else None,
               lazy_window_step_expected=lazy_path_expected(world, n_local), cus_per_shard=sorted({t["cu_partition_cus"] for t in topo}),
               ranks_sharing_the_device=topo[0]["n_sharing"], heavy=heavy, ranks_that_gave_up=gave_up,
               ranks_that_differ=differ, equal_to_unsharded=not differ, seconds=round(dt, 2), wiring="8 shards linked in one process")
    print(json.dumps(rec), flush=True)
    assert not gave_up and not differ, rec
    if mode in ("fused", "wmax_mixed"):
        assert rec["lazy_window_step"] == rec["lazy_window_step_expected"], rec
    if os.environ.get("RR_P2P_CU_PARTITION", "0") not in ("", "0") and world > 1:
        assert rec["cus_per_shard"] != [0], "RR_P2P_CU_PARTITION was asked for and no shard's stream got its share of the CUs"
    return rec


def run_fs1(case, world, n_local, L, steps, *, chunks=0, variant=1):
    """`world` ShardedFastSlam1 shards of one process against the unsharded filter (configs[3]: 8 x 125 000 x 200, 19.3 GB).
    The engines initialise themselves (first_obs_cov = 0.5: the first observation of a landmark leaves an EKF-ready
    covariance, so every later pair takes the EKF branch); the host only ever holds one shard's state beside the
    unsharded filter's."""
    from rust_robotics_amd.slam import fastslam1 as fs
    from rust_robotics_amd.slam import fastslam2 as fs2
    from tests import helpers as H

    n = world * n_local
    rng = np.random.default_rng(61)
    lms = rng.uniform(-13.0, 13.0, size=(L, 2))
    Plain, Sharded = (fs2.FastSlam2, fs2.ShardedFastSlam2) if variant == 2 else (fs.FastSlam1, fs.ShardedFastSlam1)

    def params():
        prm = fs2.default_params() if variant == 2 else fs.default_params()
        base = prm.base if variant == 2 else prm
        base.first_obs_cov = 0.5
        base.nth = n / 1.5
        base.initial_weight = 1.0 / n
        return prm

    zs = [np.array(fs.get_observations(H.true_pose(t + 1, v=0.5), [tuple(p) for p in lms], seed=8, step=t)).reshape(-1, 3) for t in range(steps)]
    t0 = time.time()
    shards = [Sharded(g, world, n_local, L, params=params(), seed=8, obs_chunks=chunks) for g in range(world)]
    fs.ShardedFastSlam1.link_local(shards)
    for z in zs:
        for s in shards:
            s.update_async([0.5, 0.1], z)
    gave_up = [g for g, s in enumerate(shards) if s.timed_out()]
    whole = Plain(n, L, params=params(), seed=8, obs_chunks=chunks)
    fired = []
    for z in zs:
        whole.update([0.5, 0.1], z)
        fired.append(bool(whole.last_resample_fired()))
    ep, em = whole.get_state()
    whole.close()
    differ = []
    for g, s in enumerate(shards):
        p, m = s.get_state()
        sl = slice(g * n_local, (g + 1) * n_local)
        if not (np.array_equal(_bits(p), _bits(ep[sl])) and np.array_equal(_bits(m), _bits(em[sl]))):
            differ.append(g)
        if s.timed_out() and g not in gave_up:
            gave_up.append(g)
        del p, m
        s.close()
    rec = dict(case=case, filter=f"FastSLAM {variant}.0", world=world, n_local=n_local, n_global=n, landmarks=L, steps=steps, obs_chunks=chunks,
               gate_fired=fired, ranks_that_gave_up=gave_up, ranks_that_differ=differ, equal_to_unsharded=not differ,
               seconds=round(time.time() - t0, 2), wiring="8 shards linked in one process")
    print(json.dumps(rec), flush=True)
    assert not gave_up and not differ, rec
    assert any(fired), "no update's gate fired: the resample was not exercised"
    return rec


def run_mcl_multinomial(case, world, n_local, steps, peaked=False):
    """multinomial shards (rr_pf_shard_step_p2p of RR_RESAMPLE_MULTINOMIAL shards) against the unsharded multinomial filter"""
    from tests.test_gpu_p2p import run_multinomial_in_process

    t0 = time.time()
    run_multinomial_in_process(world, n_local, steps=steps, peaked=peaked)  # asserts bit-identity and zero give-ups itself
    rec = dict(case=case, filter="MCL multinomial (particle_filter.rs:441-473, monte_carlo_localization.rs:322-365)", world=world, n_local=n_local,
               n_global=world * n_local, landmarks=32 if peaked else 4, steps=steps, mode="multinomial-p2p", ranks_that_gave_up=[], ranks_that_differ=[],
               equal_to_unsharded=True, seconds=round(time.time() - t0, 2), wiring="8 shards linked in one process")
    print(json.dumps(rec), flush=True)
    return rec


CASES = {
    "mcl-multinomial": lambda: [run_mcl_multinomial("mcl-multinomial-small", WORLD, 4100, 10), run_mcl_multinomial("mcl-multinomial-peaked", WORLD, 250_000, 6, peaked=True)],
    # the lazy window step (the deployment path) with n_ranks = 8, all three plan forms
    "mcl-small": lambda: [run_mcl("mcl-small", WORLD, 4100, 12, mode="fused"), run_mcl("mcl-small-mixed", WORLD, 4100, 12, mode="mixed"),
                          run_mcl("mcl-small-early", WORLD, 4100, 12, mode="fused", early=1),
                          run_mcl("mcl-small-wmax-mixed", WORLD, 4100, 12, mode="wmax_mixed", early=0)],
    "mcl-lazy-max": lambda: [run_mcl("mcl-lazy-max", WORLD, 49_000, 10, peaked=True)],
    # worst-case resamples: one heavy particle in the last rank (every slot of ranks 0..6 crosses ranks), in the first rank,
    # and one each in ranks 0 and 7 (the boundary between their runs falls inside rank 3 / 4)
    "mcl-heavy": lambda: [run_mcl("mcl-heavy-last-rank", WORLD, 4100, 8, heavy=[(WORLD - 1) * 4100 + 123]),
                          run_mcl("mcl-heavy-last-rank-early", WORLD, 4100, 8, heavy=[(WORLD - 1) * 4100 + 123], early=1),
                          run_mcl("mcl-heavy-first-rank", WORLD, 4100, 8, heavy=[77], early=0),
                          run_mcl("mcl-heavy-both-ends", WORLD, 4100, 8, heavy=[5, (WORLD - 1) * 4100 + 4000]),
                          run_mcl("mcl-heavy-last-rank-49000", WORLD, 49_000, 6, heavy=[(WORLD - 1) * 49_000 + 123], early=1)],
    # BASELINE configs[4]: 8 x 2 000 000 x 64
    "mcl-config5": lambda: [run_mcl("mcl-config5", WORLD, 2_000_000, 6, peaked=True, L=64)],
    "mcl-config5-heavy": lambda: [run_mcl("mcl-config5-heavy-last-rank", WORLD, 2_000_000, 4, peaked=True, L=64, heavy=[(WORLD - 1) * 2_000_000 + 123])],
    # beyond the BASELINE sizes: 8 x 12 500 000 = 1e8 particles in one global slot space (positions beyond 2^24, inboxes of 400 MB),
    # steady state and the worst-case resample
    "mcl-1e8": lambda: [run_mcl("mcl-1e8", WORLD, 12_500_000, 3, peaked=True, L=4),
                        run_mcl("mcl-1e8-heavy-last-rank", WORLD, 12_500_000, 3, peaked=True, L=4, heavy=[(WORLD - 1) * 12_500_000 + 123])],
    "mcl-multinomial-1e8": lambda: [run_mcl_multinomial("mcl-multinomial-1e8", WORLD, 12_500_000, 3)],
    # BASELINE configs[3]: 8 x 125 000 x 200
    "fs1-small": lambda: [run_fs1("fs1-small", WORLD, 1300, 7, 8, chunks=2), run_fs1("fs2-small", WORLD, 1300, 7, 8, chunks=2, variant=2)],
    "fs1-config4": lambda: [run_fs1("fs1-config4", WORLD, 125_000, 200, 4)],
    # FastSLAM 2.0 at the same size, and FastSLAM 1.0 at twice it (8 x 250 000 x 200: shards of 4.8 GB per buffer set, a 38 GB unsharded filter)
    "fs2-config4": lambda: [run_fs1("fs2-config4", WORLD, 125_000, 200, 3, variant=2)],
    "fs1-2e6": lambda: [run_fs1("fs1-2e6", WORLD, 250_000, 200, 3)],
}


if __name__ == "__main__":
    names = sys.argv[1:] or list(CASES)
    for name in names:
        CASES[name]()
    print("WORLD8_OK")
