"""Peer-to-peer transport (device-initiated exchange, no host code or collective library in a
step).  On the one GPU of the gpurun box it is exercised in both of its wirings: several shards
linked by pointer inside one process, and one process per shard mapping the peers through
hipIpc handles (both processes on device 0).  Either way every shard's particles must equal the
unsharded engine's bit for bit and no wait may time out."""
import math
import os
import subprocess
import sys

import numpy as np
import pytest

from tests import helpers as H

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def unsharded(n, steps, seed=42):
    import rust_robotics_amd.localization as loc
    from rust_robotics_amd import _ffi

    cfg = loc.MonteCarloLocalizationConfig(min_particles=n, max_particles=n, range_noise=0.5, velocity_noise=0.3,
                                           yaw_rate_noise=math.radians(5.0))
    ref = loc.MonteCarloLocalizer(cfg, seed=seed, resample_scheme=_ffi.RR_RESAMPLE_SYSTEMATIC)
    rng = np.random.default_rng(43)
    for t in range(steps):
        ref.step_async([1.0, 0.1], H.observations(H.REF_SCENE_LANDMARKS, H.true_pose(t + 1), 0.5, rng))
    return ref.get_particles_array()


def run_in_process(world, n_local, steps=10, mode="fused"):
    """mode: "fused" = the default step (own slots read lazily through their source index, only
    cross-rank slots move); "unfused" = eager gather of every slot; "mixed" = both alternating, with
    an accessor (which has to make the pending resample real) in the middle of the run"""
    from rust_robotics_amd.sharded import P2PShard

    shards = [P2PShard(g, world, 0, n_local, seed=42, range_noise=0.5, velocity_noise=0.3, yaw_rate_noise=math.radians(5.0))
              for g in range(world)]
    P2PShard.link_local(shards)
    rng = np.random.default_rng(43)
    for t in range(steps):
        obs = H.observations(H.REF_SCENE_LANDMARKS, H.true_pose(t + 1), 0.5, rng)
        fused = mode in ("fused", "wmax_mixed") or (mode == "mixed" and t % 3 != 2)
        for g, s in enumerate(shards):  # every shard's step is only enqueued; the device-side waits pair them up
            if mode == "wmax_mixed":  # who sends the shard's weight maximum -- the step kernel's last workgroup or the plan kernel's
                os.environ["RR_P2P_WMAX_EARLY"] = str((t + g) % 2)  # first -- changes from step to step and from rank to rank
            (s.step if fused else s.step_unfused)([1.0, 0.1], obs)
        if mode == "mixed" and t == steps // 2:
            for s in shards:
                s.particles()
    exp = unsharded(n_local * world, steps)
    for g, s in enumerate(shards):
        assert not s.timed_out(), f"rank {g}: a peer wait timed out"
        got = s.particles()
        assert np.array_equal(got.view(np.uint64), exp[g * n_local:(g + 1) * n_local].view(np.uint64)), f"rank {g} differs"
    for s in shards:
        s.close()
    print("P2P_LOCAL_OK")


def run_multinomial_in_process(world, n_local, steps=10, peaked=False):
    """MULTINOMIAL shards over the peer-to-peer transport (rr_pf_shard_step_p2p of shards created with RR_RESAMPLE_MULTINOMIAL, round 6) --
    the resampler the reference's ParticleFilterLocalizer / MonteCarloLocalizer really use (particle_filter.rs:441-473,
    monte_carlo_localization.rs:322-365, :387-392) -- against the unsharded multinomial filter: every draw searched by the shard whose
    CDF interval holds it, its source stored straight into the owner's slab.  peaked: the bench scene (a few heavy particles: nearly
    every slot of every rank is served by ONE rank)."""
    import rust_robotics_amd.localization as loc
    from rust_robotics_amd import _ffi
    from rust_robotics_amd.sharded import P2PShard

    n = n_local * world
    if peaked:
        kw = dict(seed=1, initial_state=[0.0, 0.0, 0.0, 1.0])
        lms, sigma = H.landmarks_grid(32, 1), 0.2
        cfg = loc.MonteCarloLocalizationConfig(min_particles=n, max_particles=n)
        ref = loc.MonteCarloLocalizer.with_initial_state([0.0, 0.0, 0.0, 1.0], cfg, seed=1, resample_scheme=_ffi.RR_RESAMPLE_MULTINOMIAL)
    else:
        kw = dict(seed=42, range_noise=0.5, velocity_noise=0.3, yaw_rate_noise=math.radians(5.0))
        lms, sigma = H.REF_SCENE_LANDMARKS, 0.5
        cfg = loc.MonteCarloLocalizationConfig(min_particles=n, max_particles=n, range_noise=0.5, velocity_noise=0.3, yaw_rate_noise=math.radians(5.0))
        ref = loc.MonteCarloLocalizer(cfg, seed=42, resample_scheme=_ffi.RR_RESAMPLE_MULTINOMIAL)
    shards = [P2PShard(g, world, 0, n_local, scheme=_ffi.RR_RESAMPLE_MULTINOMIAL, **kw) for g in range(world)]
    P2PShard.link_local(shards)
    rng = np.random.default_rng(43)
    for t in range(steps):
        obs = H.observations(lms, H.true_pose(t + 1), sigma, rng)
        for s in shards:
            s.step([1.0, 0.1], obs)
        ref.step_async([1.0, 0.1], obs)
        if t == steps // 2:  # an accessor in mid-run (the unsharded filter's resample is lazy: it has to make it real)
            for g, s in enumerate(shards):
                assert np.array_equal(s.particles().view(np.uint64), ref.get_particles_array()[g * n_local:(g + 1) * n_local].view(np.uint64)), f"mid-run: rank {g}"
    exp = ref.get_particles_array()
    for g, s in enumerate(shards):
        assert not s.timed_out(), f"rank {g}: a peer wait timed out"
        got = s.particles()
        bad = np.nonzero((got.view(np.uint64) != exp[g * n_local:(g + 1) * n_local].view(np.uint64)).any(axis=1))[0]
        assert bad.size == 0, f"rank {g}: {bad.size} of {n_local} particles differ from the unsharded multinomial filter, first {bad[:5]}"
    for s in shards:
        s.close()
    print("P2P_MN_OK")


@pytest.mark.parametrize("world,n_local,peaked", [(1, 5000, False), (2, 6000, False), (3, 4100, False), (4, 30_001, True), (2, 600_000, True), (8, 4100, False)])
def test_in_process_multinomial_shards_equal_unsharded(world, n_local, peaked):
    code = (f"import sys; sys.path.insert(0, {ROOT!r}); from tests.test_gpu_p2p import run_multinomial_in_process; "
            f"run_multinomial_in_process({world}, {n_local}, peaked={peaked})")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, PYTHONPATH=ROOT, GPU_MAX_HW_QUEUES="12"))
    assert r.returncode == 0 and "P2P_MN_OK" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])


def run_estimate_in_process(world, n_local, steps=12, mode="fused"):
    """rr_pf_shard_want_estimate: the shards' sums of the resampled set's fields divided by N, against the unsharded filter's
    in-step estimate (its deferred form: the same slot tiles at world size 1, hence the same bits there).  Read at once (the
    accessor's gather + k_est_slots add the sums up) and read a step later (the next step's k_step_lazy<kSrcWindow, EST> has,
    while it moved the particles)."""
    import rust_robotics_amd.localization as loc
    from rust_robotics_amd import _ffi
    from rust_robotics_amd.sharded import P2PShard

    os.environ["RR_PF_EST_DEFER"] = "1"
    kw = dict(range_noise=0.5, velocity_noise=0.3, yaw_rate_noise=math.radians(5.0))
    shards = [P2PShard(g, world, 0, n_local, seed=42, **kw) for g in range(world)]
    P2PShard.link_local(shards)
    cfg = loc.MonteCarloLocalizationConfig(min_particles=n_local * world, max_particles=n_local * world, **kw)
    ref = loc.MonteCarloLocalizer(cfg, seed=42, resample_scheme=_ffi.RR_RESAMPLE_SYSTEMATIC)
    rng = np.random.default_rng(43)

    def step_all(obs, want):
        # (no host wait here: the unsharded reference filter steps next, in ITS stream, beside the shards -- the form in which round 4
        # saw a ~1 % disagreement in 1 of ~60 runs.  Run down in round 5: not the estimate but the sharded step itself, a payload
        # scratch array in device memory written by two workgroups of k_shard_plan_mark (p2p_core.hpp, p2p_exchange);
        # tools/soak_shard_estimate.py, test_sharded_steps_beside_another_filter below)
        for s in shards:
            s.want_estimate(want)
            (s.step if mode == "fused" else s.step_unfused)([1.0, 0.1], obs)

    def mean_of_shards():
        sums = [s.estimate_sums() for s in shards]
        assert all(den == n_local * world for _, den in sums)
        return np.sum([a for a, _ in sums], axis=0) / sums[0][1]

    def check(got, want, what):
        if world == 1:
            assert np.array_equal(got.view(np.uint64), want.view(np.uint64)), (what, got, want)
        np.testing.assert_allclose(got, want, rtol=1e-11, atol=1e-11, err_msg=what)

    t = 0
    while t < steps:
        obs = H.observations(H.REF_SCENE_LANDMARKS, H.true_pose(t + 1), 0.5, rng)
        step_all(obs, True)
        ref.step_async_estimate([1.0, 0.1], obs)
        want = np.array(ref.last_step_estimate())
        t += 1
        if t % 3 == 0:  # one more step on both sides -- a plain one -- before the value of the step before is read
            obs2 = H.observations(H.REF_SCENE_LANDMARKS, H.true_pose(t + 1), 0.5, rng)
            step_all(obs2, False)
            ref.step_async([1.0, 0.1], obs2)
            t += 1
            check(mean_of_shards(), want, f"step {t - 1}, read a step later")
        else:
            check(mean_of_shards(), want, f"step {t}, read at once")
    exp = ref.get_particles_array()
    for g, s in enumerate(shards):
        assert not s.timed_out()
        assert np.array_equal(s.particles().view(np.uint64), exp[g * n_local:(g + 1) * n_local].view(np.uint64)), f"rank {g} differs"
        s.close()
    print("P2P_EST_OK")


@pytest.mark.parametrize("world,n_local,mode", [(1, 5000, "fused"), (2, 6000, "fused"), (3, 4100, "fused"), (2, 6000, "unfused"), (1, 300_000, "fused")])
def test_in_process_shards_leave_their_part_of_the_mean(world, n_local, mode):
    code = f"import sys; sys.path.insert(0, {ROOT!r}); from tests.test_gpu_p2p import run_estimate_in_process; run_estimate_in_process({world}, {n_local}, mode={mode!r})"
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, PYTHONPATH=ROOT, GPU_MAX_HW_QUEUES="8"))
    assert r.returncode == 0 and "P2P_EST_OK" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])


@pytest.mark.parametrize("world,n_local,mode", [(1, 5000, "fused"), (2, 6000, "fused"), (3, 4100, "fused"),
                                                (2, 6000, "unfused"), (3, 4100, "mixed"), (2, 700_000, "fused")])
def test_in_process_shards_equal_unsharded(world, n_local, mode):
    """Several shards of one process on ONE GPU only make progress together if each shard's stream
    has its own hardware queue (a spinning wait kernel would otherwise block the peer kernel queued
    behind it), so the case runs in a fresh interpreter with a generous queue count.  One process
    per GPU -- the deployment shape -- has no such constraint."""
    code = f"import sys; sys.path.insert(0, {ROOT!r}); from tests.test_gpu_p2p import run_in_process; run_in_process({world}, {n_local}, mode={mode!r})"
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, PYTHONPATH=ROOT, GPU_MAX_HW_QUEUES="8"))
    assert r.returncode == 0 and "P2P_LOCAL_OK" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])


@pytest.mark.parametrize("world,n_local,mode,early", [(2, 6000, "fused", "1"), (3, 4100, "mixed", "1"), (1, 300_000, "fused", "1"),
                                                      (3, 4100, "wmax_mixed", "0"), (2, 700_000, "wmax_mixed", "0")])
def test_in_process_shards_either_sender_of_the_weight_maximum(world, n_local, mode, early):
    """The one-launch plan of a shard needs the ranks' weight maxima first.  Default: the plan kernel's first workgroup runs an
    exchange and raises a flag for the others.  RR_P2P_WMAX_EARLY=1 (round 5; no gain at world size 1, kept for a real fabric):
    the LAST workgroup of every rank's step kernel sends its shard's maximum as it finishes and the plan kernel's workgroups take
    the records from their own mailbox.  Both, and any mixture over steps and ranks, give the unsharded filter's bits."""
    code = f"import sys; sys.path.insert(0, {ROOT!r}); from tests.test_gpu_p2p import run_in_process; run_in_process({world}, {n_local}, mode={mode!r})"
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, PYTHONPATH=ROOT, GPU_MAX_HW_QUEUES="8", RR_P2P_WMAX_EARLY=early))
    assert r.returncode == 0 and "P2P_LOCAL_OK" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])


@pytest.mark.parametrize("world,n_local,mode", [(2, 6000, "fused"), (3, 4100, "mixed")])
def test_in_process_shards_multi_launch_plan(world, n_local, mode):
    """The same with RR_PF_FUSED_PLAN=0: the plan of a shard as separate launches (WMAX exchange | k_quantize_reduce |
    k_scan_exchange | k_mark in the window layout) and the overhang delivered by k_push_window -- the form shards beyond 2^20
    particles take."""
    code = f"import sys; sys.path.insert(0, {ROOT!r}); from tests.test_gpu_p2p import run_in_process; run_in_process({world}, {n_local}, mode={mode!r})"
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, PYTHONPATH=ROOT, GPU_MAX_HW_QUEUES="8", RR_PF_FUSED_PLAN="0"))
    assert r.returncode == 0 and "P2P_LOCAL_OK" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])


def test_shard_estimate_says_which_step_it_belongs_to():
    """rr_pf_shard_last_estimate_sums returns the sums of the LAST STEP TAKEN WITH rr_pf_shard_want_estimate ON, validated by a stamp
    the summing kernel leaves behind the sums -- not by the live Ctl.fired, which belongs to the latest step.  (a) a step
    whose resample fired, then a plain step whose gate stays shut (no observation: the weights stay uniform, N_eff = N): the first
    step's sums are still returned, and equal the unsharded filter's estimate of that step bit for bit; (b) a step whose own gate
    stays shut: an error that says so, whether it is read at once or a step later."""
    import rust_robotics_amd.localization as loc
    from rust_robotics_amd import _ffi
    from rust_robotics_amd.sharded import P2PShard

    os.environ["RR_PF_EST_DEFER"] = "1"
    n, kw = 6000, dict(range_noise=0.5, velocity_noise=0.3, yaw_rate_noise=math.radians(5.0))
    rng = np.random.default_rng(3)
    obs = H.observations(H.REF_SCENE_LANDMARKS, H.true_pose(1), 0.5, rng)
    none = np.zeros((0, 3))
    for read_later in (False, True):
        s = P2PShard(0, 1, 0, n, seed=9, gate=_ffi.RR_GATE_NEFF, resample_threshold=1.0, **kw)
        P2PShard.link_local([s])
        ref = loc.ParticleFilterLocalizer(loc.ParticleFilterConfig(n_particles=n, resample_threshold=1.0, **kw), seed=9,
                                          resample_scheme=_ffi.RR_RESAMPLE_SYSTEMATIC)
        # (a)
        s.want_estimate(True)
        s.step([1.0, 0.1], obs)
        ref.step_async_estimate([1.0, 0.1], obs)
        want = np.array(ref.last_step_estimate())
        s.want_estimate(False)
        s.step([1.0, 0.1], none)  # gate shut: uniform weights
        s.synchronize()
        sums, den = s.estimate_sums()
        assert den == n and np.array_equal((sums / den).view(np.uint64), want.view(np.uint64)), (sums / den, want)
        # (b)
        s.want_estimate(True)
        s.step([1.0, 0.1], none)
        if read_later:
            s.want_estimate(False)
            s.step([1.0, 0.1], obs)
        with pytest.raises(Exception) as ei:
            s.estimate_sums()
        assert "stayed shut" in str(ei.value)
        assert not s.timed_out()
        s.close()


@pytest.mark.parametrize("plain", ["0", "1"])
def test_sharded_steps_beside_another_filter(plain):
    """tools/soak_shard_estimate.py as a regression: a world-size-1 shard of 5 000 particles and an unsharded filter step side by
    side on the device (two streams, no host wait between them) -- 400 trajectories of 12 steps each, with the per-step mean
    (plain = 0) and as bare steps two at a time (plain = 1).  Before the round-5 fix ~1 in 150 trajectories diverged (the shard's
    total T came back as the bit pattern of w_max)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "soak_shard_estimate.py"), "400", "5000"], capture_output=True, text=True,
                       timeout=600, env=dict(os.environ, PYTHONPATH=ROOT, GPU_MAX_HW_QUEUES="8", RR_SOAK_PLAIN=plain, RR_SOAK_BUDGET_S="120"))
    assert r.returncode == 0 and '"mismatches": 0' in r.stdout, (r.stdout[-3000:], r.stderr[-2000:])


def test_two_processes_over_ipc_handles():
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29721", os.path.join(ROOT, "tests", "_gpu_p2p_worker.py"), "8000", "8"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=dict(os.environ, PYTHONPATH=ROOT))
    assert r.returncode == 0 and r.stdout.count("P2P_OK") == 2, (r.stdout[-2000:], r.stderr[-4000:])


def test_bench_two_ranks_from_a_bare_shell():
    """`python bench.py --gpus 2` end to end on a one-GPU box (RR_BENCH_SHARE_DEVICE=1: both ranks on device 0): the
    self-launch under torch.distributed.run, the gloo group, the transport ladder (RCCL refuses two ranks on one device,
    so the peer-to-peer transport is validated against the unsharded filter of all particles ACROSS the two processes),
    the timed region, and rank 0's line -- the only thing on stdout, whatever gloo and RCCL print."""
    import json

    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(PYTHONPATH=ROOT, RR_BENCH_SHARE_DEVICE="1", RR_BENCH_DEADLINE_S="240")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5", "--particles", "100000",
                        "--no-extra-legs", "--no-cpu-baseline"], capture_output=True, text=True, timeout=400, env=env)
    assert r.returncode == 0, r.stderr[-4000:]
    lines = r.stdout.splitlines()
    # stdout: the legs in full, one JSON line each ({"leg": ...}), and LAST the compact line the driver parses (< 4 KB)
    assert all(ln.startswith("{") for ln in lines) and len(lines[-1]) < 4096, r.stdout[:2000]
    d = json.loads(lines[-1])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0 and "deadline_exceeded" not in d
    legs = {q["leg"]: q for q in map(json.loads, lines[:-1])}
    full = legs["headline"]
    assert "peer-to-peer transport validated bit-identical" in full["config"]["sharding"], full["config"]["sharding"]
    assert legs["sharded"]["transport"].startswith("p2p") and not legs["sharded"]["p2p_timed_out"]
    assert "[bench rank 0" in r.stderr and "[bench rank 1" in r.stderr


@pytest.mark.parametrize("spec", ["2,100000,40", "2,1000000,40", "3,60000,10"])
def test_shards_sharing_a_device_survive_a_resample_that_moves_most_of_a_shard(spec):
    """The round-3 "give-up at 10^6 particles per rank" of ranks sharing ONE device, run down in round 4:
    observations that jump back in time collapse the weights onto a few particles of one shard, the next resample moves most of
    the other shard across the boundary, and the consuming step kernels of the sharers -- which wait inside the kernel for those
    deliveries -- held every workgroup slot of the device before the delivering push kernel had been dispatched.  Sharers that
    can fill the device now take the eager step (rr_pf_shard_step_p2p); smaller ones keep the lazy step, whose wait polls the
    seal only.  Every shard must equal its block of the unsharded filter and no wait may time out
    (tools/p2p_shared_device_jump.py)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "p2p_shared_device_jump.py"), spec], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, PYTHONPATH=ROOT, GPU_MAX_HW_QUEUES="8", RR_P2P_TIMEOUT_MS="2000"))
    lines = [ln for ln in r.stdout.splitlines() if "timed_out=" in ln]
    assert r.returncode == 0 and len(lines) == int(spec.split(",")[0]), (r.stdout[-1500:], r.stderr[-2000:])
    assert all("timed_out=False equal=True" in ln for ln in lines), lines
