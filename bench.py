#!/usr/bin/env python3
"""bench.py -- particle-landmark updates/s of the MCL hot path on N MI355X.

Contract (driver): ``python bench.py --gpus N --steps K --warmup W``; for N > 1 it is launched
under torch.distributed.run with one rank per GPU.  One "step" = one full pass of the hot path
(propagate + weight + resample) over the whole particle set; the workload at N = 1 is
BASELINE.json configs[1]: fixed-N MCL, 1 000 000 particles x 32 landmarks, resampling every step
(weak scaling: 1e6 particles per GPU).  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12  # B/s, MI355X_MICROARCH.md "HBM3E peak BW 8.0 TB/s spec"
# algorithmic HBM bytes per particle of the dominant kernel: read x,y,yaw (24 B) + write
# x,y,yaw,v,w (40 B); the systematic path's k_step_lazy also reads and clears the 4-byte resample marker
K1_BYTES = {"systematic": 72.0, "multinomial": 64.0}


def measured_traffic(kernel_prefix, workload):
    """HBM bytes per launch of `kernel_prefix` from the committed PMC passes (separate
    rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs, read side doubled as MI355X_MICROARCH.md
    prescribes for gfx950): profiles/r01c_pmc_hbm_traffic.csv.  None if the file is absent."""
    import csv

    for name, wl in (("r01f_pmc_hbm_traffic_fastslam_timed_region.csv", workload + "_timed_region"), ("r01c_pmc_hbm_traffic.csv", workload)):
        try:
            for r in csv.DictReader(open(os.path.join(ROOT, "profiles", name))):
                if r["workload"] == wl and r["kernel"].startswith(kernel_prefix):
                    return (float(r["read_MB_corrected_x2"]) + float(r["write_MB"])) * 1e6
        except Exception:
            pass
    return None


def make_scene(L, steps, seed):
    from tests import helpers as H

    lms = H.landmarks_grid(L, seed)
    rng = np.random.default_rng(seed + 1)
    return [H.observations(lms, H.true_pose(t + 1), 0.2, rng) for t in range(steps)]


def cpu_baseline(n, L, obs_list, max_seconds=25.0):
    """The literal C restatement of the reference (oracle/ref_literal.c), one host core,
    same workload; runs whole steps until ~max_seconds have elapsed."""
    import oracle
    from oracle import dp, u32p

    ref = oracle.ref()
    det = oracle.det()
    x, y, yaw, v = (np.zeros(n) for _ in range(4))
    st = np.array([0.0, 0.0, 0.0, 1.0])
    det.det_pf_init(n, 1, 0, dp(st), dp(x), dp(y), dp(yaw), dp(v))
    w = np.full(n, 1.0 / n)
    idx = np.empty(n, np.uint32)
    est = np.empty(4)
    sv, sw = 2.0, math.radians(40.0)
    z0, z1, r, r2 = (np.empty(n) for _ in range(4))
    steps = 0
    t_total = 0.0
    while steps < len(obs_list) and t_total < max_seconds:
        obs = np.ascontiguousarray(obs_list[steps])
        # noise generation is not part of the reference's timed arithmetic budget here: the
        # reference draws from ChaCha12/ziggurat; we hand it ready samples
        det.det_normal2_v(1, 3, steps, 0, n, dp(z0), dp(z1))
        det.det_uniform2_v(1, 4, steps, 0, n, dp(r), dp(r2))
        nv, nw = sv * z0, sw * z1
        t0 = time.perf_counter()
        ref.ref_pf_step(n, dp(x), dp(y), dp(yaw), dp(v), dp(w), 1.0, 0.1, 0.1, dp(nv), dp(nw), dp(obs), L, 0.2, 1.0, 1,
                        dp(r), u32p(idx), dp(est))
        t_total += time.perf_counter() - t0
        steps += 1
    return dict(value=n * L * steps / t_total, unit="particle-landmark updates/s", cores=1, kind="port",
                sample=f"oracle/ref_literal.c ref_pf_step (literal reference arithmetic, binary-search multinomial "
                       f"resample), {n} particles x {L} landmarks x {steps} steps, {t_total:.1f} s, noise samples pre-drawn")


FS1_BYTES_PER_UPDATE = 96.0  # k_fs1_observe: read 48 B + write 48 B per (particle, observed landmark), EKF branch


def fs1_scene(L, seed, half=13.0):
    rng = np.random.default_rng(seed)
    return rng.uniform(-half, half, size=(L, 2))


def fs1_cpu_baseline(n, L, z_list, max_seconds=20.0):
    """fastslam_update of the literal C restatement (oracle/ref_literal.c), one host core."""
    import ctypes as C

    import oracle
    from oracle import dp, u32p

    ref, det = oracle.ref(), oracle.det()
    m = oracle.ref_fs1_model()
    m.init_cov = 0.5
    px, py, pyaw = (np.zeros(n) for _ in range(3))
    pw = np.full(n, 0.01)
    lm = np.tile(np.array([0, 0, 1000.0, 0, 0, 1000.0]), (n, L, 1)).reshape(-1).copy()
    idx = np.empty(n, np.uint32)
    z0, z1 = np.empty(n), np.empty(n)
    steps, t_total, updates = 0, 0.0, 0
    while steps < len(z_list) and t_total < max_seconds:
        z = np.ascontiguousarray(z_list[steps])
        det.det_normal2_v(2, 3, steps, 0, n, dp(z0), dp(z1))
        t0 = time.perf_counter()
        ref.ref_fs1_update(n, L, dp(px), dp(py), dp(pyaw), dp(pw), dp(lm), 0.5, 0.1, dp(z0), dp(z1), dp(z), len(z), C.byref(m),
                           n / 1.5, 0.3 / n, u32p(idx))
        t_total += time.perf_counter() - t0
        updates += n * len(z)
        steps += 1
    return dict(value=updates / t_total, unit="particle-landmark updates/s", cores=1, kind="port",
                sample=f"oracle/ref_literal.c ref_fs1_update (literal fastslam1.rs arithmetic), {n} particles x {L} landmarks x "
                       f"{steps} steps (first step takes the initialisation branch), {t_total:.1f} s")


def fs2_cpu_baseline(n, L, z_list, max_seconds=20.0):
    """fastslam2_update of the literal C restatement (oracle/ref_literal.c), one host core."""
    import oracle
    from oracle import dp, u32p

    ref = oracle.ref()
    px, py, pyaw = (np.zeros(n) for _ in range(3))
    pw = np.full(n, 0.01)
    lm = np.tile(np.array([0, 0, 1000.0, 0, 0, 1000.0]), (n, L, 1)).reshape(-1).copy()
    idx = np.empty(n, np.uint32)
    rng = np.random.default_rng(2)
    steps, t_total, updates = 0, 0.0, 0
    while steps < len(z_list) and t_total < max_seconds:
        z = np.ascontiguousarray(z_list[steps])
        noise = np.ascontiguousarray(rng.normal(size=(n, 3)))
        t0 = time.perf_counter()
        ref.ref_fs2_update(n, L, dp(px), dp(py), dp(pyaw), dp(pw), dp(lm), 0.5, 0.1, dp(noise), dp(z), len(z), n / 1.5, 0.3 / n, u32p(idx))
        t_total += time.perf_counter() - t0
        updates += n * len(z)
        steps += 1
    return dict(value=updates / t_total, unit="particle-landmark updates/s", cores=1, kind="port",
                sample=f"oracle/ref_literal.c ref_fs2_update (literal fastslam2.rs arithmetic), {n} particles x {L} landmarks x "
                       f"{steps} steps (first step takes the initialisation branch), {t_total:.1f} s, normals pre-drawn")


def run_fastslam(args):
    """BASELINE.json configs[2]: FastSLAM 1.0, 100 000 particles x 200 landmarks, every landmark observed
    every step, EKF branch (first_obs_cov = 0.5 initialises the maps on the first, untimed, step),
    N_eff threshold N/1.5 so that resampling triggers data-dependently."""
    from rust_robotics_amd.slam import fastslam1 as fs
    from tests import helpers as H

    n, L, K, W = args.particles, args.landmarks, args.steps, args.warmup
    lms = fs1_scene(L, 2)
    v2 = args.workload == "fastslam2"
    if v2:  # the same configuration with the FastSLAM 2.0 proposal (fastslam2.rs); first_obs_cov = 10 is its own constant
        from rust_robotics_amd.slam import fastslam2 as fs2

        prm2 = fs2.default_params()
        prm2.base.nth = n / 1.5
        f = fs2.FastSlam2(n, L, params=prm2, seed=2)
    else:
        prm = fs.default_params()
        prm.first_obs_cov = 0.5
        prm.nth = n / 1.5
        f = fs.FastSlam1(n, L, params=prm, seed=2)
    zs = [np.array(fs.get_observations(H.true_pose(t + 1, v=0.5), [tuple(p) for p in lms], seed=2, step=t)).reshape(-1, 3)
          for t in range(K + W)]
    u = [0.5, 0.1]
    for t in range(W):
        f.update_async(u, zs[t])
    f.synchronize()
    # the dominant kernel is timed INSIDE the timed region by the timestamps of its own dispatch packets
    f.profile_enable(2)
    f.profile_reset()
    t0 = time.perf_counter()
    for t in range(W, W + K):
        f.update_async(u, zs[t])
    f.synchronize()
    dt = time.perf_counter() - t0
    k_n, k_ms = f.profile_read()["k_fs1_observe"]
    updates = float(sum(n * len(zs[t]) for t in range(W, W + K)))
    # per-kernel breakdown: instrumented CONTINUATION over the same inputs (HIP events around every launch;
    # the filter has moved on, so these averages belong to later, calmer steps -- informational only)
    prof, dt_i = {"k_fs1_observe": (k_n, k_ms)}, 0.0
    if not args.no_breakdown:
        f.profile_enable(1)
        f.profile_reset()
        t1 = time.perf_counter()
        for t in range(W, W + K):
            f.update_async(u, zs[t])
        f.synchronize()
        dt_i = time.perf_counter() - t1
        prof = f.profile_read()
    f.profile_enable(0)
    pose, w, i = f.best_particle()
    avg_s = k_ms / max(k_n, 1) * 1e-3
    per_launch = FS1_BYTES_PER_UPDATE * n * np.mean([len(zs[t]) for t in range(W, W + K)])
    achieved = per_launch / avg_s
    out = {
        "metric": "particle-landmark updates/sec", "value": updates / dt, "unit": "particle-landmark updates/s", "n_gpus": 1,
        "steps": K, "warmup": W, "ms_per_step": dt / K * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": (f"FastSLAM 2.0 (the configs[2] shape with the proposal of fastslam2.rs): " if v2 else
                                "FastSLAM 1.0 (BASELINE.json configs[2]): ") +
                               f"{n} particles x {L} landmarks, all observed, 2x2 EKF branch, N_eff-gated systematic resample",
                   "particles_per_gpu": n, "landmarks": L},
        "roofline": {"bound": "hbm", "kernel": "k_fs1_observe", "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK,
                     "traffic": measured_traffic("k_fs1_observe", "fs1") if (n, L) == (100_000, 200) and not v2 else None,
                     "traffic_source": "profiles/r01f_pmc_hbm_traffic_fastslam_timed_region.csv (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE "
                                       "passes over `bench.py --workload fastslam --no-breakdown`, bytes per launch, read side x2)",
                     "avg_kernel_ms": avg_s * 1e3, "timed_launches": k_n,
                     "timing": "dispatch timestamps of the K launches inside the timed region",
                     "algorithmic_bytes_per_launch": per_launch},
        "kernel_ms_avg": {k: v[1] / max(v[0], 1) for k, v in prof.items() if v[0]},
        "kernel_launches": {k: v[0] for k, v in prof.items() if v[0]},
        "ms_per_step_instrumented": dt_i / K * 1e3,
        "obs_chunks": f.counters()[2],
        "best_particle": {"index": i, "weight": w, "pose": [float(a) for a in pose]},
    }
    if not args.no_cpu_baseline:
        out["cpu_baseline"] = (fs2_cpu_baseline if v2 else fs1_cpu_baseline)(min(n, 20000), L, zs)
    emit(out)


def run_fastslam_sharded(args, rank, world, local_rank):
    """BASELINE.json configs[3] shape: FastSLAM 1.0 sharded over the GPUs of a node (125 000 particles x 200
    landmarks per GPU at 8 GPUs = 1e6 x 200), weak scaling.  The step runs over the peer-to-peer transport
    (include/rr_fastslam1.h rr_fs1_shard_update_p2p); torch.distributed (gloo) only carries the IPC handles and
    the timing barrier.  Before anything is timed every rank checks, on this machine, that a small sharded run
    reproduces its block of the unsharded filter bit for bit."""
    import torch
    import torch.distributed as dist

    from rust_robotics_amd.sharded import gloo_allgather
    from rust_robotics_amd.slam import fastslam1 as fs
    from tests import helpers as H

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29555")  # a lone rank started without a launcher
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(local_rank)
    n, L, K, W = args.particles, args.landmarks, args.steps, args.warmup
    u = [0.5, 0.1]

    def observations(lms, steps, seed):
        return [np.array(fs.get_observations(H.true_pose(t + 1, v=0.5), [tuple(p) for p in lms], seed=seed, step=t)).reshape(-1, 3)
                for t in range(steps)]

    def make(n_local, Lm, chunks):
        prm = fs.default_params()
        prm.first_obs_cov = 0.5
        prm.nth = n_local * world / 1.5
        prm.initial_weight = 1.0 / (n_local * world)
        f = fs.ShardedFastSlam1(rank, world, n_local, Lm, device=local_rank, params=prm, seed=2, obs_chunks=chunks)
        f.connect_ipc(gloo_allgather(dist))
        return f, prm

    # run-time validation of the cross-GPU hand-off
    nv, Lv, Sv = 4096, 8, 8
    fv, prm_v = make(nv, Lv, 2)
    zv = observations(fs1_scene(Lv, 3), Sv, 3)
    dist.barrier()
    for z in zv:
        fv.update_async(u, z)
    ok = not fv.timed_out()
    if ok:
        whole = fs.FastSlam1(nv * world, Lv, params=prm_v, seed=2, device=local_rank, obs_chunks=2)
        for z in zv:
            whole.update_async(u, z)
        ep, em = whole.get_state()
        gp, gm = fv.get_state()
        sl = slice(rank * nv, (rank + 1) * nv)
        ok = np.array_equal(gp.view(np.uint64), ep[sl].view(np.uint64)) and np.array_equal(gm.view(np.uint64), em[sl].view(np.uint64))
        del whole
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    dist.barrier()
    # host-runtime warm-up: in a process that has torch's HIP context loaded, the first ~50 updates of the first
    # big filter are enqueued at ~0.85 ms each instead of ~0.05 ms (measured, scratch experiment in DESIGN.md
    # section 6); spend them on the small validation filter instead of inside the timed region
    for k in range(96):
        fv.update_async(u, zv[k % Sv])
    fv.synchronize()
    dist.barrier()
    del fv
    if not flag.item():
        raise SystemExit("sharded FastSLAM: the peer-to-peer transport did not reproduce the unsharded filter on this machine")

    f, _ = make(n, L, 0)
    zs = observations(fs1_scene(L, 2), K + W, 2)

    def fence():
        f.synchronize()
        torch.cuda.synchronize()
        dist.barrier()
        f.synchronize()
        torch.cuda.synchronize()

    for t in range(W):
        f.update_async(u, zs[t])
    fence()
    f.profile_enable(2)  # k_fs1_observe timed by its own dispatch timestamps, inside the timed region
    f.profile_reset()
    t0 = time.perf_counter()
    for t in range(W, W + K):
        f.update_async(u, zs[t])
    fence()
    dt = time.perf_counter() - t0
    dom = f.profile_read()["k_fs1_observe"]
    tmax = torch.tensor([dt], dtype=torch.float64)
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    timed_out = f.timed_out()
    f.profile_enable(1)
    f.profile_reset()
    t1 = time.perf_counter()
    for t in range(W, W + K):
        f.update_async(u, zs[t])
    f.synchronize()
    dt_i = time.perf_counter() - t1
    prof = f.profile_read()
    f.profile_enable(0)
    chunks = f.counters()[2]
    dist.barrier()
    del f
    dist.destroy_process_group()
    if rank != 0:
        return
    seconds = float(tmax.item())
    updates = float(sum(n * world * len(zs[t]) for t in range(W, W + K)))
    k_n, k_ms = dom
    avg_s = k_ms / max(k_n, 1) * 1e-3
    per_launch = FS1_BYTES_PER_UPDATE * n * np.mean([len(zs[t]) for t in range(W, W + K)])
    achieved = per_launch / avg_s
    emit({
        "metric": "particle-landmark updates/sec", "value": updates / seconds, "unit": "particle-landmark updates/s", "n_gpus": world,
        "steps": K, "warmup": W, "ms_per_step": seconds / K * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"FastSLAM 1.0 sharded (BASELINE.json configs[3] shape): {n} particles x {L} landmarks per GPU, "
                               f"{n * world} particles over {world} GPU(s), all landmarks observed, 2x2 EKF branch, N_eff-gated "
                               f"global systematic resample", "particles_per_gpu": n, "landmarks": L,
                   "transport": "p2p (xGMI, device-initiated); validated bit-identical to the unsharded filter at run time"},
        "roofline": {"bound": "hbm", "kernel": "k_fs1_observe", "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK, "traffic": None, "avg_kernel_ms": avg_s * 1e3,
                     "algorithmic_bytes_per_launch": per_launch},
        "kernel_ms_avg": {k: v[1] / max(v[0], 1) for k, v in prof.items() if v[0]},
        "kernel_launches": {k: v[0] for k, v in prof.items() if v[0]},
        "ms_per_step_instrumented": dt_i / K * 1e3, "obs_chunks": chunks, "p2p_timed_out": bool(timed_out),
    })


def replicas_fallback(rank, world, local_rank, n, L, K, W, obs_list, scheme, lik, reason):
    import torch
    import torch.distributed as dist

    import rust_robotics_amd.localization as loc

    cfg = loc.MonteCarloLocalizationConfig(min_particles=n, max_particles=n)
    pf = loc.MonteCarloLocalizer.with_initial_state([0.0, 0.0, 0.0, 1.0], cfg, seed=1 + rank, device=local_rank,
                                                    resample_scheme=scheme, likelihood_mode=lik)
    u = [1.0, 0.1]

    def fence():
        pf.synchronize()
        torch.cuda.synchronize()
        dist.barrier()
        pf.synchronize()
        torch.cuda.synchronize()

    W0 = max(len(obs_list) - 2 * K, W)  # W + 76: host-runtime warm-up of a process that has torch's HIP context loaded
    for t in range(W0):
        pf.step_async(u, obs_list[t])
    fence()
    t0 = time.perf_counter()
    for t in range(W0, W0 + K):
        pf.step_async(u, obs_list[t])
    fence()
    tmax = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    pf.profile_enable(1)
    pf.profile_reset()
    t1 = time.perf_counter()
    for t in range(W0 + K, W0 + 2 * K):
        pf.step_async(u, obs_list[t])
    pf.synchronize()
    dt_instr = time.perf_counter() - t1
    prof = pf.profile_read()
    pf.profile_enable(0)
    est = pf.estimate()
    dist.barrier()
    dist.destroy_process_group()
    return dict(seconds=float(tmax.item()), seconds_instrumented=dt_instr, kernels=prof, estimate=[float(a) for a in est],
                dominant=None, transport="NONE -- independent replicas, no exchange",
                transport_note="SHARDING FAILED: " + reason, p2p_timed_out=False, migrated_particles_last_step=0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", choices=["mcl", "fastslam", "fastslam2"], default="mcl")
    ap.add_argument("--particles", type=int, default=None, help="particles PER GPU (default 1e6 for mcl, 1e5 for fastslam)")
    ap.add_argument("--landmarks", type=int, default=None, help="default 32 for mcl, 200 for fastslam")
    ap.add_argument("--scheme", choices=["systematic", "multinomial"], default="systematic")
    ap.add_argument("--likelihood", choices=["fused", "product"], default="fused")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-breakdown", action="store_true",
                    help="skip the instrumented per-kernel re-run (use under rocprofv3 so that its averages cover the timed launches only)")
    ap.add_argument("--force-sharded", action="store_true", help="run the sharded path even at --gpus 1")
    ap.add_argument("--transport", choices=["auto", "p2p", "rccl", "torch", "p2p-only"], default="auto",
                    help="sharded exchange: auto = peer-to-peer over xGMI if it validates at run time, else native RCCL, else "
                         "torch.distributed NCCL; p2p / rccl / torch restrict the ladder; p2p-only validates against the unsharded filter")
    args = ap.parse_args()

    if os.environ.get("NCCL_DEBUG", "VERSION").upper() == "VERSION":
        os.environ["NCCL_DEBUG"] = "WARN"  # keep RCCL's version banner off stdout
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")

    if args.particles is None:
        args.particles = 1_000_000 if args.workload == "mcl" else 100_000
    if args.landmarks is None:
        args.landmarks = 32 if args.workload == "mcl" else 200
    if args.workload in ("fastslam", "fastslam2"):
        if args.steps == 200 and args.warmup == 20:
            args.steps, args.warmup = 50, 5
        if (world != 1 or args.force_sharded) and args.workload == "fastslam2":
            raise SystemExit("the sharded bench leg times FastSLAM 1.0 (the sharded update itself also serves FastSLAM 2.0 handles)")
        if world != 1 or args.force_sharded:
            if args.particles == 100_000 and world == 8:
                args.particles = 125_000  # configs[3]: 1e6 particles over 8 GPUs
            return run_fastslam_sharded(args, rank, world, local_rank)
        return run_fastslam(args)
    n, L, K, W = args.particles, args.landmarks, args.steps, args.warmup
    # time only moves forward for every filter: W warm-up + K timed + K dispatch-stamped + K breakdown steps; the sharded
    # legs also validate (12 steps) and warm up 64 steps longer
    obs_list = make_scene(L, W + 3 * K + (76 if (world > 1 or args.force_sharded) else 0), seed=1)
    scheme = 1 if args.scheme == "systematic" else 0
    lik = 0 if args.likelihood == "fused" else 1

    if world > 1 or args.force_sharded:
        from rust_robotics_amd import sharded

        try:
            res = sharded.bench_sharded(rank, world, local_rank, n, L, K, W, obs_list, scheme, lik, args.transport)
        except RuntimeError as e:
            # no sharded transport works on this machine.  Last resort so that the run still leaves a line:
            # every rank steps its own, independent filter (NO exchange, NOT one sharded filter) and the line
            # says so in config.sharding -- the number is an upper bound for the sharded step, not a measurement of it.
            res = replicas_fallback(rank, world, local_rank, n, L, K, W, obs_list, scheme, lik, str(e))
    else:
        import rust_robotics_amd.localization as loc

        cfg = loc.MonteCarloLocalizationConfig(min_particles=n, max_particles=n)
        pf = loc.MonteCarloLocalizer.with_initial_state([0.0, 0.0, 0.0, 1.0], cfg, seed=1, device=local_rank,
                                                        resample_scheme=scheme, likelihood_mode=lik)
        u = [1.0, 0.1]
        for t in range(W):
            pf.step_async(u, obs_list[t])
        pf.synchronize()
        t0 = time.perf_counter()
        for t in range(W, W + K):
            pf.step_async(u, obs_list[t])
        pf.synchronize()
        dt = time.perf_counter() - t0
        est = pf.estimate()
        # roofline kernel: the NEXT K steps (the filter resamples every step, so the work
        # per step is stationary) in which ONLY the propagate+weight kernel is timed, by the start/stop
        # timestamps of its own dispatch packets (hipExtLaunchKernelGGL on the filter's stream): no event
        # packets in the stream, the kernel runs as in the timed loop.  Kept out of the timed region because
        # the stamped launch costs the host ~3 us per step on this launch-rate-sensitive 3-launch step.
        pf.profile_enable(2)
        pf.profile_reset()
        for t in range(W + K, W + 2 * K):
            pf.step_async(u, obs_list[t])
        pf.synchronize()
        dominant = pf.profile_read()["k_propagate_weight"]
        # per-kernel breakdown: an instrumented re-run of the same K steps with HIP events around every
        # launch (adds ~3 us per launch; informational, kept out of `value` and of `roofline`)
        prof, dt_instr = {"k_propagate_weight": dominant}, 0.0
        if not args.no_breakdown:
            pf.profile_enable(1)
            pf.profile_reset()
            t1 = time.perf_counter()
            for t in range(W + 2 * K, W + 3 * K):
                pf.step_async(u, obs_list[t])
            pf.synchronize()
            dt_instr = time.perf_counter() - t1
            prof = pf.profile_read()
        pf.profile_enable(0)
        res = dict(seconds=dt, seconds_instrumented=dt_instr, kernels=prof, estimate=[float(a) for a in est], dominant=dominant)

    if rank != 0:
        return
    total_updates = float(n) * world * L * K
    value = total_updates / res["seconds"]
    kern = res["kernels"]
    dominant = res.get("dominant")
    if dominant and not dominant[0]:
        dominant = None  # this path does not stamp its dispatches (multinomial): fall back to the instrumented re-run
    k1_n, k1_ms = dominant or kern["k_propagate_weight"]
    k1_avg_s = (k1_ms / max(k1_n, 1)) * 1e-3
    k1_bytes = K1_BYTES[args.scheme] if world == 1 and not args.force_sharded else 64.0
    achieved = k1_bytes * n / k1_avg_s if k1_avg_s > 0 else 0.0
    step_kernel_ms = {k: (v[1] / max(v[0], 1)) for k, v in kern.items() if v[0]}
    out = {
        "metric": "particle-landmark updates/sec",
        "value": value,
        "unit": "particle-landmark updates/s",
        "n_gpus": world,
        "steps": K,
        "warmup": W,
        "ms_per_step": res["seconds"] / K * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": f"fixed-N MCL (BASELINE.json configs[1]): {n} particles/GPU x {L} landmarks, "
                        f"propagate+weight+{args.scheme} resample every step, likelihood={args.likelihood}",
            "particles_per_gpu": n,
            "landmarks": L,
            "resample": args.scheme,
            "sharding": "none" if world == 1 and not args.force_sharded else
                        (f"{res.get('transport')} ({res.get('transport_note')})" if str(res.get("transport", "")).startswith("NONE") else
                         f"contiguous particle blocks over {world} GPUs; transport {res.get('transport')} ({res.get('transport_note')})"),
        },
        "roofline": {
            "bound": "hbm",
            "kernel": "k_step_lazy (propagate + weight + folded resample gather)" if k1_bytes == 72.0 else "k_propagate_weight",
            "achieved": achieved / 1e9,
            "peak": HBM_PEAK / 1e9,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK,
            "traffic": measured_traffic("k_step_lazy", "mcl") if k1_bytes == 72.0 and n == 1_000_000 else None,
            "traffic_source": "profiles/r01c_pmc_hbm_traffic.csv (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, bytes per launch)",
            "avg_kernel_ms": k1_avg_s * 1e3,
            "timed_launches": k1_n,
            "timing": "dispatch timestamps (hipExtLaunchKernelGGL) of the K launches of the K steps that follow the timed region" if dominant else
                      "HIP events in an instrumented re-run of the K steps",
            "algorithmic_bytes_per_launch": k1_bytes * n,
            "note": "this kernel is FP64-VALU bound at L=32 (~19 f64-rate instructions per particle-landmark pair, "
                    "VALU issue slots ~94 % busy per rocprofv3 PMC, profiles/r01f_mcl_pmc_sq_summary.csv); the HBM-bound workload is `--workload fastslam` (DESIGN.md section 4)",
        },
        "kernel_ms_avg": step_kernel_ms,
        "ms_per_step_instrumented": res.get("seconds_instrumented", 0.0) / K * 1e3,
        "estimate": res.get("estimate"),
    }
    if not args.no_cpu_baseline and world == 1:  # rank 0 at N = 1 only
        out["cpu_baseline"] = cpu_baseline(n, L, obs_list)
    if world > 1 or args.force_sharded:
        out["sharded"] = {k: res.get(k) for k in ("transport", "transport_note", "p2p_timed_out", "migrated_particles_last_step")}
    emit(out)


def emit(out):
    """ONE JSON line, last on stdout: native libraries (RCCL's version banner) write through C
    stdio, so drain that buffer first."""
    import ctypes

    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    sys.stdout.write(json.dumps(out) + "\n")
    sys.stdout.flush()


if __name__ == "__main__":
    main()
