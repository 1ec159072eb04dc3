"""FastSLAM 1.0 / 2.0 legs: configs[2] (1e5 x 200), configs[3] at full size on one GPU, the sharded legs."""
import json
import math
import os
import socket
import subprocess
import sys
import time

import numpy as np

from .common import BENCH_PY, ROOT  # noqa: F401
from .common import DEVICE_WARMUP_FS, FS1_BYTES_PER_UPDATE, HBM_PEAK, ctx_transport, free_port, log, measured_traffic
from .cpu_baselines import fs1_cpu_baseline, fs1_scene, fs2_cpu_baseline


# ------------------------------------------------------------------------------------------------------
def leg_fastslam(args, n, L, K, W, v2=False, with_cpu=True, breakdown=True, device_warmup=None, label="configs[2]"):
    """BASELINE.json configs[2]: FastSLAM 1.0, 100 000 particles x 200 landmarks, every landmark observed
    every step, EKF branch (first_obs_cov = 0.5 initialises the maps on the first, untimed, step),
    N_eff threshold N/1.5 so that resampling triggers data-dependently."""
    from rust_robotics_amd.slam import fastslam1 as fs
    from tests import helpers as H

    lms = fs1_scene(L, 2)
    if v2:  # the same configuration with the FastSLAM 2.0 proposal (fastslam2.rs); first_obs_cov = 10 is its own constant
        from rust_robotics_amd.slam import fastslam2 as fs2

        prm2 = fs2.default_params()
        prm2.base.nth = n / 1.5
        f = fs2.FastSlam2(n, L, params=prm2, seed=2)
    else:
        prm = fs.default_params()
        prm.first_obs_cov = 0.5
        prm.nth = n / 1.5 * float(os.environ.get("RR_BENCH_NTH_SCALE", "1"))  # development knob: 0 = never resample, 10 = every step
        f = fs.FastSlam1(n, L, params=prm, seed=2, obs_chunks=int(os.environ.get("RR_BENCH_OBS_CHUNKS", "0")))  # 0 = the engine's own choice
    D = DEVICE_WARMUP_FS if device_warmup is None else device_warmup
    zs = [np.array(fs.get_observations(H.true_pose(t + 1, v=0.5), [tuple(p) for p in lms], seed=2, step=t)).reshape(-1, 3)
          for t in range(D + 2 * K + W)]
    u = [0.5, 0.1]
    for t in range(D):  # device warm-up (see DEVICE_WARMUP_FS), then time moves on
        f.update_async(u, zs[t])
    zs = zs[D:]
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
    # per-kernel breakdown: instrumented CONTINUATION (HIP events around every launch; the filter has moved
    # on, so these averages belong to later steps -- informational only)
    prof, dt_i = {"k_fs1_observe": (k_n, k_ms)}, 0.0
    if breakdown:
        f.profile_enable(1)
        f.profile_reset()
        t1 = time.perf_counter()
        for t in range(W + K, W + 2 * K):
            f.update_async(u, zs[t])
        f.synchronize()
        dt_i = time.perf_counter() - t1
        prof = f.profile_read()
    f.profile_enable(0)
    pose, w, i = f.best_particle()
    chunks = f.counters()[2]
    del f
    avg_s = k_ms / max(k_n, 1) * 1e-3
    per_launch = FS1_BYTES_PER_UPDATE * n * np.mean([len(zs[t]) for t in range(W, W + K)])
    achieved = per_launch / avg_s if avg_s > 0 else 0.0
    traffic, traffic_src = measured_traffic("k_fs1_observe", "fs2" if v2 else "fs1") if (n, L) == (100_000, 200) else (None, None)
    out = {
        "metric": "particle-landmark updates/sec", "value": updates / dt, "unit": "particle-landmark updates/s", "n_gpus": 1,
        "steps": K, "warmup": W, "ms_per_step": dt / K * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": (f"FastSLAM 2.0 (the configs[2] shape with the proposal of fastslam2.rs): " if v2 else
                                f"FastSLAM 1.0 (BASELINE.json {label}): ") +
                               f"{n} particles x {L} landmarks, all observed, 2x2 EKF branch, N_eff-gated systematic resample",
                   "particles_per_gpu": n, "landmarks": L},
        "roofline": {"bound": "hbm", "kernel": "k_fs1_observe", "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK, "traffic": traffic,
                     "traffic_source": ((traffic_src + (" (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over `bench.py --workload fastslam "
                                                        "--no-breakdown`, bytes per launch, read side x2)" if traffic is not None else ""))
                                        if traffic_src else None),
                     "avg_kernel_ms": avg_s * 1e3, "timed_launches": k_n,
                     "timing": "dispatch timestamps of the K launches inside the timed region",
                     "algorithmic_bytes_per_launch": per_launch,
                     "note": "`peak` is the HBM peak; a plain copy of the same bytes (dst[i] = src[i], 16 B per thread) runs at 4.6-6.2 TB/s on this "
                             "GPU depending on its launch shape (hipMemcpyAsync D2D: 5.1; tools/ubench/copy_rates.hip) and the kernel's own "
                             "access pattern with synthetic arithmetic at 5.2-5.4 TB/s (tools/ubench/plane_layout.hip, DESIGN.md section 4)"
                             + ("; FastSLAM 2.0 on this trajectory resamples to few distinct ancestors, so lanes share source lines and the "
                                "launch READS 0.74 GB instead of the algorithmic 0.96 GB (rocprofv3 FETCH_SIZE, DESIGN.md section 6): "
                                "`frac` is by algorithmic bytes, the moved bytes correspond to ~5.5 TB/s" if v2 else "")},
        "kernel_ms_avg": {k: v[1] / max(v[0], 1) for k, v in prof.items() if v[0]},
        "kernel_launches": {k: v[0] for k, v in prof.items() if v[0]},
        "ms_per_step_instrumented": dt_i / K * 1e3,
        "obs_chunks": chunks,
        "device_warmup_steps": D,
        "best_particle": {"index": i, "weight": w, "pose": [float(a) for a in pose]},
    }
    if with_cpu:
        # FastSLAM 1.0: the literal restatement at the FULL particle count (1e5 x 200: ~2 GB of host memory for the maps and the
        # clone buffer of the resample); FastSLAM 2.0 (not a BASELINE config) keeps the 20 000-particle sample
        out["cpu_baseline"] = fs2_cpu_baseline(min(n, 20000), L, zs) if v2 else fs1_cpu_baseline(n, L, zs)
    return out


def leg_fastslam_sharded(ctx, n, L, K, W):
    """BASELINE.json configs[3] shape: FastSLAM 1.0 sharded over the GPUs of a node (125 000 particles x 200
    landmarks per GPU at 8 GPUs = 1e6 x 200), weak scaling.  Transport ladder as for MCL: the peer-to-peer
    transport (rr_fs1_shard_update_p2p) is timed iff it connects and reproduces its block of the unsharded
    filter bit for bit on this machine; otherwise the RCCL transport (rr_fs1_shard_update: all-reduce MAX,
    all-gather of the integer sums, grouped send/recv of whole particles) -- itself validated the same way."""
    import torch

    from rust_robotics_amd.sharded import gloo_allgather, gloo_exchange
    from rust_robotics_amd.slam import fastslam1 as fs
    from tests import helpers as H

    dist = ctx.dist
    rank, world, local_rank = ctx.rank, ctx.world, ctx.local_rank
    u = [0.5, 0.1]
    notes = []

    def agree(ok):
        t = torch.tensor([1 if ok else 0], dtype=torch.int32)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(t.item())

    def observations(lms, steps, seed):
        return [np.array(fs.get_observations(H.true_pose(t + 1, v=0.5), [tuple(p) for p in lms], seed=seed, step=t)).reshape(-1, 3)
                for t in range(steps)]

    def params(n_local):
        prm = fs.default_params()
        prm.first_obs_cov = 0.5
        prm.nth = n_local * world / 1.5
        prm.initial_weight = 1.0 / (n_local * world)
        return prm

    def make(kind, n_local, Lm, chunks):
        f = fs.ShardedFastSlam1(rank, world, n_local, Lm, device=local_rank, params=params(n_local), seed=2, obs_chunks=chunks)
        if kind == "p2p":
            f.connect_ipc(gloo_allgather(dist))
        else:
            f.connect_rccl(gloo_exchange(dist))
        return f

    def attempt(kind, *a):
        obj, err = None, None
        try:
            obj = make(kind, *a)
        except Exception as e:  # noqa: BLE001 -- any failure means "next rung of the ladder"
            err = f"{type(e).__name__}: {e}"
        if agree(obj is not None):
            return obj
        notes.append(f"{kind} transport unavailable" + (f" ({err})" if err else " (failed on another rank)"))
        if obj is not None:
            obj.close()
        return None

    # run-time validation of the cross-GPU hand-off on a small filter
    nv, Lv, Sv = 4096, 8, 8
    zv = observations(fs1_scene(Lv, 3), Sv, 3)

    def validate(fv):
        ok = True
        try:
            for z in zv:
                fv.update_async(u, z)
            fv.synchronize()
            ok = not fv.timed_out()
        except Exception:  # a latched peer-wait timeout surfaces as an error
            ok = False
        if ok:
            whole = fs.FastSlam1(nv * world, Lv, params=params(nv), seed=2, device=local_rank, obs_chunks=2)
            for z in zv:
                whole.update_async(u, z)
            ep, em = whole.get_state()
            gp, gm = fv.get_state()
            sl = slice(rank * nv, (rank + 1) * nv)
            ok = np.array_equal(gp.view(np.uint64), ep[sl].view(np.uint64)) and np.array_equal(gm.view(np.uint64), em[sl].view(np.uint64))
            del whole
        return agree(ok)

    kind = None
    for cand in (("p2p", "rccl") if ctx_transport(ctx) == "auto" else (ctx_transport(ctx),)):
        log(f"fastslam sharded: trying the {cand} transport")
        fv = attempt(cand, nv, Lv, 2)
        if fv is None:
            continue
        dist.barrier()
        good = validate(fv)
        if good:
            # host-runtime warm-up: in a process that has torch's HIP context loaded, the first ~50 updates of the first
            # big filter are enqueued at ~0.85 ms each instead of ~0.05 ms; spend them here
            for k in range(96):
                fv.update_async(u, zv[k % Sv])
            fv.synchronize()
        dist.barrier()
        fv.close()
        notes.append(f"{cand} transport " + ("validated bit-identical to the unsharded filter" if good else "FAILED validation against the unsharded filter"))
        if good:
            kind = cand
            break
    if kind is None:
        return {"error": "sharded FastSLAM: no transport reproduced the unsharded filter on this machine", "transport_note": "; ".join(notes)}

    log(f"fastslam sharded: {kind} transport validated, timing {n} particles x {L} landmarks per GPU")
    f = attempt(kind, n, L, 0)
    if f is None:
        return {"error": "sharded FastSLAM: the validated transport could not be set up at full size", "transport_note": "; ".join(notes)}
    zs = observations(fs1_scene(L, 2), 2 * K + W, 2)
    # From here to the end every rank runs the SAME sequence of collectives whatever happens on its device: an error of one
    # rank (a latched peer-wait time-out surfaces as an exception of synchronize / update_async) is remembered, not raised,
    # and the ranks decide together at the end -- a rank that left early would leave the others in a barrier.
    trouble = []

    def quiet(fn, *a):
        try:
            return fn(*a)
        except Exception as e:  # noqa: BLE001
            if not trouble:
                trouble.append(f"rank {rank}: {type(e).__name__}: {e}")
            return None

    def fence():
        quiet(f.synchronize)
        torch.cuda.synchronize()
        dist.barrier()
        quiet(f.synchronize)
        torch.cuda.synchronize()

    for t in range(W):
        quiet(f.update_async, u, zs[t])
    fence()
    if not agree(not trouble and not quiet(f.timed_out)):  # do not spend K steps on a transport that is already dead
        note = "; ".join(notes + trouble + ["a peer wait gave up during the warm-up steps at full size"])
        quiet(f.close)
        return {"error": "sharded FastSLAM: the transport failed at full size", "transport_note": note}
    quiet(f.profile_enable, 2)  # k_fs1_observe timed by its own dispatch timestamps, inside the timed region
    quiet(f.profile_reset)
    t0 = time.perf_counter()
    for t in range(W, W + K):
        quiet(f.update_async, u, zs[t])
    fence()
    dt = time.perf_counter() - t0
    dom = (quiet(f.profile_read) or {}).get("k_fs1_observe", (0, 0.0))
    tmax = torch.tensor([dt], dtype=torch.float64)
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    timed_out = bool(quiet(f.timed_out))
    quiet(f.profile_enable, 1)
    quiet(f.profile_reset)
    t1 = time.perf_counter()
    for t in range(W + K, W + 2 * K):
        quiet(f.update_async, u, zs[t])
    quiet(f.synchronize)
    dt_i = time.perf_counter() - t1
    prof = quiet(f.profile_read) or {}
    quiet(f.profile_enable, 0)
    chunks = (quiet(f.counters) or (0, 0, 0))[2]
    dist.barrier()
    quiet(f.close)
    if not agree(not trouble and not timed_out):
        return {"error": "sharded FastSLAM: the transport failed inside the timed region",
                "transport_note": "; ".join(notes + trouble + (["a peer wait gave up"] if timed_out else []))}
    seconds = float(tmax.item())
    updates = float(sum(n * world * len(zs[t]) for t in range(W, W + K)))
    k_n, k_ms = dom
    avg_s = k_ms / max(k_n, 1) * 1e-3
    per_launch = FS1_BYTES_PER_UPDATE * n * np.mean([len(zs[t]) for t in range(W, W + K)])
    achieved = per_launch / avg_s if avg_s > 0 else 0.0
    return {
        "metric": "particle-landmark updates/sec", "value": updates / seconds, "unit": "particle-landmark updates/s", "n_gpus": world,
        "steps": K, "warmup": W, "ms_per_step": seconds / K * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"FastSLAM 1.0 sharded (BASELINE.json configs[3] shape): {n} particles x {L} landmarks per GPU, "
                               f"{n * world} particles over {world} GPU(s), all landmarks observed, 2x2 EKF branch, N_eff-gated "
                               f"global systematic resample", "particles_per_gpu": n, "landmarks": L,
                   "transport": ("p2p (xGMI, device-initiated)" if kind == "p2p" else "RCCL (all-reduce MAX, all-gather sums, grouped send/recv of whole particles)"),
                   "transport_note": "; ".join(notes)},
        "roofline": {"bound": "hbm", "kernel": "k_fs1_observe", "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK, "traffic": None, "avg_kernel_ms": avg_s * 1e3,
                     "algorithmic_bytes_per_launch": per_launch},
        "kernel_ms_avg": {k: v[1] / max(v[0], 1) for k, v in prof.items() if v[0]},
        "kernel_launches": {k: v[0] for k, v in prof.items() if v[0]},
        "ms_per_step_instrumented": dt_i / K * 1e3, "obs_chunks": chunks, "p2p_timed_out": bool(timed_out),
    }


def leg_fastslam_sharded_world1(n, L):
    """BASELINE.json configs[3] per-GPU shape (125 000 particles x 200 landmarks) through the sharded FastSLAM update with ONE
    rank, in a process of its own like the MCL world-1 legs."""
    log("extra leg fastslam_sharded_world1")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR")}
    env["MASTER_PORT"] = str(free_port())
    cmd = [sys.executable, BENCH_PY, "--gpus", "1", "--force-sharded", "--workload", "fastslam", "--particles", str(n), "--landmarks", str(L),
           "--no-cpu-baseline"]
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env)
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        if r.returncode != 0 or not lines:
            raise RuntimeError(f"rc {r.returncode}: {r.stderr[-400:]}")
        d = json.loads(lines[-1])
        return {k: d[k] for k in ("value", "unit", "ms_per_step", "steps", "warmup", "config", "roofline", "kernel_ms_avg", "obs_chunks") if k in d}
    except Exception as e:  # noqa: BLE001
        return {"error": f"{type(e).__name__}: {e}"}
