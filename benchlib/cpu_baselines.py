"""The `cpu_baseline` legs: the oracle's literal restatement (oracle/ref_literal.c) timed on the host cores beside the GPU number -- the
only place of bench.py that touches oracle/ (as the thing timed BESIDE the product, never as the product) -- and `index_parity`."""
import json
import math
import os
import socket
import subprocess
import sys
import time

import numpy as np

from .common import BENCH_PY, ROOT  # noqa: F401
from .common import _THREADS, host_cpu, pick_threads


# ------------------------------------------------------------------------------------------------------
# CPU baselines: oracle/ref_literal.c (the reference's arithmetic restated line by line) timed on this
# box's host cores.  SURVEY.md 8d: per-particle stages under OpenMP on all cores, cumsum / resample walk
# serial as in the reference; the reference's own O(N^2) multinomial resample timed separately at N = 1e4.
def cpu_baseline(n, L, obs_list, max_seconds=12.0, scheme="systematic", brief=False):
    """`value` = the literal restatement running the SAME step as the GPU leg it stands beside: the MCL step with the
    reference's systematic walk (fastslam1.rs:205-234) for the systematic legs, with its multinomial draws
    (monte_carlo_localization.rs:322-365, binary search) for the multinomial leg; the other variant is reported next to it."""
    import oracle
    from oracle import dp, u32p

    ref = oracle.ref()
    det = oracle.det()
    model, nproc = host_cpu()
    sv, sw = 2.0, math.radians(40.0)

    def run(threads, n_run, budget, literal_scan, scheme_id):
        used = ref.ref_set_threads(threads)
        x, y, yaw, v = (np.zeros(n_run) for _ in range(4))
        st = np.array([0.0, 0.0, 0.0, 1.0])
        det.det_pf_init(n_run, 1, 0, dp(st), dp(x), dp(y), dp(yaw), dp(v))
        w = np.full(n_run, 1.0 / n_run)
        idx = np.empty(n_run, np.uint32)
        est = np.empty(4)
        z0, z1, r, r2 = (np.empty(n_run) for _ in range(4))
        steps, t_total = 0, 0.0
        while steps < len(obs_list) and t_total < budget:
            obs = np.ascontiguousarray(obs_list[steps])
            # noise generation is not part of the timed arithmetic: the reference draws from ChaCha12/ziggurat,
            # we hand it ready samples
            det.det_normal2_v(1, 3, steps, 0, n_run, dp(z0), dp(z1))
            det.det_uniform2_v(1, 4, steps, 0, n_run, dp(r), dp(r2))
            nv, nw = sv * z0, sw * z1
            t0 = time.perf_counter()
            ref.ref_pf_step_ex(n_run, dp(x), dp(y), dp(yaw), dp(v), dp(w), 1.0, 0.1, 0.1, dp(nv), dp(nw), dp(obs), L, 0.2, 1.0, scheme_id,
                               dp(r), u32p(idx), dp(est), 1 if literal_scan else 0)
            dt = time.perf_counter() - t0
            if steps or budget < 1.0:  # the first step also pays for the thread team's creation
                t_total += dt
            steps += 1
        ref.ref_set_threads(1)
        timed = max(steps - (0 if budget < 1.0 else 1), 1)
        return n_run * L * timed / max(t_total, 1e-9), used, timed, t_total

    threads, table = pick_threads()
    main_id, other_id = (2, 1) if scheme == "systematic" else (1, 2)
    names = {1: "multinomial draws + binary search (monte_carlo_localization.rs:322-365,387-392)", 2: "systematic walk (fastslam1.rs:205-234)"}
    v_all, cores, s_all, t_all = run(threads, n, max_seconds, False, main_id)
    if brief:  # the extra legs: the like-for-like number only (the variants are in the headline legs of the same line)
        return dict(value=v_all, unit="particle-landmark updates/s", cores=cores, kind="port",
                    sample=f"oracle/ref_literal.c ref_pf_step (literal reference arithmetic; predict / weight / gather under OpenMP on {cores} threads, "
                           f"cumsum + resample -- {names[main_id]} -- serial), {n} particles x {L} landmarks x {s_all} steps, {t_all:.1f} s, noise samples pre-drawn",
                    host={"cpu_model": model, "nproc": nproc, "threads": cores})
    v_one, _, s_one, t_one = run(1, n, max_seconds / 2, False, main_id)
    v_oth, _, s_oth, t_oth = run(threads, n, max_seconds / 3, False, other_id)
    # the reference's own resample: a linear scan of the cumulative weights per draw (particle_filter.rs:455-470),
    # gate forced open (threshold 1.0 + scheme 0 fires whenever N_eff < N, i.e. always after a weight update)
    n_f = min(n, 10_000)
    v_f, _, s_f, t_f = run(threads, n_f, 4.0, True, 0)
    return dict(value=v_all, unit="particle-landmark updates/s", cores=cores, kind="port",
                sample=f"oracle/ref_literal.c ref_pf_step (literal reference arithmetic; predict / weight / gather under OpenMP on "
                       f"{cores} threads, cumsum + resample -- {names[main_id]} -- serial), {n} particles x {L} landmarks x {s_all} "
                       f"steps, {t_all:.1f} s, noise samples pre-drawn",
                host={"cpu_model": model, "nproc": nproc, "usable_cpus": _THREADS.get("avail"), "threads": cores,
                      "thread_calibration_updates_per_s": table},
                single_thread={"value": v_one, "steps": s_one, "seconds": round(t_one, 2)},
                other_resampler={"value": v_oth, "resample": names[other_id], "steps": s_oth, "seconds": round(t_oth, 2), "threads": cores},
                reference_faithful={"value": v_f, "particles": n_f, "steps": s_f, "seconds": round(t_f, 2), "threads": cores,
                                    "note": "the reference's own O(N^2) resample (linear scan per draw, particle_filter.rs:455-470); "
                                            "infeasible at 1e6 particles (~5e11 compares per step), so measured at N = 1e4 and never extrapolated"})


def index_parity(pf, n, L, scheme, obs):
    """Checker, not product: how many output slots of ONE resample at this size pick a different particle than the reference's
    own float walk over the same normalised weights and the same draws (the integer CDF is order-independent, the reference's
    serial float cumsum is not: DESIGN.md section 2).  Runs after the timed regions on the filter the leg just timed."""
    import oracle
    from oracle import dp, u32p

    ref = oracle.ref()
    rng = np.random.default_rng(17)
    pf.predict_with_control([1.0, 0.1])
    pf.update_with_observations(obs)
    w = pf.get_particles_array()[:, 4].copy()
    lit = np.empty(n, np.uint32)
    if scheme == "systematic":
        rho = float(np.floor(rng.random() * 2**53) / 2**53)
        pf.resample_systematic(rho)
        ref.ref_fs1_resample_indices(n, dp(w.copy()), rho / n, u32p(lit))
        walk = "fastslam1.rs:205-234 (r += 1/n accumulated serially)"
    else:
        r = np.floor(rng.random(n) * 2**53) / 2**53
        pf.resample_with_uniforms(r)
        ref.ref_mcl_resample_indices(n, dp(w), dp(r), u32p(lit))
        walk = "monte_carlo_localization.rs:328-392 (serial float cumsum, first i with r <= c[i])"
    got = pf.last_resample_indices()
    diff = np.nonzero(got != lit)[0]
    far = int(np.max(np.abs(got[diff].astype(np.int64) - lit[diff].astype(np.int64)))) if diff.size else 0
    return {"resample": scheme, "slots": n, "differing_slots_vs_literal_float_walk": int(diff.size), "max_index_distance": far,
            "literal_walk": walk,
            "note": "identical weights and draws into the engine and into the literal restatement; a differing slot picks the neighbouring "
                    "particle (the draw lies within the float cumsum's own rounding error of a boundary); bit-exact against the "
                    "order-independent integer CDF of the D-spec at every size (tests/)"}


def fs1_scene(L, seed, half=13.0):
    rng = np.random.default_rng(seed)
    return rng.uniform(-half, half, size=(L, 2))


def fs1_cpu_baseline(n, L, z_list, max_seconds=10.0):
    """fastslam_update of the literal C restatement (oracle/ref_literal.c), all host cores + one core."""
    import ctypes as C

    import oracle
    from oracle import dp, u32p

    ref, det = oracle.ref(), oracle.det()
    model, nproc = host_cpu()

    def run(threads, budget):
        used = ref.ref_set_threads(threads)
        m = oracle.ref_fs1_model()
        m.init_cov = 0.5
        px, py, pyaw = (np.zeros(n) for _ in range(3))
        pw = np.full(n, 0.01)
        lm = np.tile(np.array([0, 0, 1000.0, 0, 0, 1000.0]), (n, L, 1)).reshape(-1).copy()
        idx = np.empty(n, np.uint32)
        z0, z1 = np.empty(n), np.empty(n)
        steps, t_total, updates = 0, 0.0, 0
        while steps < len(z_list) and t_total < budget:
            z = np.ascontiguousarray(z_list[steps])
            det.det_normal2_v(2, 3, steps, 0, n, dp(z0), dp(z1))
            t0 = time.perf_counter()
            ref.ref_fs1_update(n, L, dp(px), dp(py), dp(pyaw), dp(pw), dp(lm), 0.5, 0.1, dp(z0), dp(z1), dp(z), len(z), C.byref(m),
                               n / 1.5, 0.3 / n, u32p(idx))
            dt = time.perf_counter() - t0
            if steps:  # step 0 takes the initialisation branch and creates the thread team
                t_total += dt
                updates += n * len(z)
            steps += 1
        ref.ref_set_threads(1)
        return updates / max(t_total, 1e-9), used, steps - 1, t_total

    threads, table = pick_threads()
    v_all, cores, s_all, t_all = run(threads, max_seconds)
    v_one, _, s_one, t_one = run(1, max_seconds / 2)
    return dict(value=v_all, unit="particle-landmark updates/s", cores=cores, kind="port",
                sample=f"oracle/ref_literal.c ref_fs1_update (literal fastslam1.rs arithmetic; predict / EKF / clone under OpenMP on {cores} "
                       f"threads, normalise + systematic walk serial), {n} particles x {L} landmarks x {s_all} EKF-branch steps, {t_all:.1f} s",
                host={"cpu_model": model, "nproc": nproc, "usable_cpus": _THREADS.get("avail"), "threads": cores,
                      "thread_calibration_updates_per_s": table},
                single_thread={"value": v_one, "steps": s_one, "seconds": round(t_one, 2)})


def fs2_cpu_baseline(n, L, z_list, max_seconds=10.0):
    """fastslam2_update of the literal C restatement (oracle/ref_literal.c), all host cores."""
    import oracle
    from oracle import dp, u32p

    ref = oracle.ref()
    model, nproc = host_cpu()
    cores = ref.ref_set_threads(pick_threads()[0])
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
        dt = time.perf_counter() - t0
        if steps:
            t_total += dt
            updates += n * len(z)
        steps += 1
    ref.ref_set_threads(1)
    return dict(value=updates / max(t_total, 1e-9), unit="particle-landmark updates/s", cores=cores, kind="port",
                sample=f"oracle/ref_literal.c ref_fs2_update (literal fastslam2.rs arithmetic, OpenMP over particles on {cores} threads), "
                       f"{n} particles x {L} landmarks x {steps - 1} steps, {t_total:.1f} s, normals pre-drawn",
                host={"cpu_model": model, "nproc": nproc, "threads": cores})
