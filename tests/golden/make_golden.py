#!/usr/bin/env python3
"""Regenerates tests/golden/*.npz from the literal C restatement of the reference
(oracle/ref_literal.c).  The reference itself (Rust) cannot be built in this image and its hot
path is not seedable, so these vectors pin (a) the restatement against silent edits and (b) the
GPU engine against the reference arithmetic on fixed, committed inputs: every random quantity
(motion noise, resample uniforms, observations) is stored as an INPUT array.

    python tests/golden/make_golden.py
"""
import ctypes as C
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import oracle  # noqa: E402
from oracle import dp, u32p  # noqa: E402
from tests import helpers as H  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def pf_mcl_case(n=96, steps=6, seed=2024):
    """fixed-N MCL semantics (monte_carlo_localization.rs:291-300): predict, weight, normalise,
    multinomial resample every step with forced last cum = 1.0 / fallback last."""
    ref = oracle.ref()
    rng = np.random.default_rng(seed)
    lms = H.REF_SCENE_LANDMARKS
    sig, sv, sw, dt = 0.5, 0.3, math.radians(5.0), 0.1
    x, y, yaw, v = H.cloud(n, seed + 1)
    w = np.full(n, 1.0 / n)
    rec = dict(init_x=x.copy(), init_y=y.copy(), init_yaw=yaw.copy(), init_v=v.copy(), sigma=sig, dt=dt, u=np.array([1.0, 0.1]))
    idx = np.empty(n, np.uint32)
    est = np.empty(4)
    for t in range(steps):
        obs = H.observations(lms, H.true_pose(t + 1), sig, rng)
        nv, nw = rng.normal(0, sv, n), rng.normal(0, sw, n)
        r = np.floor(rng.random(n) * 2**53) / 2**53
        ref.ref_pf_predict(n, dp(x), dp(y), dp(yaw), dp(v), 1.0, 0.1, dt, dp(nv), dp(nw))
        ref.ref_pf_update_raw(n, dp(x), dp(y), dp(w), dp(obs), len(obs), sig)
        raw = w.copy()
        ref.ref_pf_normalize(n, dp(w))
        wn = w.copy()
        neff = ref.ref_pf_neff(n, dp(w))
        ref.ref_pf_estimate(n, dp(x), dp(y), dp(yaw), dp(v), dp(w), dp(est))
        cov = np.empty(16)
        ref.ref_pf_covariance(n, dp(x), dp(y), dp(yaw), dp(v), dp(w), dp(est), dp(cov))
        ref.ref_mcl_resample_indices(n, dp(w), dp(r), u32p(idx))
        idx_pf = np.empty(n, np.uint32)
        ref.ref_pf_resample_indices(n, dp(w), dp(r), u32p(idx_pf))
        ref.ref_pf_gather(n, dp(x), dp(y), dp(yaw), dp(v), dp(w), u32p(idx))
        rec.update({f"obs{t}": obs, f"nv{t}": nv, f"nw{t}": nw, f"r{t}": r, f"raw{t}": raw, f"wn{t}": wn,
                    f"neff{t}": neff, f"est{t}": est.copy(), f"cov{t}": cov, f"idx{t}": idx.copy(), f"idx_pf{t}": idx_pf,
                    f"x{t}": x.copy(), f"y{t}": y.copy(), f"yaw{t}": yaw.copy(), f"v{t}": v.copy()})
    rec["steps"] = steps
    np.savez_compressed(os.path.join(OUT, "pf_mcl_n96.npz"), **rec)


def fs1_case(n=40, L=5, steps=6, seed=77):
    """fastslam_update (fastslam1.rs:237-266) with first_obs_cov = 2.0 so that the EKF branch runs,
    NTH = n/1.5."""
    ref = oracle.ref()
    rng = np.random.default_rng(seed)
    lms = rng.uniform(-8, 8, size=(L, 2))
    m = oracle.ref_fs1_model()
    m.init_cov = 2.0
    px, py, pyaw = (np.zeros(n) for _ in range(3))
    pw = np.full(n, 0.01)
    lm = np.tile(np.array([0, 0, 1000.0, 0, 0, 1000.0]), (n, L, 1)).reshape(-1).copy()
    rec = dict(landmarks=lms, first_obs_cov=2.0, nth=n / 1.5, u=np.array([1.0, 0.1]), steps=steps, n=n, L=L)
    idx = np.empty(n, np.uint32)
    for t in range(steps):
        xt = H.true_pose(t + 1)
        zn = rng.normal(size=2 * L)
        z = np.empty((L, 3))
        cnt = ref.ref_fs1_get_observations(dp(xt), dp(np.ascontiguousarray(lms)), L, 20.0, dp(zn), C.byref(m), dp(z))
        z = np.ascontiguousarray(z[:cnt])
        z0, z1 = rng.normal(size=n), rng.normal(size=n)
        rho = float(np.floor(rng.random() * 2**53) / 2**53)
        fired = ref.ref_fs1_update(n, L, dp(px), dp(py), dp(pyaw), dp(pw), dp(lm), 1.0, 0.1, dp(z0), dp(z1), dp(z), cnt,
                                   C.byref(m), n / 1.5, rho / n, u32p(idx))
        rec.update({f"z{t}": z, f"z0_{t}": z0, f"z1_{t}": z1, f"rho{t}": rho, f"fired{t}": fired,
                    f"idx{t}": idx.copy() if fired else np.zeros(0, np.uint32), f"px{t}": px.copy(), f"py{t}": py.copy(),
                    f"pyaw{t}": pyaw.copy(), f"pw{t}": pw.copy(), f"lm{t}": lm.copy(),
                    f"best{t}": ref.ref_fs1_best_particle(n, dp(pw))})
    np.savez_compressed(os.path.join(OUT, "fs1_n40_l5.npz"), **rec)


def fs2_case(n=40, L=4, steps=7, seed=78):
    """fastslam2_update_with_rng (fastslam2.rs:331-374): proposal sampling from the first observation,
    landmark EKF with the FastSLAM 2.0 constants, NTH = n/1.5; step 3 has no observation (:349-357)."""
    ref = oracle.ref()
    rng = np.random.default_rng(seed)
    lms = rng.uniform(-8, 8, size=(L, 2))
    m = oracle.ref_fs1_model()
    px, py, pyaw = (np.zeros(n) for _ in range(3))
    pw = np.full(n, 1.0 / n)
    lm = np.tile(np.array([0, 0, 1000.0, 0, 0, 1000.0]), (n, L, 1)).reshape(-1).copy()
    rec = dict(landmarks=lms, nth=n / 1.5, u=np.array([1.0, 0.1]), steps=steps, n=n, L=L)
    idx = np.empty(n, np.uint32)
    for t in range(steps):
        xt = H.true_pose(t + 1)
        zn = rng.normal(size=2 * L)
        z = np.empty((L, 3))
        cnt = ref.ref_fs1_get_observations(dp(xt), dp(np.ascontiguousarray(lms)), L, 20.0, dp(zn), C.byref(m), dp(z))
        z = np.ascontiguousarray(z[:0] if t == 3 else z[:cnt])
        noise = np.ascontiguousarray(rng.normal(size=(n, 3)))
        rho = float(np.floor(rng.random() * 2**53) / 2**53)
        fired = ref.ref_fs2_update(n, L, dp(px), dp(py), dp(pyaw), dp(pw), dp(lm), 1.0, 0.1, dp(noise), dp(z) if len(z) else None,
                                   len(z), n / 1.5, rho / n, u32p(idx))
        rec.update({f"z{t}": z, f"noise{t}": noise, f"rho{t}": rho, f"fired{t}": fired,
                    f"idx{t}": idx.copy() if fired else np.zeros(0, np.uint32), f"px{t}": px.copy(), f"py{t}": py.copy(),
                    f"pyaw{t}": pyaw.copy(), f"pw{t}": pw.copy(), f"lm{t}": lm.copy()})
    np.savez_compressed(os.path.join(OUT, "fs2_n40_l4.npz"), **rec)


def kld_case(seed=79):
    """resample_adaptive (monte_carlo_localization.rs:322-365) on three clouds: new count and the
    source index of every draw, for the uniforms stored as inputs."""
    ref = oracle.ref()
    rng = np.random.default_rng(seed)
    rec = {}
    cases = []
    n = 400
    cases.append((rng.uniform(-5, 5, n), rng.uniform(-5, 5, n), rng.uniform(-3, 3, n), rng.random(n) + 0.02, 100, 1500))
    cases.append((1.0 + rng.normal(0, 0.03, n), 1.0 + rng.normal(0, 0.03, n), 0.1 + rng.normal(0, 0.01, n), rng.random(n) + 0.02, 50, 600))
    i = np.arange(n)
    cases.append(((i % 4) * 3.0 + i * 0.002, (i % 4) * 2.0 + 0.0 * i, 0.0 * i, np.full(n, 1.0), 100, 1500))  # the reference's own test cloud
    for c, (x, y, yaw, w, lo, hi) in enumerate(cases):
        x, y, yaw = (np.ascontiguousarray(a, dtype=np.float64) for a in (x, y, yaw))
        w = np.ascontiguousarray(w / w.sum())
        r = np.floor(rng.random(hi) * 2**53) / 2**53
        idx = np.empty(hi, np.uint32)
        cnt = ref.ref_mcl_resample_adaptive(n, dp(x), dp(y), dp(yaw), dp(w), dp(r), lo, hi, 0.05, 2.326, u32p(idx))
        rec.update({f"x{c}": x, f"y{c}": y, f"yaw{c}": yaw, f"w{c}": w, f"r{c}": r, f"min{c}": lo, f"max{c}": hi, f"count{c}": cnt,
                    f"idx{c}": idx[:cnt].copy()})
    rec["cases"] = len(cases)
    np.savez_compressed(os.path.join(OUT, "kld_adaptive.npz"), **rec)


if __name__ == "__main__":
    pf_mcl_case()
    fs1_case()
    fs2_case()
    kld_case()
    for f in sorted(os.listdir(OUT)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(OUT, f)), "bytes")
