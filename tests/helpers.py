"""Shared scene builders and oracle drivers for the parity tests."""
import ctypes as C
import math

import numpy as np

import oracle
from oracle import dp, u32p, u64p

REF_SCENE_LANDMARKS = np.array([[10.0, 0.0], [0.0, 15.0], [-5.0, 20.0], [10.0, 10.0]])  # unified_filter_comparison.rs:43


def landmarks_grid(L, seed, half=20.0):
    rng = np.random.default_rng(seed)
    return rng.uniform(-half, half, size=(L, 2))


def true_pose(step, dt=0.1, v=1.0, w=0.1):
    """pose on the u = (1.0, 0.1) circle starting at the origin"""
    t = step * dt
    yaw = w * t
    return np.array([v / w * math.sin(yaw), v / w * (1 - math.cos(yaw)), yaw])


def observations(lms, pose, sigma, rng):
    d = np.hypot(lms[:, 0] - pose[0], lms[:, 1] - pose[1]) + rng.normal(0, sigma, len(lms))
    d = np.maximum(d, 0.0)
    return np.column_stack([d, lms[:, 0], lms[:, 1]])


def cloud(n, seed, center=(0.0, 0.0, 0.0, 1.0)):
    rng = np.random.default_rng(seed)
    x = center[0] + rng.uniform(-1, 1, n)
    y = center[1] + rng.uniform(-1, 1, n)
    yaw = center[2] + rng.uniform(-0.25, 0.25, n)
    v = center[3] + rng.uniform(-0.5, 0.5, n)
    return x, y, yaw, v


def aos(x, y, yaw, v, w):
    return np.ascontiguousarray(np.column_stack([x, y, yaw, v, w]))


def det_fixed(det, w, n_global=None):
    n = w.size
    sh, tot, qh, ql = C.c_int(), C.c_uint64(), C.c_uint64(), C.c_uint64()
    wmax = det.det_wmax(n, dp(w))
    usable = det.det_fix_reduce(n, dp(w), wmax, n_global or n, C.byref(sh), C.byref(tot), C.byref(qh), C.byref(ql))
    return dict(usable=usable, shift=sh.value, total=tot.value, q2_hi=qh.value, q2_lo=ql.value, wmax=wmax)


def det_cdf(det, w, fx, base=0):
    cdf = np.empty(w.size, np.uint64)
    det.det_fix_cdf(w.size, dp(w), fx["usable"], fx["shift"], base, u64p(cdf))
    return cdf


class DetPF:
    """Drives oracle/det_spec.c through whole steps with the engine's counter conventions."""

    def __init__(self, det, x, y, yaw, v, *, dt, sigma, sigma_v, sigma_w, threshold, gate, scheme, lik, seed):
        self.det = det
        self.x, self.y, self.yaw, self.v = (np.array(a, dtype=np.float64) for a in (x, y, yaw, v))
        self.n = self.x.size
        self.w = np.full(self.n, 1.0 / self.n)
        self.s = 1.0
        self.p = dict(dt=dt, sigma=sigma, sv=sigma_v, sw=sigma_w, thr=threshold, gate=gate, scheme=scheme, lik=lik, seed=seed)
        self.step_ctr = 0
        self.rstep_ctr = 0
        self.idx = np.zeros(self.n, np.uint32)

    def step(self, u, obs):
        p = self.p
        obs = np.ascontiguousarray(obs, dtype=np.float64)
        s = C.c_double()
        fired = self.det.det_pf_step(self.n, dp(self.x), dp(self.y), dp(self.yaw), dp(self.v), dp(self.w),
                                     u[0], u[1], p["dt"], p["sv"], p["sw"], dp(obs), obs.shape[0], p["sigma"], p["lik"],
                                     p["thr"], p["gate"], p["scheme"], p["seed"], self.step_ctr, self.rstep_ctr,
                                     u32p(self.idx), C.byref(s))
        self.step_ctr += 1
        self.rstep_ctr += 1
        self.s = s.value
        return bool(fired)

    def normalized_weights(self):
        return self.w / self.s

    def moments(self):
        est = np.empty(4)
        cov = np.empty(16)
        self.det.det_pf_moments(self.n, dp(self.x), dp(self.y), dp(self.yaw), dp(self.v), dp(self.w), self.s, dp(est), dp(cov))
        return est, cov.reshape(4, 4)


def unified_sim_data():
    """`generate_sim_data` of rust_robotics_localization/tests/unified_filter_comparison.rs:67-121 with the reference's own stream
    (StdRng::seed_from_u64(42) through oracle/rand_rs.py): per step one normal each for the velocity and yaw-rate input noise
    (sigma 0.3, 5 deg), the x and y position observation (0.5, 0.5) and the four landmark ranges (0.5, clamped at 0), in that
    order.  Returns (ground truth [100][4], noisy controls [100][2], landmark observations [100][4][3] = (d, lx, ly))."""
    from oracle import rand_rs as R

    rng = R.StdRng.seed_from_u64(42)
    lms = [(10.0, 0.0), (0.0, 15.0), (-5.0, 20.0), (10.0, 10.0)]
    x = np.zeros(4)
    truth, controls, lm_obs = [], [], []
    for _ in range(100):
        yaw = x[2]
        x = np.array([x[0] + 0.1 * 1.0 * math.cos(yaw), x[1] + 0.1 * 1.0 * math.sin(yaw), x[2] + 0.1 * 0.1, 1.0])
        u = (1.0 + R.normal(rng, 0.0, 0.3), 0.1 + R.normal(rng, 0.0, math.radians(5.0)))
        R.normal(rng, 0.0, 0.5)  # the position observation of the Kalman-family filters (x, then y): drawn, not used here
        R.normal(rng, 0.0, 0.5)
        obs = []
        for lx, ly in lms:
            d_true = math.sqrt((x[0] - lx) ** 2 + (x[1] - ly) ** 2)
            obs.append((max(d_true + R.normal(rng, 0.0, 0.5), 0.0), lx, ly))
        truth.append(x)
        controls.append(u)
        lm_obs.append(obs)
    return np.array(truth), np.array(controls), np.array(lm_obs)


def gpu_free_bytes(device=0):
    """Free device memory by the HIP runtime's own count (hipMemGetInfo), for the tests that need a large share of the 288 GB:
    they skip -- saying so -- on a device that cannot hold them (a partitioned or shared GPU), they do not fail the suite there."""
    hip = C.CDLL("libamdhip64.so")
    free, total = C.c_size_t(0), C.c_size_t(0)
    if hip.hipSetDevice(C.c_int(device)) != 0 or hip.hipMemGetInfo(C.byref(free), C.byref(total)) != 0:
        return 0
    return int(free.value)
