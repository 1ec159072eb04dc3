"""The reference's two SEEDED FastSLAM 2.0 tests (rust_robotics_slam/src/fastslam2.rs:443-456 `test_fastslam2_update_does_not_panic`,
StdRng seed 7; :491-545 `test_landmark_convergence`, StdRng seed 17, `lm_err < 6.0`) replayed draw for draw: the random stream is
oracle/rand_rs.py's restatement of rand 0.9 StdRng + rand_distr 0.5.1, consumed in exactly the reference's order --

    get_observations_with_rng (:392-418)   per landmark within MAX_RANGE, in landmark order: one normal for the range, one for the bearing
    fastslam2_update_with_rng (:331-374)   per particle, in particle order: three normals (the proposal sample, :236) -- two when there is
                                           no observation (:350-354); then, only if N_eff < NTH, ONE Uniform(0, 1/n) for the resample (:310-311)

-- and handed to an engine through its explicit-noise seams.  Two engines: the literal restatement (oracle/ref_literal.c, CPU) and
the GPU (rr_fs2_predict_with_noise + rr_fs1_observe + rr_fs1_n_eff + rr_fs1_resample_systematic / rr_fs1_normalize_resample)."""
import math

import numpy as np

import oracle
from oracle import dp, u32p
from oracle import rand_rs as R

DT, MAX_RANGE, R00, R11, NTH, W0 = 0.1, 20.0, 0.5, 0.0305, 100.0 / 1.5, 1.0 / 100.0  # fastslam2.rs:14-35, :66


def normalize_angle(a):  # :85-94
    while a > math.pi:
        a -= 2.0 * math.pi
    while a < -math.pi:
        a += 2.0 * math.pi
    return a


def motion_model(x, u):  # :96-103
    return np.array([x[0] + u[0] * DT * math.cos(x[2]), x[1] + u[0] * DT * math.sin(x[2]), normalize_angle(x[2] + u[1] * DT)])


def observations_with_rng(x_true, landmarks, rng):  # :392-418
    z = []
    for lm_id, (lx, ly) in enumerate(landmarks):
        dx, dy = lx - x_true[0], ly - x_true[1]
        d = math.sqrt(dx * dx + dy * dy)
        if d <= MAX_RANGE:
            angle = normalize_angle(math.atan2(dy, dx) - x_true[2])
            d_noisy = d + R.normal(rng) * math.sqrt(R00)
            angle_noisy = angle + R.normal(rng) * math.sqrt(R11)
            z.append((d_noisy, angle_noisy, float(lm_id)))
    return np.array(z, dtype=np.float64).reshape(-1, 3)


class LiteralEngine:
    """oracle/ref_literal.c ref_fs2_update (fastslam2.rs:331-374 line by line), particles as create_particles leaves them (:425-429)"""

    def __init__(self, n, n_lm):
        self.n, self.L, self.ref = n, n_lm, oracle.ref()
        self.px, self.py, self.pyaw = (np.zeros(n) for _ in range(3))
        self.pw = np.full(n, W0)
        self.lm = np.tile(np.array([0, 0, 1000.0, 0, 0, 1000.0]), (n, n_lm, 1)).reshape(-1).copy()
        self.idx = np.empty(n, np.uint32)

    def _run(self, state, u, z, noise, r0):
        px, py, pyaw, pw, lm = state
        return self.ref.ref_fs2_update(self.n, self.L, dp(px), dp(py), dp(pyaw), dp(pw), dp(lm), u[0], u[1], dp(noise), dp(z) if len(z) else None, len(z),
                                       NTH, r0, u32p(self.idx))

    def update(self, u, z, noise, draw_r0):
        """draw_r0() is called only if the gate fires (the reference draws its uniform inside resample_with_rng)"""
        trial = [a.copy() for a in (self.px, self.py, self.pyaw, self.pw, self.lm)]
        fired = self._run(trial, u, z, noise, 0.0)
        if fired:
            self._run([self.px, self.py, self.pyaw, self.pw, self.lm], u, z, noise, draw_r0())
        else:
            self.px, self.py, self.pyaw, self.pw, self.lm = trial
        return bool(fired)

    def state(self):
        return self.pw, self.lm.reshape(self.n, self.L, 6)


class GpuEngine:
    def __init__(self, fs2, n, n_lm):
        self.n, self.f = n, fs2.FastSlam2(n, n_lm, seed=0)  # (the engine's own stream is never used: every draw comes from outside)

    def update(self, u, z, noise, draw_r0):
        self.f.propose_with_noise(u, z, noise)
        self.f.observe(z)
        fired = self.f.n_eff() < NTH
        if fired:
            self.f.resample_systematic(draw_r0() * self.n)
        else:
            self.f.normalize_resample()
            assert not self.f.last_resample_fired()
        return fired

    def state(self):
        poses, maps = self.f.get_state()
        return poses[:, 0], maps


def replay(engine, seed, n, landmarks, x_true, u, steps, truth_moves):
    rng = R.StdRng.seed_from_u64(seed)
    uniform = R.Uniform(0.0, 1.0 / n)
    fired = []
    for _ in range(steps):
        if truth_moves:
            x_true = motion_model(x_true, u)
        z = observations_with_rng(x_true, landmarks, rng)
        per = 3 if len(z) else 2
        noise = np.zeros((n, 3))
        for p in range(n):
            for k in range(per):
                noise[p, k] = R.normal(rng)
        fired.append(engine.update(u, z, np.ascontiguousarray(noise), lambda: uniform.sample(rng)))
    return fired, rng


def landmark_error(engine, lm_xy):
    """the closing assertion of :504-544: weighted mean of landmark 0 over the particles that have initialised it"""
    w, maps = engine.state()
    init = maps[:, 0, 2] < 100.0  # Landmark::is_initialized, :49-51
    assert init.any(), "at least one particle should initialize the landmark"
    ww = w[init]
    if ww.sum() > 0.0:
        mx, my = (ww * maps[init, 0, 0]).sum() / ww.sum(), (ww * maps[init, 0, 1]).sum() / ww.sum()
    else:
        mx, my = maps[init, 0, 0].mean(), maps[init, 0, 1].mean()
    return math.hypot(mx - lm_xy[0], my - lm_xy[1])


class PythonEngine:
    """oracle/ref_python.py's fastslam2_update -- the second, independently written restatement -- behind the same two methods"""

    def __init__(self, n, n_lm):
        from oracle import ref_python as P

        self.P, self.n, self.L = P, n, n_lm
        self.parts = [P.Particle(n_lm) for _ in range(n)]

    def update(self, u, z, noise, draw_r0):
        per = 3 if len(z) else 2
        self.parts, fired = self.P.fastslam2_update(self.parts, [float(u[0]), float(u[1])], [tuple(float(t) for t in row) for row in z],
                                                    [[float(t) for t in row[:per]] for row in noise], draw_r0)
        return fired

    def state(self):
        w = np.array([p.weight for p in self.parts])
        maps = np.array([[[m.x, m.y, m.cov[0][0], m.cov[1][0], m.cov[0][1], m.cov[1][1]] for m in p.landmarks] for p in self.parts])
        return w, maps

    def poses(self):
        return np.array([[p.x, p.y, p.yaw] for p in self.parts])
