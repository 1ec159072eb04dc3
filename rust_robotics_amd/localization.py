"""Host-side mirror of ``rust_robotics_localization``'s particle filters over the HIP engine.

Same names, argument meaning and error behaviour as
  /root/reference/crates/rust_robotics_localization/src/particle_filter.rs:121-573
  /root/reference/crates/rust_robotics_localization/src/monte_carlo_localization.rs:136-462
so that the reference's tests and examples read the same against this module (the Rust
crate that binds the same C ABI is sketched in INTEGRATION.md).  Every method is a thin
call into ``librust_robotics_amd.so``; nothing is computed in Python and nothing falls back
to the CPU.

Particle state lives on the GPU.  ``get_particles()`` copies it back (N x 5 doubles).
"""
from __future__ import annotations

import ctypes as C
import math
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import _ffi
from .core import ControlInput, Obstacles, Point2D, RoboticsError, State2D

PFState = np.ndarray  # Vector4 (x, y, yaw, v)            particle_filter.rs:16
PFControl = np.ndarray  # Vector2 (v, yaw_rate)            particle_filter.rs:19
PFMeasurement = List[Tuple[float, float, float]]  # (d, landmark_x, landmark_y)  particle_filter.rs:22


def _check(status: int) -> None:
    if status == _ffi.RR_OK:
        return
    msg = _ffi.last_error()
    if status == _ffi.RR_INVALID_PARAMETER:
        raise RoboticsError.invalid_parameter(msg)
    raise RoboticsError.runtime(msg)


def _dp(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _vec(v, n: int, what: str) -> np.ndarray:
    a = np.ascontiguousarray(np.asarray(v, dtype=np.float64).reshape(-1))
    if a.size != n:
        raise RoboticsError.invalid_parameter(f"{what} must have {n} components")
    return a


def _obs_array(observations) -> np.ndarray:
    a = np.ascontiguousarray(np.asarray(observations, dtype=np.float64).reshape(-1, 3)) if len(observations) else np.zeros((0, 3))
    return a


@dataclass
class Particle:
    """particle_filter.rs:25-32"""

    x: float
    y: float
    yaw: float
    v: float
    w: float

    def to_state(self) -> np.ndarray:
        return np.array([self.x, self.y, self.yaw, self.v])


@dataclass
class ParticleFilterConfig:
    """particle_filter.rs:51-78 (defaults of ``impl Default``)"""

    n_particles: int = 100
    resample_threshold: float = 0.5
    range_noise: float = 0.2
    velocity_noise: float = 2.0
    yaw_rate_noise: float = math.radians(40.0)
    dt: float = 0.1

    def _c(self) -> _ffi.PfConfig:
        if self.n_particles < 0:
            raise RoboticsError.invalid_parameter("particle filter requires at least one particle")
        return _ffi.PfConfig(int(self.n_particles), self.resample_threshold, self.range_noise,
                             self.velocity_noise, self.yaw_rate_noise, self.dt)

    def validate(self) -> None:
        """particle_filter.rs:81-117"""
        _check(_ffi.lib().rr_pf_config_validate(C.byref(self._c())))


class ParticleFilterLocalizer:
    """particle_filter.rs:121-573 over the MI355X engine.

    Engine-only keyword arguments (the reference has no counterpart because its RNG is the
    unseedable thread-local generator): ``seed`` keys the Philox noise streams, ``device`` picks
    the GPU, ``resample_scheme`` / ``likelihood_mode`` select the resampler and the (equivalent)
    likelihood evaluation, ``record_indices`` keeps the last resample's source indices.
    """

    _GATE = _ffi.RR_GATE_NEFF
    _SCHEME = _ffi.RR_RESAMPLE_MULTINOMIAL

    def __init__(self, config: Optional[ParticleFilterConfig] = None, *, _initial_state=None, seed: int = 0,
                 device: int = 0, resample_scheme: Optional[int] = None, likelihood_mode: int = _ffi.RR_LIK_FUSED,
                 record_indices: bool = False, first_global_index: int = 0, n_global: int = 0):
        config = config or ParticleFilterConfig()
        L = _ffi.lib()
        opt = _ffi.PfOptions()
        L.rr_pf_options_default(C.byref(opt))
        opt.device = device
        opt.seed = seed
        opt.resample_gate = self._GATE
        opt.resample_scheme = self._SCHEME if resample_scheme is None else resample_scheme
        opt.likelihood_mode = likelihood_mode
        opt.record_indices = 1 if record_indices else 0
        opt.first_global_index = first_global_index
        opt.n_global = n_global
        self._h = C.c_void_p()
        cfg = config._c()
        if _initial_state is None:
            _check(L.rr_pf_create(C.byref(cfg), C.byref(opt), C.byref(self._h)))
        else:
            st = _vec(_initial_state, 4, "particle filter state")
            _check(L.rr_pf_create_with_state(C.byref(cfg), C.byref(opt), _dp(st), C.byref(self._h)))
        self.config = config
        self._L = L
        self._state_estimate = np.zeros(4)
        self._covariance = np.zeros((4, 4))
        self._cache_valid = False

    # ---- constructors (particle_filter.rs:131-207)
    @classmethod
    def new(cls, config: ParticleFilterConfig, **kw) -> "ParticleFilterLocalizer":
        return cls(config, **kw)

    try_new = new

    @classmethod
    def with_defaults(cls, **kw) -> "ParticleFilterLocalizer":
        return cls(ParticleFilterConfig(), **kw)

    @classmethod
    def with_initial_state(cls, initial_state, config: ParticleFilterConfig, **kw) -> "ParticleFilterLocalizer":
        return cls(config, _initial_state=initial_state, **kw)

    try_with_initial_state = with_initial_state

    @classmethod
    def with_initial_state_2d(cls, initial_state: State2D, config: ParticleFilterConfig, **kw):
        return cls(config, _initial_state=initial_state.to_vector(), **kw)

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            self._L.rr_pf_destroy(h)
            self._h = None

    # ---- landmarks (particle_filter.rs:209-241)
    def set_landmarks(self, landmarks: Sequence[Point2D]) -> None:
        xy = np.array([[p.x, p.y] for p in landmarks], dtype=np.float64).reshape(-1)
        _check(self._L.rr_pf_set_landmarks(self._h, _dp(xy) if xy.size else None, len(landmarks)))

    try_set_landmarks = set_landmarks

    def set_landmarks_from_obstacles(self, landmarks: Obstacles) -> None:
        self.set_landmarks(landmarks.points)

    def get_landmarks(self) -> List[Point2D]:
        n = self._L.rr_pf_landmark_count(self._h)
        xy = np.zeros(2 * n)
        self._L.rr_pf_get_landmarks(self._h, _dp(xy), n)
        return [Point2D(xy[2 * k], xy[2 * k + 1]) for k in range(n)]

    def set_range_noise(self, range_noise: float) -> None:
        _check(self._L.rr_pf_set_range_noise(self._h, float(range_noise)))
        self.config.range_noise = float(range_noise)

    # ---- particles
    def get_particles_array(self) -> np.ndarray:
        """N x 5 array (x, y, yaw, v, w) -- the bulk form of get_particles()."""
        n = self.particle_count()
        out = np.empty((n, 5))
        _check(self._L.rr_pf_get_particles(self._h, _dp(out)))
        return out

    def get_particles(self) -> List[Particle]:
        """particle_filter.rs:244-246"""
        return [Particle(*row) for row in self.get_particles_array()]

    def particle_count(self) -> int:
        return int(self._L.rr_pf_particle_count(self._h))

    # ---- predict / update / resample (particle_filter.rs:248-345)
    def predict_with_control(self, control) -> None:
        u = _vec(control, 2, "particle filter control input")
        _check(self._L.rr_pf_predict(self._h, _dp(u)))
        self._cache_valid = False

    try_predict_with_control = predict_with_control

    def try_predict_input(self, control: ControlInput) -> None:
        self.predict_with_control(control.to_vector())

    def update_with_observations(self, observations: PFMeasurement) -> None:
        obs = _obs_array(observations)
        _check(self._L.rr_pf_update(self._h, _dp(obs) if obs.size else None, obs.shape[0]))
        self._cache_valid = False

    try_update_with_observations = update_with_observations

    def resample(self) -> None:
        _check(self._L.rr_pf_resample(self._h))
        self._cache_valid = False

    # ---- step (particle_filter.rs:481-497, 368-380)
    def step(self, control, observations: PFMeasurement) -> np.ndarray:
        u = _vec(control, 2, "particle filter control input")
        obs = _obs_array(observations)
        out = np.empty(4)
        _check(self._L.rr_pf_step(self._h, _dp(u), _dp(obs) if obs.size else None, obs.shape[0], _dp(out)))
        self._state_estimate = out
        self._cache_valid = False
        return out.copy()

    try_step = step

    def try_step_state(self, control: ControlInput, observations: PFMeasurement) -> State2D:
        e = self.step(control.to_vector(), observations)
        return State2D(e[0], e[1], e[2], e[3])

    def step_async(self, control, observations) -> None:
        """Enqueue one step without waiting for its result (engine extension)."""
        u = _vec(control, 2, "particle filter control input")
        obs = _obs_array(observations)
        _check(self._L.rr_pf_step_async(self._h, _dp(u), _dp(obs) if obs.size else None, obs.shape[0]))
        self._cache_valid = False

    def step_async_estimate(self, control, observations) -> None:
        """``step_async`` that also leaves the mean ``try_step`` returns (particle_filter.rs:496) on the device --
        computed inside the step's plan kernel (systematic) or by the kernel that gathers the drawn sources
        (multinomial); read it with ``last_step_estimate`` (engine extension)."""
        u = _vec(control, 2, "particle filter control input")
        obs = _obs_array(observations)
        _check(self._L.rr_pf_step_async_estimate(self._h, _dp(u), _dp(obs) if obs.size else None, obs.shape[0]))
        self._cache_valid = False

    def step_many(self, controls, observations, estimates: bool = True):
        """``len(controls)`` steps in one call (``rr_pf_step_many``): controls K x 2, observations K x n_obs x 3 (the same
        number of observations every step).  Returns the K estimates ``try_step`` would have returned (K x 4), or None
        with ``estimates=False`` (asynchronous).  Up to 2048 particles the whole batch is ONE kernel launch (engine extension)."""
        u = np.ascontiguousarray(controls, dtype=np.float64).reshape(-1, 2)
        K = u.shape[0]
        obs = np.ascontiguousarray(observations, dtype=np.float64).reshape(K, -1, 3) if K else np.zeros((0, 0, 3))
        out = np.empty((K, 4)) if estimates else None
        _check(self._L.rr_pf_step_many(self._h, _dp(u), _dp(obs) if obs.size else None, obs.shape[1], K, _dp(out) if estimates else None))
        self._cache_valid = False
        if estimates and K:
            self._state_estimate = out[-1].copy()
        return out

    def last_step_estimate(self) -> np.ndarray:
        e = np.empty(4)
        _check(self._L.rr_pf_last_step_estimate(self._h, _dp(e)))
        return e

    def warm(self, ms: float = 0.0) -> None:
        """rr_pf_warm (engine extension): `ms` milliseconds (0: the default, 50) of step-shaped work on the filter's stream, so that
        the first step runs at the rate of the thousandth (an idle MI355X starts at reduced clocks).  The particles are not touched."""
        _check(self._L.rr_pf_warm(self._h, float(ms)))

    def synchronize(self) -> None:
        _check(self._L.rr_pf_synchronize(self._h))

    def set_resident(self, idle_us: float) -> None:
        """Resident service (``rr_pf_set_resident``, engine extension): with ``idle_us > 0`` the steps of a filter of up to
        2048 particles are served by ONE kernel that stays on the device between them (no launch per step); it leaves by
        itself after ``idle_us`` microseconds without a step.  0 switches it off."""
        _check(self._L.rr_pf_set_resident(self._h, float(idle_us)))

    def resident_stats(self):
        """(incarnations of the resident kernel launched, steps served by them)"""
        import ctypes as C

        a, b = C.c_uint64(0), C.c_uint64(0)
        _check(self._L.rr_pf_resident_stats(self._h, C.byref(a), C.byref(b)))
        return int(a.value), int(b.value)

    # ---- estimate / covariance (particle_filter.rs:347-365; evaluated lazily on the GPU)
    def _refresh(self) -> None:
        if self._cache_valid:
            return
        e = np.empty(4)
        c = np.empty(16)
        _check(self._L.rr_pf_estimate(self._h, _dp(e)))
        _check(self._L.rr_pf_covariance(self._h, _dp(c)))
        self._state_estimate, self._covariance = e, c.reshape(4, 4)
        self._cache_valid = True

    def estimate(self) -> np.ndarray:
        self._refresh()
        return self._state_estimate.copy()

    def state_2d(self) -> State2D:
        e = self.estimate()
        return State2D(e[0], e[1], e[2], e[3])

    def calc_covariance(self) -> np.ndarray:
        self._refresh()
        return self._covariance.copy()

    def n_eff(self) -> float:
        out = C.c_double()
        _check(self._L.rr_pf_n_eff(self._h, C.byref(out)))
        return out.value

    def last_resample_fired(self) -> bool:
        out = C.c_int32()
        _check(self._L.rr_pf_last_resample_fired(self._h, C.byref(out)))
        return bool(out.value)

    # ---- StateEstimator (particle_filter.rs:552-573)
    def predict(self, control, _dt: float = 0.0) -> None:  # dt ignored (Q17)
        self.predict_with_control(control)

    def update(self, measurement: PFMeasurement) -> None:
        self.update_with_observations(measurement)
        self.resample()

    def get_state(self) -> np.ndarray:
        return self.estimate()

    def get_covariance(self) -> Optional[np.ndarray]:
        return self.calc_covariance()

    # ---- parity seams (include/rr_pf.h "parity seams")
    def set_particles_array(self, aos: np.ndarray) -> None:
        a = np.ascontiguousarray(aos, dtype=np.float64).reshape(-1, 5)
        if a.shape[0] != self.particle_count():
            raise RoboticsError.invalid_parameter("need one row per particle")
        _check(self._L.rr_pf_set_particles(self._h, _dp(a)))
        self._cache_valid = False

    def predict_with_noise(self, control, n_v: np.ndarray, n_w: np.ndarray) -> None:
        u = _vec(control, 2, "particle filter control input")
        a = np.ascontiguousarray(n_v, dtype=np.float64)
        b = np.ascontiguousarray(n_w, dtype=np.float64)
        if a.size != self.particle_count() or b.size != self.particle_count():  # the C entry point copies n doubles from each
            raise RoboticsError.invalid_parameter("need one noise sample per particle")
        _check(self._L.rr_pf_predict_with_noise(self._h, _dp(u), _dp(a), _dp(b)))
        self._cache_valid = False

    def resample_with_uniforms(self, r: np.ndarray) -> None:
        a = np.ascontiguousarray(r, dtype=np.float64)
        _check(self._L.rr_pf_resample_with_uniforms(self._h, _dp(a), a.size))
        self._cache_valid = False

    def resample_systematic(self, rho: float) -> None:
        _check(self._L.rr_pf_resample_systematic(self._h, float(rho)))
        self._cache_valid = False

    def last_resample_indices(self) -> np.ndarray:
        n = self.particle_count()
        out = np.empty(n, dtype=np.uint32)
        _check(self._L.rr_pf_last_resample_indices(self._h, out.ctypes.data_as(C.POINTER(C.c_uint32)), n))
        return out

    def raw_weights(self) -> np.ndarray:
        out = np.empty(self.particle_count())
        _check(self._L.rr_pf_get_raw_weights(self._h, _dp(out)))
        return out

    def fixed_sums(self) -> _ffi.PfFixedSums:
        out = _ffi.PfFixedSums()
        _check(self._L.rr_pf_get_fixed_sums(self._h, C.byref(out)))
        return out

    def counters(self) -> Tuple[int, int]:
        a, b = C.c_uint32(), C.c_uint32()
        _check(self._L.rr_pf_get_counters(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def plan_stats(self) -> Tuple[int, bool]:
        """(launches of the one-launch resample plan that degraded to the serial plan because the device did not run
        all of their workgroups at once, whether the handle still uses the one-launch plan) -- ``rr_pf_plan_stats``"""
        g, e = C.c_uint64(), C.c_int32()
        _check(self._L.rr_pf_plan_stats(self._h, C.byref(g), C.byref(e)))
        return g.value, bool(e.value)

    # ---- measurement hooks
    def profile_enable(self, on) -> None:
        """False/0 off; True/1 HIP events around every launch (adds ~3 us per launch); 2 only the
        propagate+weight kernel, timed by the timestamps of its own dispatch (nothing extra in the stream)"""
        _check(self._L.rr_pf_profile_enable(self._h, int(on)))

    def profile_reset(self) -> None:
        _check(self._L.rr_pf_profile_reset(self._h))

    def profile_read(self) -> dict:
        out = {}
        for k in range(_ffi.RR_K_COUNT):
            n, ms = C.c_uint64(), C.c_double()
            _check(self._L.rr_pf_profile_read(self._h, k, C.byref(n), C.byref(ms)))
            out[self._L.rr_pf_kernel_name(k).decode()] = (n.value, ms.value)
        return out


# --------------------------------------------------------------------------------------------
MCLState = np.ndarray
MCLControl = np.ndarray
MCLMeasurement = PFMeasurement


@dataclass
class MonteCarloLocalizationConfig:
    """monte_carlo_localization.rs:50-83"""

    min_particles: int = 100
    max_particles: int = 5000
    kld_epsilon: float = 0.05
    kld_z: float = 2.326
    range_noise: float = 0.2
    velocity_noise: float = 2.0
    yaw_rate_noise: float = math.radians(40.0)
    dt: float = 0.1

    def validate(self) -> None:
        """monte_carlo_localization.rs:87-131 (same messages)"""
        inv = RoboticsError.invalid_parameter
        if self.min_particles <= 0:
            raise inv("MCL min_particles must be greater than zero")
        if self.max_particles < self.min_particles:
            raise inv("MCL max_particles must be greater than or equal to min_particles")
        if not math.isfinite(self.kld_epsilon) or self.kld_epsilon <= 0.0:
            raise inv("MCL kld_epsilon must be positive and finite")
        if not math.isfinite(self.kld_z) or self.kld_z <= 0.0:
            raise inv("MCL kld_z must be positive and finite")
        if not math.isfinite(self.range_noise) or self.range_noise <= 0.0:
            raise inv("MCL range_noise must be positive and finite")
        if not math.isfinite(self.velocity_noise) or self.velocity_noise < 0.0:
            raise inv("MCL velocity_noise must be non-negative and finite")
        if not math.isfinite(self.yaw_rate_noise) or self.yaw_rate_noise < 0.0:
            raise inv("MCL yaw_rate_noise must be non-negative and finite")
        if not math.isfinite(self.dt) or self.dt <= 0.0:
            raise inv("MCL dt must be positive and finite")


class MonteCarloLocalizer(ParticleFilterLocalizer):
    """monte_carlo_localization.rs:136-462: the same propagate/weight arithmetic as the PF
    (:209-288) and an unconditional resample every step (:298).  With min_particles ==
    max_particles the particle count is fixed (the BASELINE "MCL" configurations: fused
    three-launch step); otherwise the count adapts every step to the KLD bound (:322-385,
    ``rr_pf_create_adaptive``) and ``particle_count()`` follows it.
    """

    _GATE = _ffi.RR_GATE_ALWAYS
    _SCHEME = _ffi.RR_RESAMPLE_MULTINOMIAL

    def __init__(self, config: Optional[MonteCarloLocalizationConfig] = None, *, _initial_state=None, **kw):
        config = config or MonteCarloLocalizationConfig()
        config.validate()
        if _initial_state is not None and not np.all(np.isfinite(np.asarray(_initial_state, dtype=np.float64))):
            raise RoboticsError.invalid_parameter("MCL initial state must contain only finite values")
        pf_cfg = ParticleFilterConfig(n_particles=config.min_particles, resample_threshold=1.0,
                                      range_noise=config.range_noise, velocity_noise=config.velocity_noise,
                                      yaw_rate_noise=config.yaw_rate_noise, dt=config.dt)
        self.mcl_config = config
        if config.min_particles == config.max_particles:
            super().__init__(pf_cfg, _initial_state=_initial_state, **kw)
            return
        scheme = kw.pop("resample_scheme", None)
        if scheme not in (None, _ffi.RR_RESAMPLE_MULTINOMIAL):
            raise RoboticsError.invalid_parameter("the KLD-adaptive filter resamples multinomially (monte_carlo_localization.rs:343-355)")
        L = _ffi.lib()
        opt = _ffi.PfOptions()
        L.rr_pf_options_mcl(C.byref(opt))
        opt.device = kw.pop("device", 0)
        opt.seed = kw.pop("seed", 0)
        opt.likelihood_mode = kw.pop("likelihood_mode", _ffi.RR_LIK_FUSED)
        kw.pop("record_indices", None)
        if kw:
            raise TypeError(f"unexpected arguments for an adaptive MonteCarloLocalizer: {sorted(kw)}")
        kld = _ffi.MclAdaptive(config.min_particles, config.max_particles, config.kld_epsilon, config.kld_z)
        self._h = C.c_void_p()
        cfg = pf_cfg._c()
        st = None if _initial_state is None else _vec(_initial_state, 4, "MCL initial state")
        _check(L.rr_pf_create_adaptive(C.byref(cfg), C.byref(opt), C.byref(kld), _dp(st) if st is not None else None,
                                       C.byref(self._h)))
        self.config = pf_cfg
        self._L = L
        self._state_estimate = np.zeros(4)
        self._covariance = np.zeros((4, 4))
        self._cache_valid = False

    def is_adaptive(self) -> bool:
        return self.mcl_config.min_particles != self.mcl_config.max_particles

    def particle_capacity(self) -> int:
        return int(self._L.rr_pf_particle_capacity(self._h))

    # ---- parity seams of the adaptive filter
    def set_particles_array(self, aos: np.ndarray) -> None:
        if not self.is_adaptive():
            return super().set_particles_array(aos)
        a = np.ascontiguousarray(aos, dtype=np.float64).reshape(-1, 5)
        _check(self._L.rr_pf_set_particles_n(self._h, _dp(a), a.shape[0]))
        self._cache_valid = False

    def resample_adaptive_with_uniforms(self, r: np.ndarray) -> int:
        """resample_adaptive (:322-365) with the uniforms the reference would have drawn; returns the new count"""
        a = np.ascontiguousarray(r, dtype=np.float64)
        n_new = C.c_uint64()
        _check(self._L.rr_pf_resample_adaptive_with_uniforms(self._h, _dp(a), a.size, C.byref(n_new)))
        self._cache_valid = False
        return int(n_new.value)

    @classmethod
    def new(cls, config: MonteCarloLocalizationConfig, **kw) -> "MonteCarloLocalizer":
        return cls(config, **kw)

    try_new = new

    @classmethod
    def with_initial_state(cls, initial_state, config: MonteCarloLocalizationConfig, **kw):
        return cls(config, _initial_state=initial_state, **kw)

    try_with_initial_state = with_initial_state

    # monte_carlo_localization.rs:291-300
    def try_step(self, control, observations: MCLMeasurement) -> np.ndarray:
        return self.step(control, observations)

    # StateEstimator for MCL swallows errors (monte_carlo_localization.rs:455-462)
    def predict(self, control, _dt: float = 0.0) -> None:
        try:
            self.predict_with_control(control)
        except RoboticsError:
            pass

    def update(self, measurement: MCLMeasurement) -> None:
        try:
            self.update_with_observations(measurement)
        except RoboticsError:
            pass
        self.resample()
