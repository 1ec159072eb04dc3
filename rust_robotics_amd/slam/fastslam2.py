"""Host-side mirror of ``rust_robotics_slam::fastslam2`` over the HIP engine.

Reference: /root/reference/crates/rust_robotics_slam/src/fastslam2.rs -- ``create_particles``
(:425-429), ``fastslam2_update`` (:376-383; seedable body :331-374), ``get_best_particle``
(:385-390), ``get_observations`` (:419-423), ``Landmark`` (:32-52) / ``Particle`` (:54-82).

FastSLAM 2.0 draws each particle's pose from a proposal that fuses the motion prior with the
first observation of the step (compute_proposal :173-216, sample_pose :219-239); the landmark
update, weights and resampling are FastSLAM 1.0's with three changed constants.  The engine
object is therefore a ``FastSlam1`` created through ``rr_fs2_create``: every method of
``slam.fastslam1.FastSlam1`` (state access, seams, profiling, sharding) applies.

Nothing is computed in Python; there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence, Tuple

import numpy as np

from .. import _ffi
from . import fastslam1 as _f1
from .fastslam1 import Landmark, Particle, _check, _dp, _z_array, get_best_particle  # noqa: F401  (same pub structs)

# fastslam2.rs:18-30
DT = 0.1
MAX_RANGE = 20.0
N_PARTICLE = 100
NTH = N_PARTICLE / 1.5
Q_SIM = [[0.3, 0.0], [0.0, 0.0305]]
R_SIM = [[0.5, 0.0], [0.0, 0.0305]]
MOTION_COV = [[0.1, 0.0, 0.0], [0.0, 0.1, 0.0], [0.0, 0.0, 0.01]]


def default_params() -> _ffi.Fs2Params:
    p = _ffi.Fs2Params()
    _ffi.lib().rr_fs2_params_default(C.byref(p))
    return p


class FastSlam2(_f1.FastSlam1):
    """Device-resident FastSLAM 2.0 filter (engine extension; the reference has no struct)."""

    def __init__(self, n_particles: int, n_landmarks: int, *, params: Optional[_ffi.Fs2Params] = None, **kw):
        self.params2 = params or default_params()
        super().__init__(n_particles, n_landmarks, params=self.params2.base, **kw)

    def _create(self, n_particles: int, n_landmarks: int, opt) -> None:
        self.params2.base = self.params
        _check(self._L.rr_fs2_create(n_particles, n_landmarks, C.byref(self.params2), C.byref(opt), C.byref(self._h)))

    # ---- parity seams: the sampling step alone (fastslam2.rs:339-358)
    def propose_with_noise(self, u, z, noise: np.ndarray) -> None:
        u = np.ascontiguousarray(u, dtype=np.float64)
        za = _z_array(z)
        nz = np.ascontiguousarray(noise, dtype=np.float64).reshape(self.n, 3)
        _check(self._L.rr_fs2_predict_with_noise(self._h, _dp(u), _dp(za) if za.size else None, za.shape[0], _dp(nz)))

    def propose(self, u, z) -> None:
        u = np.ascontiguousarray(u, dtype=np.float64)
        za = _z_array(z)
        _check(self._L.rr_fs2_predict(self._h, _dp(u), _dp(za) if za.size else None, za.shape[0]))


class ShardedFastSlam2(_f1.ShardedFastSlam1):
    """One shard of a FastSLAM 2.0 filter over the GPUs of a node (see ShardedFastSlam1)."""

    def __init__(self, rank: int, world: int, n_local: int, n_landmarks: int, *, params: Optional[_ffi.Fs2Params] = None, **kw):
        self.params2 = params or default_params()
        super().__init__(rank, world, n_local, n_landmarks, params=self.params2.base, **kw)

    _create = FastSlam2._create


def create_particles(n_particles: int, n_landmarks: int) -> List[Particle]:
    """fastslam2.rs:425-429"""
    return _f1.create_particles(n_particles, n_landmarks)


def fastslam2_update(particles: List[Particle], u, z: Sequence[Tuple[float, float, int]], *, seed: int = 0, device: int = 0) -> None:
    """fastslam2.rs:376-383 on the GPU: upload the caller's particles, one update, download."""
    _f1.fastslam_update(particles, u, z, seed=seed, device=device, _engine=FastSlam2)


def get_observations(x_true, landmarks: Sequence[Tuple[float, float]], *, seed: int = 0, step: int = 0):
    """fastslam2.rs:392-423 (same simulator as FastSLAM 1.0) with the engine's seedable noise stream."""
    return _f1.get_observations(x_true, landmarks, seed=seed, step=step)
