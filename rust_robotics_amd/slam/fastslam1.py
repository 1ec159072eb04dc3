"""Host-side mirror of ``rust_robotics_slam::fastslam1`` over the HIP engine.

Reference: /root/reference/crates/rust_robotics_slam/src/fastslam1.rs.  The reference API is
free functions over a caller-owned ``Vec<Particle>`` -- ``create_particles`` (:302-306),
``fastslam_update`` (:237-266), ``get_best_particle`` (:269-274), ``get_observations`` (:277-299)
-- and the ``pub`` structs ``Landmark`` (:26-31) / ``Particle`` (:44-51).  Both shapes exist here:

* ``FastSlam1``: the particle set and all maps stay on the GPU (what a node would hold);
* the free functions keep the reference's signatures; ``fastslam_update(particles, u, z)``
  mutates the caller's list through the upload -> step -> download shim ``rr_fs1_update_host``
  and is meant for example-sized problems (the reference examples use 60 x 6).

Nothing is computed in Python; there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import math
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

import numpy as np

from .. import _ffi
from ..core import RoboticsError

# fastslam1.rs:13-23
DT = 0.1
MAX_RANGE = 20.0
N_PARTICLE = 100
NTH = N_PARTICLE / 1.5
Q_SIM = [[0.3, 0.0], [0.0, 0.0305]]
R_SIM = [[0.5, 0.0], [0.0, 0.0305]]

_DP = C.POINTER(C.c_double)


def _dp(a: np.ndarray):
    return a.ctypes.data_as(_DP)


def _check(status: int) -> None:
    if status == _ffi.RR_OK:
        return
    msg = _ffi.last_error()
    raise (RoboticsError.invalid_parameter if status == _ffi.RR_INVALID_PARAMETER else RoboticsError.runtime)(msg)


@dataclass
class Landmark:
    """fastslam1.rs:26-41"""

    x: float = 0.0
    y: float = 0.0
    cov: np.ndarray = field(default_factory=lambda: np.eye(2) * 1000.0)


@dataclass
class Particle:
    """fastslam1.rs:44-67"""

    weight: float
    x: float
    y: float
    yaw: float
    landmarks: List[Landmark]

    def pose(self) -> np.ndarray:
        return np.array([self.x, self.y, self.yaw])


def default_params() -> _ffi.Fs1Params:
    p = _ffi.Fs1Params()
    _ffi.lib().rr_fs1_params_default(C.byref(p))
    return p


def _z_array(z: Sequence[Tuple[float, float, int]]) -> np.ndarray:
    if len(z) == 0:
        return np.zeros((0, 3))
    return np.ascontiguousarray(np.asarray(z, dtype=np.float64).reshape(-1, 3))


class FastSlam1:
    """Device-resident FastSLAM 1.0 filter (engine extension; the reference has no struct)."""

    def __init__(self, n_particles: int, n_landmarks: int, *, params: Optional[_ffi.Fs1Params] = None, seed: int = 0,
                 device: int = 0, obs_chunks: int = 0, first_global_index: int = 0, n_global: int = 0):
        L = _ffi.lib()
        self._L = L
        self.params = params or default_params()
        opt = _ffi.Fs1Options()
        L.rr_fs1_options_default(C.byref(opt))
        opt.device, opt.seed, opt.obs_chunks = device, seed, obs_chunks
        opt.first_global_index, opt.n_global = int(first_global_index), int(n_global)
        self._h = C.c_void_p()
        self._create(int(n_particles), int(n_landmarks), opt)
        self.n, self.L = int(n_particles), int(n_landmarks)

    def _create(self, n_particles: int, n_landmarks: int, opt) -> None:
        _check(self._L.rr_fs1_create(n_particles, n_landmarks, C.byref(self.params), C.byref(opt), C.byref(self._h)))

    def close(self) -> None:
        h = getattr(self, "_h", None)
        if h:
            self._L.rr_fs1_destroy(h)
            self._h = None

    def __del__(self):
        self.close()

    # ---- the reference's step
    def update(self, u, z) -> None:
        u = np.ascontiguousarray(u, dtype=np.float64)
        za = _z_array(z)
        _check(self._L.rr_fs1_update(self._h, _dp(u), _dp(za) if za.size else None, za.shape[0]))

    def update_async(self, u, z) -> None:
        u = np.ascontiguousarray(u, dtype=np.float64)
        za = _z_array(z)
        _check(self._L.rr_fs1_update_async(self._h, _dp(u), _dp(za) if za.size else None, za.shape[0]))

    def synchronize(self) -> None:
        _check(self._L.rr_fs1_synchronize(self._h))

    def warm(self, ms: float = 0.0) -> None:
        """rr_fs1_warm: `ms` milliseconds (0: 50) of step-shaped work on the filter's stream before the first update"""
        _check(self._L.rr_fs1_warm(self._h, float(ms)))

    def set_resident(self, idle_us: float) -> None:
        """Resident service (``rr_fs1_set_resident``, engine extension): with ``idle_us > 0`` the updates of a FastSLAM 1.0 filter
        of up to 1024 particles are served by ONE kernel that stays on the device and answers each of them with the best
        particle (``best_particle()`` right after an update is then free); 0 switches it off."""
        _check(self._L.rr_fs1_set_resident(self._h, float(idle_us)))

    def resident_stats(self) -> Tuple[int, int]:
        a, b = C.c_uint64(0), C.c_uint64(0)
        _check(self._L.rr_fs1_resident_stats(self._h, C.byref(a), C.byref(b)))
        return int(a.value), int(b.value)

    def best_particle(self) -> Tuple[np.ndarray, float, int]:
        pose = np.empty(3)
        w = C.c_double()
        i = C.c_uint64()
        _check(self._L.rr_fs1_best_particle(self._h, _dp(pose), C.byref(w), C.byref(i)))
        return pose, w.value, int(i.value)

    def landmarks_of(self, particle_index: int) -> np.ndarray:
        out = np.empty((self.L, 6))
        _check(self._L.rr_fs1_get_landmarks(self._h, int(particle_index), _dp(out)))
        return out

    def poses(self) -> np.ndarray:
        out = np.empty((self.n, 4))
        _check(self._L.rr_fs1_get_poses(self._h, _dp(out)))
        return out

    def get_state(self) -> Tuple[np.ndarray, np.ndarray]:
        poses = np.empty((self.n, 4))
        maps = np.empty((self.n, self.L, 6))
        _check(self._L.rr_fs1_get_state(self._h, _dp(poses), _dp(maps) if maps.size else None))
        return poses, maps

    def set_state(self, poses: Optional[np.ndarray], maps: Optional[np.ndarray]) -> None:
        p = None if poses is None else np.ascontiguousarray(poses, dtype=np.float64).reshape(self.n, 4)
        m = None if maps is None else np.ascontiguousarray(maps, dtype=np.float64).reshape(self.n, self.L, 6)
        _check(self._L.rr_fs1_set_state(self._h, _dp(p) if p is not None else None, _dp(m) if m is not None and m.size else None))

    # ---- parity seams
    def predict(self, u) -> None:
        u = np.ascontiguousarray(u, dtype=np.float64)
        _check(self._L.rr_fs1_predict(self._h, _dp(u)))

    def predict_with_noise(self, u, z0, z1) -> None:
        u = np.ascontiguousarray(u, dtype=np.float64)
        a, b = np.ascontiguousarray(z0, dtype=np.float64), np.ascontiguousarray(z1, dtype=np.float64)
        if a.size != self.n or b.size != self.n:  # the C entry point copies n doubles from each
            raise RoboticsError.invalid_parameter("need one noise sample per particle")
        _check(self._L.rr_fs1_predict_with_noise(self._h, _dp(u), _dp(a), _dp(b)))

    def observe(self, z) -> None:
        za = _z_array(z)
        _check(self._L.rr_fs1_observe(self._h, _dp(za) if za.size else None, za.shape[0]))

    def normalize_resample(self) -> None:
        _check(self._L.rr_fs1_normalize_resample(self._h))

    def resample_systematic(self, rho: float) -> None:
        _check(self._L.rr_fs1_resample_systematic(self._h, float(rho)))

    def last_resample_fired(self) -> bool:
        out = C.c_int32()
        _check(self._L.rr_fs1_last_resample_fired(self._h, C.byref(out)))
        return bool(out.value)

    def last_resample_indices(self) -> np.ndarray:
        out = np.empty(self.n, dtype=np.uint32)
        _check(self._L.rr_fs1_last_resample_indices(self._h, out.ctypes.data_as(C.POINTER(C.c_uint32)), self.n))
        return out

    def n_eff(self) -> float:
        out = C.c_double()
        _check(self._L.rr_fs1_n_eff(self._h, C.byref(out)))
        return out.value

    def fixed_sums(self) -> _ffi.PfFixedSums:
        out = _ffi.PfFixedSums()
        _check(self._L.rr_fs1_get_fixed_sums(self._h, C.byref(out)))
        return out

    def counters(self) -> Tuple[int, int, int]:
        a, b, c = C.c_uint32(), C.c_uint32(), C.c_int32()
        _check(self._L.rr_fs1_get_counters(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def plan_stats(self) -> Tuple[int, bool]:
        """(launches of the one-launch resample plan that degraded to the serial plan, whether the handle still uses
        the one-launch plan) -- ``rr_fs1_plan_stats``"""
        g, e = C.c_uint64(), C.c_int32()
        _check(self._L.rr_fs1_plan_stats(self._h, C.byref(g), C.byref(e)))
        return g.value, bool(e.value)

    def observe_stats(self) -> Tuple[int, bool]:
        """(host looks at the device that found a closing workgroup of the observation kernel had given up waiting for a chunk's
        weight factor -- the follow-up kernel formed those weights, same bits --, whether the handle still waits inside the
        kernel) -- ``rr_fs1_observe_stats``"""
        g, e = C.c_uint64(), C.c_int32()
        _check(self._L.rr_fs1_observe_stats(self._h, C.byref(g), C.byref(e)))
        return g.value, bool(e.value)

    # ---- measurement hooks
    def profile_enable(self, on) -> None:
        """False/0 off; True/1 HIP events around every launch; 2 only k_fs1_observe, timed by the
        timestamps of its own dispatch (nothing extra in the stream)"""
        _check(self._L.rr_fs1_profile_enable(self._h, int(on)))

    def profile_reset(self) -> None:
        _check(self._L.rr_fs1_profile_reset(self._h))

    def profile_read(self) -> dict:
        out = {}
        for k in range(_ffi.RR_FK_COUNT):
            n, ms = C.c_uint64(), C.c_double()
            _check(self._L.rr_fs1_profile_read(self._h, k, C.byref(n), C.byref(ms)))
            out[self._L.rr_fs1_kernel_name(k).decode()] = (n.value, ms.value)
        return out


# ---------------------------------------------------------------------------------------------
# the reference's free functions

class ShardedFastSlam1(FastSlam1):
    """One shard of a FastSLAM 1.0 filter spread over the GPUs of a node:
    particles ``[rank * n_local, (rank + 1) * n_local)`` with their whole maps.  The step
    (``update``) runs over the peer-to-peer transport of include/rr_pf.h -- no host code and no
    collective library inside it -- and gives the bits the unsharded filter gives.

    ``connect_ipc(allgather)`` is for one process per GPU (``allgather`` returns every rank's
    handle bytes in rank order, e.g. ``sharded.gloo_allgather(dist)``); ``link_local(shards)``
    is for several shards inside one process."""

    def __init__(self, rank: int, world: int, n_local: int, n_landmarks: int, *, device: int = 0, **kw):
        super().__init__(n_local, n_landmarks, device=device, first_global_index=rank * n_local, n_global=world * n_local,
                         **kw)
        self.rank, self.world, self.device = rank, world, device
        self._comm = None  # RCCL communicator (connect_rccl); None => the peer-to-peer transport

    def connect_rccl(self, exchange) -> None:
        """Use the RCCL transport (``rr_fs1_shard_update``: all-reduce MAX, all-gather of the integer sums,
        grouped send/recv of whole particles).  ``exchange(bytes) -> bytes`` broadcasts rank 0's value
        (e.g. ``sharded.gloo_exchange(dist)``); every rank calls this together."""
        uid = (C.c_uint8 * 128)()
        if self.rank == 0:
            _check(self._L.rr_comm_unique_id(uid))
        raw = exchange(bytes(uid))
        uid = (C.c_uint8 * 128).from_buffer_copy(raw)
        comm = C.c_void_p()
        _check(self._L.rr_comm_create(uid, self.rank, self.world, self.device, C.byref(comm)))
        self._comm = comm

    def close(self) -> None:
        if getattr(self, "_comm", None):
            if getattr(self, "_h", None):
                self._L.rr_fs1_synchronize(self._h)
            self._L.rr_comm_destroy(self._comm)
            self._comm = None
        super().close()

    def migrated(self) -> int:
        return int(self._L.rr_fs1_shard_last_migrated(self._h))

    def connect_ipc(self, allgather) -> None:
        blob = (C.c_uint8 * _ffi.RR_P2P_HANDLE_BYTES)()
        _check(self._L.rr_fs1_p2p_export(self._h, blob))
        parts = allgather(bytes(blob))
        allb = (C.c_uint8 * (_ffi.RR_P2P_HANDLE_BYTES * self.world)).from_buffer_copy(b"".join(parts))
        _check(self._L.rr_fs1_p2p_connect(self._h, allb, self.world, self.rank))

    @staticmethod
    def link_local(shards: Sequence["ShardedFastSlam1"]) -> None:
        arr = (C.c_void_p * len(shards))(*[s._h for s in shards])
        _check(_ffi.lib().rr_fs1_p2p_connect_local(arr, len(shards)))

    def update_async(self, u, z) -> None:
        u = np.ascontiguousarray(u, dtype=np.float64)
        za = _z_array(z)
        if self._comm:
            _check(self._L.rr_fs1_shard_update(self._h, self._comm, _dp(u), _dp(za) if za.size else None, za.shape[0]))
        else:
            _check(self._L.rr_fs1_shard_update_p2p(self._h, _dp(u), _dp(za) if za.size else None, za.shape[0]))

    def update(self, u, z) -> None:
        self.update_async(u, z)
        self.synchronize()

    def timed_out(self) -> bool:
        if self._comm:
            return False  # RCCL has no bounded waits of its own to report
        out = C.c_int32()
        _check(self._L.rr_fs1_p2p_status(self._h, C.byref(out)))
        return bool(out.value)


def create_particles(n_particles: int, n_landmarks: int) -> List[Particle]:
    """fastslam1.rs:302-306"""
    return [Particle(1.0 / N_PARTICLE, 0.0, 0.0, 0.0, [Landmark() for _ in range(n_landmarks)]) for _ in range(n_particles)]


def _pack(particles: List[Particle]) -> Tuple[np.ndarray, np.ndarray]:
    n = len(particles)
    L = len(particles[0].landmarks) if n else 0
    poses = np.empty((n, 4))
    maps = np.empty((n, L, 6))
    for i, p in enumerate(particles):
        poses[i] = (p.weight, p.x, p.y, p.yaw)
        for l, lm in enumerate(p.landmarks):
            c = np.asarray(lm.cov, dtype=np.float64)
            maps[i, l] = (lm.x, lm.y, c[0, 0], c[1, 0], c[0, 1], c[1, 1])  # column-major Matrix2
    return poses, maps


def _unpack(particles: List[Particle], poses: np.ndarray, maps: np.ndarray) -> None:
    for i, p in enumerate(particles):
        p.weight, p.x, p.y, p.yaw = (float(v) for v in poses[i])
        for l, lm in enumerate(p.landmarks):
            e = maps[i, l]
            lm.x, lm.y = float(e[0]), float(e[1])
            lm.cov = np.array([[e[2], e[4]], [e[3], e[5]]])


_SHIM_CACHE = {}


def fastslam_update(particles: List[Particle], u, z: Sequence[Tuple[float, float, int]], *, seed: int = 0,
                    device: int = 0, _engine=None) -> None:
    """fastslam1.rs:237-266 on the GPU: upload the caller's particles, one update, download."""
    if not particles:
        return
    n, L = len(particles), len(particles[0].landmarks)
    engine = _engine or FastSlam1
    key = (engine.__name__, n, L, seed, device)
    fs = _SHIM_CACHE.get(key)
    if fs is None:
        fs = _SHIM_CACHE[key] = engine(n, L, seed=seed, device=device)
    poses, maps = _pack(particles)
    u = np.ascontiguousarray(u, dtype=np.float64)
    za = _z_array(z)
    _check(fs._L.rr_fs1_update_host(fs._h, _dp(poses), _dp(maps) if maps.size else None, _dp(u), _dp(za) if za.size else None,
                                    za.shape[0]))
    _unpack(particles, poses, maps)


def get_best_particle(particles: List[Particle]) -> Particle:
    """fastslam1.rs:269-274 (ties -> last).  Host-side list helper for the shim shape."""
    best = particles[0]
    for p in particles[1:]:
        if not (p.weight < best.weight):
            best = p
    return best


def get_observations(x_true, landmarks: Sequence[Tuple[float, float]], *, seed: int = 0, step: int = 0,
                     params: Optional[_ffi.Fs1Params] = None) -> List[Tuple[float, float, int]]:
    """fastslam1.rs:277-299 with the engine's seedable noise stream."""
    L = _ffi.lib()
    xt = np.ascontiguousarray(x_true, dtype=np.float64)
    lms = np.ascontiguousarray(np.asarray(landmarks, dtype=np.float64).reshape(-1, 2))
    out = np.empty((max(len(lms), 1), 3))
    prm = params or default_params()
    cnt = L.rr_fs1_get_observations(_dp(xt), _dp(lms) if lms.size else None, len(lms), C.byref(prm), seed, step, _dp(out), len(lms))
    return [(float(out[k, 0]), float(out[k, 1]), int(out[k, 2])) for k in range(cnt)]
