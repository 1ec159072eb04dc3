"""CPU oracles -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this package; the product (``rust_robotics_amd``) never does.

* ``ref``  -> ``oracle/ref_literal.c``: line-by-line C restatement of the
  reference's Rust CPU path (libm, serial sums, no FMA).
* ``det``  -> ``oracle/det_spec.c``: serial CPU evaluation of the engine's
  deterministic spec (``include/rr_pf_spec.h``); the HIP kernels must match it
  bit for bit.

Both are built by ``oracle/Makefile`` into ``oracle/_build`` (``build()`` below
or ``__graft_entry__.build()``); prebuilt ``.so`` files travel to the GPU box.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from functools import lru_cache

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_BUILD = os.environ.get("RR_ORACLE_DIR") or os.path.join(_HERE, "_build")  # RR_ORACLE_DIR: the sanitizer build (make -C oracle asan)

c_double_p = C.POINTER(C.c_double)
c_u32_p = C.POINTER(C.c_uint32)
c_u64_p = C.POINTER(C.c_uint64)


def build(force: bool = False) -> None:
    """Compile both oracles with ``make`` (gcc, a second or two)."""
    args = ["make", "-C", _HERE]
    if force:
        args.append("-B")
    subprocess.run(args, check=True, capture_output=True)


def _load(name: str) -> C.CDLL:
    path = os.path.join(_BUILD, name)
    if not os.path.exists(path):
        build()
    return C.CDLL(path)


def dp(a: np.ndarray):
    assert a.dtype == np.float64 and a.flags["C_CONTIGUOUS"], (a.dtype, a.flags)
    return a.ctypes.data_as(c_double_p)


def u32p(a: np.ndarray):
    assert a.dtype == np.uint32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(c_u32_p)


def u64p(a: np.ndarray):
    assert a.dtype == np.uint64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(c_u64_p)


class RefFs1Model(C.Structure):
    _fields_ = [(k, C.c_double) for k in ("dt", "q00", "q11", "r00", "r11", "init_threshold", "init_cov")]


class DetFs1Model(C.Structure):
    """mirrors rr_fs1_model (include/rr_pf_spec.h)"""

    _fields_ = [(k, C.c_double) for k in ("dt", "q_sqrt0", "q_sqrt1", "r00", "r11", "init_threshold", "init_cov",
                                          "init_test_lt", "nonpos_det_w")]


class DetFs2Model(C.Structure):
    """mirrors rr_fs2_model (include/rr_pf_spec.h)"""

    _fields_ = [("base", DetFs1Model), ("m0", C.c_double), ("m1", C.c_double), ("m2", C.c_double)]


@lru_cache(maxsize=None)
def ref() -> C.CDLL:
    L = _load("libref_literal.so")
    d, sz, u32, i = C.c_double, C.c_size_t, c_u32_p, C.c_int
    P = c_double_p
    L.ref_gauss_likelihood.restype = d
    L.ref_gauss_likelihood.argtypes = [d, d]
    L.ref_set_threads.argtypes = [i]
    L.ref_set_threads.restype = i
    L.ref_get_threads.restype = i
    L.ref_pf_predict.argtypes = [sz, P, P, P, P, d, d, d, P, P]
    L.ref_pf_predict.restype = None
    L.ref_pf_update_raw.argtypes = [sz, P, P, P, P, sz, d]
    L.ref_pf_update_raw.restype = None
    L.ref_pf_normalize.argtypes = [sz, P]
    L.ref_pf_normalize.restype = d
    L.ref_pf_neff.argtypes = [sz, P]
    L.ref_pf_neff.restype = d
    L.ref_pf_estimate.argtypes = [sz, P, P, P, P, P, P]
    L.ref_pf_estimate.restype = None
    L.ref_pf_covariance.argtypes = [sz, P, P, P, P, P, P, P]
    L.ref_pf_covariance.restype = None
    for f in (L.ref_pf_resample_indices, L.ref_pf_resample_indices_bsearch, L.ref_mcl_resample_indices):
        f.argtypes = [sz, P, P, u32]
        f.restype = None
    L.ref_kld_required.argtypes = [sz, sz, sz, d, d]
    L.ref_kld_required.restype = sz
    L.ref_mcl_resample_adaptive.argtypes = [sz, P, P, P, P, P, sz, sz, d, d, u32]
    L.ref_mcl_resample_adaptive.restype = sz
    L.ref_pf_gather.argtypes = [sz, P, P, P, P, P, u32]
    L.ref_pf_gather.restype = None
    L.ref_pf_step.argtypes = [sz, P, P, P, P, P, d, d, d, P, P, P, sz, d, d, i, P, u32, P]
    L.ref_pf_step.restype = i
    L.ref_pf_step_ex.argtypes = [sz, P, P, P, P, P, d, d, d, P, P, P, sz, d, d, i, P, u32, P, i]
    L.ref_pf_step_ex.restype = i
    L.ref_pf_try_step_loop.argtypes = [sz, P, P, P, P, P, P, d, P, P, P, sz, d, d, i, P, u32, sz, P, i, i]
    L.ref_pf_try_step_loop.restype = d
    MP = C.POINTER(RefFs1Model)
    L.ref_fs1_model_default.argtypes = [MP]
    L.ref_fs1_model_default.restype = None
    L.ref_normalize_angle.argtypes = [d]
    L.ref_normalize_angle.restype = d
    L.ref_fs1_create.argtypes = [sz, sz, P, P, P, P, P]
    L.ref_fs1_create.restype = None
    L.ref_fs1_predict.argtypes = [sz, P, P, P, d, d, P, P, MP]
    L.ref_fs1_predict.restype = None
    L.ref_fs1_update_landmark.argtypes = [d, d, d, P, d, d, P, MP]
    L.ref_fs1_update_landmark.restype = None
    L.ref_fs1_normalize.argtypes = [sz, P]
    L.ref_fs1_normalize.restype = d
    L.ref_fs1_neff.argtypes = [sz, P]
    L.ref_fs1_neff.restype = d
    L.ref_fs1_resample_indices.argtypes = [sz, P, d, u32]
    L.ref_fs1_resample_indices.restype = None
    L.ref_fs1_gather.argtypes = [sz, sz, P, P, P, P, P, u32]
    L.ref_fs1_gather.restype = None
    L.ref_fs1_observe.argtypes = [sz, sz, P, P, P, P, P, P, sz, MP]
    L.ref_fs1_observe.restype = None
    L.ref_fs1_update.argtypes = [sz, sz, P, P, P, P, P, d, d, P, P, P, sz, MP, d, d, u32]
    L.ref_fs1_update.restype = i
    L.ref_fs1_best_particle.argtypes = [sz, P]
    L.ref_fs1_best_particle.restype = sz
    L.ref_fs1_get_observations.argtypes = [P, P, sz, d, P, MP, P]
    L.ref_fs1_get_observations.restype = sz
    L.ref_fs2_proposal.argtypes = [P, d, d, d, d, P, d, d, P, P]
    L.ref_fs2_proposal.restype = None
    L.ref_fs2_sample.argtypes = [P, P, P, P]
    L.ref_fs2_sample.restype = None
    L.ref_fs2_update_landmark.argtypes = [d, d, d, d, d, P, d, d]
    L.ref_fs2_update_landmark.restype = d
    L.ref_fs2_update.argtypes = [sz, sz, P, P, P, P, P, d, d, P, P, sz, d, d, u32]
    L.ref_fs2_update.restype = i
    return L


@lru_cache(maxsize=None)
def det() -> C.CDLL:
    L = _load("libdet_spec.so")
    d, sz, u32, i = C.c_double, C.c_size_t, c_u32_p, C.c_int
    u64, u32v = C.c_uint64, C.c_uint32
    P = c_double_p
    for name in ("det_exp_v", "det_log_v", "det_sqrt_v"):
        f = getattr(L, name)
        f.argtypes = [sz, P, P]
        f.restype = None
    for name in ("det_sincos_v", "det_sincos2pi_v", "det_atan2_v", "det_div_v", "det_fma_v"):
        f = getattr(L, name)
        f.argtypes = [sz, P, P, P]
        f.restype = None
    for name in ("det_uniform2_v", "det_normal2_v"):
        f = getattr(L, name)
        f.argtypes = [u64, u32v, u32v, u64, sz, P, P]
        f.restype = None
    L.det_targets_multinomial.argtypes = [u64, sz, sz, u64, u32v, c_u64_p]
    L.det_targets_multinomial.restype = None
    L.det_philox_raw.argtypes = [u32v] * 6 + [c_u32_p]
    L.det_philox_raw.restype = None
    L.det_philox_raw_n.argtypes = [u32v] * 6 + [i, c_u32_p]
    L.det_philox_raw_n.restype = None
    L.det_philox_rounds.restype = i
    L.det_pf_init.argtypes = [sz, u64, u64, P, P, P, P, P]
    L.det_pf_init.restype = None
    L.det_pf_predict.argtypes = [sz, P, P, P, P, d, d, d, P, P, u64, u32v, u64, d, d]
    L.det_pf_predict.restype = None
    L.det_pf_weights.argtypes = [sz, P, P, P, P, sz, d, i]
    L.det_pf_weights.restype = None
    L.det_wmax.argtypes = [sz, P]
    L.det_wmax.restype = d
    L.det_fix_reduce.argtypes = [sz, P, d, u64, C.POINTER(i), c_u64_p, c_u64_p, c_u64_p]
    L.det_fix_reduce.restype = i
    L.det_fix_cdf.argtypes = [sz, P, i, i, u64, c_u64_p]
    L.det_fix_cdf.restype = None
    L.det_fix_total_to_double.argtypes = [u64, i]
    L.det_fix_total_to_double.restype = d
    L.det_fix_neff.argtypes = [u64, u64, u64]
    L.det_fix_neff.restype = d
    L.det_indices_multinomial.argtypes = [sz, c_u64_p, u64, sz, sz, P, u64, u32v, u32]
    L.det_indices_multinomial.restype = None
    L.det_kld_required.argtypes = [u64, u64, u64, d, d]
    L.det_kld_required.restype = u64
    L.det_mcl_resample_adaptive.argtypes = [sz, P, P, P, c_u64_p, u64, P, u64, u32v, u64, u64, d, d, u32]
    L.det_mcl_resample_adaptive.restype = sz
    L.det_resample_rho.argtypes = [u64, u32v]
    L.det_resample_rho.restype = d
    L.det_indices_systematic.argtypes = [sz, c_u64_p, u64, u64, sz, sz, d, u32]
    L.det_indices_systematic.restype = None
    L.det_pf_moments.argtypes = [sz, P, P, P, P, P, d, P, P]
    L.det_pf_moments.restype = None
    L.det_pf_step.argtypes = [sz, P, P, P, P, P, d, d, d, d, d, P, sz, d, i, d, i, i, u64, u32v, u32v, u32, P]
    L.det_pf_step.restype = i
    MP = C.POINTER(DetFs1Model)
    L.det_fs1_model_default.argtypes = [MP]
    L.det_fs1_model_default.restype = None
    L.det_fs1_predict.argtypes = [sz, P, P, P, d, d, P, P, u64, u32v, u64, MP]
    L.det_fs1_predict.restype = None
    L.det_fs1_observe.argtypes = [sz, P, P, P, P, P, P, sz, MP, i]
    L.det_fs1_observe.restype = None
    L.det_fs1_update.argtypes = [sz, sz, P, P, P, P, P, d, d, P, sz, MP, d, u64, u32v, u32v, i, u32]
    L.det_fs1_get_observations.argtypes = [P, P, sz, d, d, d, u64, u32v, P]
    L.det_fs1_get_observations.restype = sz
    L.det_fs1_update.restype = i
    M2 = C.POINTER(DetFs2Model)
    L.det_fs2_model_default.argtypes = [M2]
    L.det_fs2_model_default.restype = None
    L.det_fs2_proposal.argtypes = [P, d, d, d, d, P, M2, P, P]
    L.det_fs2_proposal.restype = None
    L.det_fs2_sample.argtypes = [P, P, P, P]
    L.det_fs2_sample.restype = None
    L.det_fs2_predict.argtypes = [sz, P, P, P, P, d, d, P, sz, P, u64, u32v, u64, M2]
    L.det_fs2_predict.restype = None
    L.det_fs2_update.argtypes = [sz, sz, P, P, P, P, P, d, d, P, sz, M2, P, d, u64, u32v, u32v, i, u32]
    L.det_fs2_update.restype = i
    L.det_fs1_best_particle.argtypes = [sz, P]
    L.det_fs1_best_particle.restype = sz
    return L


def ref_fs1_model() -> RefFs1Model:
    m = RefFs1Model()
    ref().ref_fs1_model_default(C.byref(m))
    return m


def det_fs1_model() -> DetFs1Model:
    m = DetFs1Model()
    det().det_fs1_model_default(C.byref(m))
    return m


def det_fs2_model() -> DetFs2Model:
    m = DetFs2Model()
    det().det_fs2_model_default(C.byref(m))
    return m


def maps_aos_to_planes(lm: np.ndarray, n: int, L: int) -> np.ndarray:
    """[p][l][6] (reference order) -> [l][6][p] (device / D-spec order)."""
    return np.ascontiguousarray(lm.reshape(n, L, 6).transpose(1, 2, 0)).reshape(-1)


def maps_planes_to_aos(pl: np.ndarray, n: int, L: int) -> np.ndarray:
    return np.ascontiguousarray(pl.reshape(L, 6, n).transpose(2, 0, 1)).reshape(-1)
