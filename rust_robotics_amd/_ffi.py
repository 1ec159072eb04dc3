"""ctypes binding of the engine's C ABI (include/rr_pf.h, include/rr_fastslam1.h).

This is the Python twin of the Rust ``-sys`` crate shown in INTEGRATION.md: it
declares exactly the entry points the headers declare and nothing else.  There is
no fallback path: if ``librust_robotics_amd.so`` is missing the import fails
loudly (build it with ``python -c "import __graft_entry__ as g; g.build()"`` or
``make -C rust_robotics_amd/csrc``).
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# RR_AMD_LIBRARY: development override (A/B timing of two builds of the same ABI); never a fallback
LIB_PATH = os.environ.get("RR_AMD_LIBRARY") or os.path.join(_HERE, "librust_robotics_amd.so")

RR_OK, RR_INVALID_PARAMETER, RR_RUNTIME_ERROR = 0, 1, 2
RR_RESAMPLE_MULTINOMIAL, RR_RESAMPLE_SYSTEMATIC = 0, 1
RR_GATE_NEFF, RR_GATE_ALWAYS = 0, 1
RR_LIK_FUSED, RR_LIK_PRODUCT = 0, 1
RR_K_PROPAGATE_WEIGHT, RR_K_QUANTIZE_REDUCE, RR_K_SCAN_TILES, RR_K_CDF = 0, 1, 2, 3
RR_K_RESAMPLE_GATHER, RR_K_COMMIT, RR_K_MOMENTS, RR_K_COUNT = 4, 5, 6, 7


class PfConfig(C.Structure):
    """rr_pf_config == ParticleFilterConfig (particle_filter.rs:51-65)"""

    _fields_ = [
        ("n_particles", C.c_uint64),
        ("resample_threshold", C.c_double),
        ("range_noise", C.c_double),
        ("velocity_noise", C.c_double),
        ("yaw_rate_noise", C.c_double),
        ("dt", C.c_double),
    ]


class PfOptions(C.Structure):
    _fields_ = [
        ("device", C.c_int32),
        ("resample_scheme", C.c_int32),
        ("resample_gate", C.c_int32),
        ("likelihood_mode", C.c_int32),
        ("seed", C.c_uint64),
        ("first_global_index", C.c_uint64),
        ("n_global", C.c_uint64),
        ("record_indices", C.c_int32),
        ("reserved", C.c_int32),
    ]


class PfFixedSums(C.Structure):
    _fields_ = [
        ("usable", C.c_int32),
        ("shift", C.c_int32),
        ("total", C.c_uint64),
        ("q2_hi", C.c_uint64),
        ("q2_lo", C.c_uint64),
        ("w_max", C.c_double),
        ("sum", C.c_double),
    ]


class MclAdaptive(C.Structure):
    """rr_mcl_adaptive (include/rr_pf.h)"""

    _fields_ = [("min_particles", C.c_uint64), ("max_particles", C.c_uint64), ("kld_epsilon", C.c_double),
                ("kld_z", C.c_double)]


class Fs1Params(C.Structure):
    """rr_fs1_params: the constants of fastslam1.rs:13-23 as fields"""

    _fields_ = [(k, C.c_double) for k in ("dt", "q00", "q11", "r00", "r11", "max_range", "nth", "initial_weight",
                                          "init_cov", "init_threshold", "first_obs_cov")]


class Fs2Params(C.Structure):
    """rr_fs2_params (include/rr_fastslam2.h)"""

    _fields_ = [("base", Fs1Params), ("motion_cov", C.c_double * 3), ("nonpos_det_weight", C.c_double)]


class Fs1Options(C.Structure):
    _fields_ = [("device", C.c_int32), ("record_indices", C.c_int32), ("seed", C.c_uint64), ("obs_chunks", C.c_int32),
                ("reserved", C.c_int32), ("first_global_index", C.c_uint64), ("n_global", C.c_uint64)]


RR_FK_COUNT = 10
RR_P2P_HANDLE_BYTES = 256  # include/rr_pf.h


class PfShardSums(C.Structure):
    _fields_ = [("total", C.c_uint64), ("q2_hi", C.c_uint64), ("q2_lo", C.c_uint64)]


class PfShardPlan(C.Structure):
    _fields_ = [
        ("fired", C.c_int32),
        ("usable", C.c_int32),
        ("total_global", C.c_uint64),
        ("base", C.c_uint64),
        ("total_local", C.c_uint64),
        ("rho", C.c_double),
    ]


_lib = None


def lib() -> C.CDLL:
    """Load the shared library once and declare every prototype of include/rr_pf.h."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: the HIP engine has not been built and there is no CPU fallback. "
            "Run `make -C rust_robotics_amd/csrc` (hipcc --offload-arch=gfx950)."
        )
    L = C.CDLL(LIB_PATH)
    d, i32, u32, u64, sz = C.c_double, C.c_int32, C.c_uint32, C.c_uint64, C.c_size_t
    P = C.POINTER(d)
    H = C.c_void_p
    st = C.c_int

    def proto(name, res, args):
        try:
            f = getattr(L, name)
        except AttributeError:
            if os.environ.get("RR_AMD_LIBRARY"):  # an OLDER build loaded for an A/B run (tools/ab_bench.sh): entry points added since
                return                            # are simply absent -- calling one raises AttributeError at the call site
            raise
        f.restype = res
        f.argtypes = args

    proto("rr_last_error", C.c_char_p, [])
    proto("rr_version", C.c_char_p, [])
    proto("rr_device_count", C.c_int, [])
    proto("rr_device_pci_bus_id", st, [C.c_int32, C.c_char_p, sz])
    proto("rr_pf_config_default", None, [C.POINTER(PfConfig)])
    proto("rr_pf_config_validate", st, [C.POINTER(PfConfig)])
    proto("rr_pf_options_default", None, [C.POINTER(PfOptions)])
    proto("rr_pf_options_mcl", None, [C.POINTER(PfOptions)])
    proto("rr_pf_create", st, [C.POINTER(PfConfig), C.POINTER(PfOptions), C.POINTER(H)])
    proto("rr_pf_create_with_state", st, [C.POINTER(PfConfig), C.POINTER(PfOptions), P, C.POINTER(H)])
    proto("rr_pf_destroy", None, [H])
    proto("rr_pf_set_landmarks", st, [H, P, sz])
    proto("rr_pf_landmark_count", sz, [H])
    proto("rr_pf_get_landmarks", sz, [H, P, sz])
    proto("rr_pf_set_range_noise", st, [H, d])
    proto("rr_pf_predict", st, [H, P])
    proto("rr_pf_update", st, [H, P, sz])
    proto("rr_pf_resample", st, [H])
    proto("rr_pf_step", st, [H, P, P, sz, P])
    proto("rr_pf_step_async", st, [H, P, P, sz])
    proto("rr_pf_step_async_estimate", st, [H, P, P, sz])
    proto("rr_pf_last_step_estimate", st, [H, P])
    proto("rr_pf_step_many", st, [H, P, P, sz, sz, P])
    proto("rr_pf_set_resident", st, [H, d])
    proto("rr_pf_resident_stats", st, [H, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)])
    proto("rr_pf_synchronize", st, [H])
    proto("rr_pf_warm", st, [H, d])
    proto("rr_pf_estimate", st, [H, P])
    proto("rr_pf_covariance", st, [H, P])
    proto("rr_pf_particle_count", u64, [H])
    proto("rr_pf_get_particles", st, [H, P])
    proto("rr_pf_n_eff", st, [H, P])
    proto("rr_pf_last_resample_fired", st, [H, C.POINTER(i32)])
    proto("rr_pf_set_particles", st, [H, P])
    proto("rr_pf_predict_with_noise", st, [H, P, P, P])
    proto("rr_pf_resample_with_uniforms", st, [H, P, sz])
    proto("rr_pf_resample_systematic", st, [H, d])
    proto("rr_pf_last_resample_indices", st, [H, C.POINTER(u32), sz])
    proto("rr_pf_get_raw_weights", st, [H, P])
    proto("rr_pf_get_fixed_sums", st, [H, C.POINTER(PfFixedSums)])
    proto("rr_pf_get_counters", st, [H, C.POINTER(u32), C.POINTER(u32)])
    proto("rr_pf_plan_stats", st, [H, C.POINTER(u64), C.POINTER(i32)])
    proto("rr_pf_profile_enable", st, [H, i32])
    proto("rr_pf_profile_read", st, [H, i32, C.POINTER(u64), P])
    proto("rr_pf_profile_reset", st, [H])
    proto("rr_pf_kernel_name", C.c_char_p, [i32])
    proto("rr_selftest_math", st, [i32, i32, sz, P, P, P, P])
    V = C.c_void_p  # device pointers
    proto("rr_pf_set_stream", st, [H, V])
    proto("rr_pf_shard_propagate_weight", st, [H, P, P, sz, V])
    proto("rr_pf_shard_quantize", st, [H, V, V])
    proto("rr_pf_shard_cdf", st, [H, V, i32, i32])
    proto("rr_pf_shard_get_plan", st, [H, C.POINTER(PfShardPlan)])
    proto("rr_pf_shard_gather_slots", st, [H, u64, u64, V])
    proto("rr_pf_shard_adopt", st, [H, V])
    proto("rr_pf_shard_select", st, [H, i32, V])
    proto("rr_pf_shard_pack_selected", st, [H, i32, V])
    proto("rr_pf_shard_adopt_records", st, [H, V, u64])
    proto("rr_sys_first_slot_above", u64, [d, u64, u64, u64])
    U8 = C.POINTER(C.c_uint8)
    proto("rr_comm_unique_id", st, [U8])
    proto("rr_comm_create", st, [U8, i32, i32, i32, C.POINTER(H)])
    proto("rr_comm_destroy", None, [H])
    proto("rr_comm_create_local", st, [i32, i32, i32, C.POINTER(H)])
    proto("rr_pf_shard_step_local", st, [C.POINTER(H), C.POINTER(H), i32, P, P, sz])
    proto("rr_pf_shard_step", st, [H, H, P, P, sz])
    proto("rr_pf_shard_estimate", st, [H, H, P, P])
    proto("rr_pf_shard_last_migrated", u64, [H])
    proto("rr_pf_p2p_export", st, [H, U8])
    proto("rr_pf_p2p_connect", st, [H, U8, i32, i32])
    proto("rr_pf_p2p_connect_local", st, [C.POINTER(H), i32])
    KP = C.POINTER(MclAdaptive)
    proto("rr_mcl_adaptive_default", None, [KP])
    proto("rr_mcl_adaptive_validate", st, [KP])
    proto("rr_pf_create_adaptive", st, [C.POINTER(PfConfig), C.POINTER(PfOptions), KP, P, C.POINTER(H)])
    proto("rr_pf_particle_capacity", u64, [H])
    proto("rr_pf_set_particles_n", st, [H, P, u64])
    proto("rr_pf_resample_adaptive_with_uniforms", st, [H, P, sz, C.POINTER(u64)])
    proto("rr_pf_shard_step_p2p", st, [H, P, P, sz])
    proto("rr_pf_shard_step_p2p_unfused", st, [H, P, P, sz])
    proto("rr_pf_p2p_status", st, [H, C.POINTER(i32)])
    proto("rr_pf_p2p_topology", st, [H, C.POINTER(i32)])
    proto("rr_pf_shard_want_estimate", st, [H, i32])
    proto("rr_pf_shard_last_estimate_sums", st, [H, P, P])
    proto("rr_sys_segment_matrix", u64, [d, C.POINTER(u64), i32, u64, u64, i32, C.POINTER(C.c_int64)])
    FP, FO = C.POINTER(Fs1Params), C.POINTER(Fs1Options)
    proto("rr_fs1_params_default", None, [FP])
    proto("rr_fs1_options_default", None, [FO])
    proto("rr_fs1_create", st, [u64, u64, FP, FO, C.POINTER(H)])
    F2 = C.POINTER(Fs2Params)
    proto("rr_fs2_params_default", None, [F2])
    proto("rr_fs2_create", st, [u64, u64, F2, FO, C.POINTER(H)])
    proto("rr_fs2_update", st, [H, P, P, sz])
    proto("rr_fs2_update_async", st, [H, P, P, sz])
    proto("rr_fs2_predict_with_noise", st, [H, P, P, sz, P])
    proto("rr_fs2_predict", st, [H, P, P, sz])
    proto("rr_fs1_destroy", None, [H])
    proto("rr_fs1_particle_count", u64, [H])
    proto("rr_fs1_landmark_count", u64, [H])
    proto("rr_fs1_update", st, [H, P, P, sz])
    proto("rr_fs1_update_async", st, [H, P, P, sz])
    proto("rr_fs1_synchronize", st, [H])
    proto("rr_fs1_warm", st, [H, d])
    proto("rr_fs1_best_particle", st, [H, P, P, C.POINTER(u64)])
    proto("rr_fs1_set_resident", st, [H, d])
    proto("rr_fs1_resident_stats", st, [H, C.POINTER(u64), C.POINTER(u64)])
    proto("rr_fs1_get_landmarks", st, [H, u64, P])
    proto("rr_fs1_get_poses", st, [H, P])
    proto("rr_fs1_get_state", st, [H, P, P])
    proto("rr_fs1_set_state", st, [H, P, P])
    proto("rr_fs1_update_host", st, [H, P, P, P, P, sz])
    proto("rr_fs1_get_observations", sz, [P, P, sz, FP, u64, u32, P, sz])
    proto("rr_fs1_predict_with_noise", st, [H, P, P, P])
    proto("rr_fs1_predict", st, [H, P])
    proto("rr_fs1_observe", st, [H, P, sz])
    proto("rr_fs1_normalize_resample", st, [H])
    proto("rr_fs1_resample_systematic", st, [H, d])
    proto("rr_fs1_last_resample_fired", st, [H, C.POINTER(i32)])
    proto("rr_fs1_last_resample_indices", st, [H, C.POINTER(u32), sz])
    proto("rr_fs1_n_eff", st, [H, P])
    proto("rr_fs1_get_fixed_sums", st, [H, C.POINTER(PfFixedSums)])
    proto("rr_fs1_get_counters", st, [H, C.POINTER(u32), C.POINTER(u32), C.POINTER(i32)])
    proto("rr_fs1_plan_stats", st, [H, C.POINTER(u64), C.POINTER(i32)])
    proto("rr_fs1_observe_stats", st, [H, C.POINTER(u64), C.POINTER(i32)])
    proto("rr_fs1_p2p_export", st, [H, U8])
    proto("rr_fs1_p2p_connect", st, [H, U8, i32, i32])
    proto("rr_fs1_p2p_connect_local", st, [C.POINTER(H), i32])
    proto("rr_fs1_shard_update_p2p", st, [H, P, P, sz])
    proto("rr_fs1_p2p_status", st, [H, C.POINTER(i32)])
    proto("rr_fs1_shard_update", st, [H, H, P, P, sz])
    proto("rr_fs1_shard_last_migrated", u64, [H])
    proto("rr_fs1_shard_local", st, [H, P, P, sz, V])
    proto("rr_fs1_shard_quantize", st, [H, V, V])
    proto("rr_fs1_shard_plan", st, [H, V, i32, i32])
    proto("rr_fs1_shard_get_plan", st, [H, C.POINTER(PfShardPlan)])
    proto("rr_fs1_shard_pack", st, [H, C.POINTER(C.c_int64), i32, i32, V])
    proto("rr_fs1_shard_unpack", st, [H, C.POINTER(C.c_int64), i32, i32, V])
    proto("rr_fs1_profile_enable", st, [H, i32])
    proto("rr_fs1_profile_read", st, [H, i32, C.POINTER(u64), P])
    proto("rr_fs1_profile_reset", st, [H])
    proto("rr_fs1_kernel_name", C.c_char_p, [i32])
    _lib = L
    return L


def last_error() -> str:
    return lib().rr_last_error().decode()
