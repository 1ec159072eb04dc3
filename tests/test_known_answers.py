"""Pins oracle/ref_literal.c (and the D-spec) with the hand-derivable known
answers KA1-KA7 of SURVEY.md Appendix B -- arithmetic a reviewer can redo by
hand following the reference lines cited there.  rtol 1e-12 (libm last-ulp)."""
import ctypes as C
import math

import numpy as np

import oracle
from oracle import dp, u32p, u64p

RT = 1e-12


def close(a, b, rt=RT):
    return abs(a - b) <= rt * max(1.0, abs(b))


def test_KA1_gauss_likelihood(ref):
    assert close(ref.ref_gauss_likelihood(0.0, 0.2), 1.9947114020071635)
    assert close(ref.ref_gauss_likelihood(0.3, 0.2), 0.6475879783294588)


def test_KA2_predict_zero_noise(ref, det):
    exp = (1.0877582561890373, 2.04794255386042, 0.51, 1.0)
    for which in ("ref", "det"):
        x, y, yaw, v = (np.array([a]) for a in (1.0, 2.0, 0.5, 0.0))
        if which == "ref":
            ref.ref_pf_predict(1, dp(x), dp(y), dp(yaw), dp(v), 1.0, 0.1, 0.1, None, None)
        else:
            z = np.zeros(1)
            det.det_pf_predict(1, dp(x), dp(y), dp(yaw), dp(v), 1.0, 0.1, 0.1, dp(z), dp(z), 0, 0, 0, 0.0, 0.0)
        for got, e in zip((x[0], y[0], yaw[0], v[0]), exp):
            assert close(got, e), (which, got, e)


def test_KA3_update_normalize_neff(ref, det):
    x = np.array([0.0, 1.0])
    y = np.array([0.0, 0.0])
    obs = np.array([5.0, 3.0, 4.0, 4.0, 0.0, 4.0])
    w = np.empty(2)
    ref.ref_pf_update_raw(2, dp(x), dp(y), dp(w), dp(obs), 2, 0.2)
    assert close(w[0], 3.978873577297384) and close(w[1], 0.10112030778506782)
    for mode in (0, 1):
        wd = np.empty(2)
        det.det_pf_weights(2, dp(x), dp(y), dp(wd), dp(obs), 2, 0.2, mode)
        assert close(wd[0], 3.978873577297384) and close(wd[1], 0.10112030778506782)
    ref.ref_pf_normalize(2, dp(w))
    assert close(w[0], 0.9752155736912276) and close(w[1], 0.02478442630877235)
    assert close(ref.ref_pf_neff(2, dp(w)), 1.050795802226709)
    # D-spec N_eff from the exact integer sums
    sh, tot, q2h, q2l = C.c_int(), C.c_uint64(), C.c_uint64(), C.c_uint64()
    assert det.det_fix_reduce(2, dp(wd), float(wd.max()), 2, C.byref(sh), C.byref(tot), C.byref(q2h), C.byref(q2l)) == 1
    assert close(det.det_fix_neff(tot, q2h, q2l), 1.050795802226709, 1e-10)
    assert close(det.det_fix_total_to_double(tot, sh), wd.sum(), 1e-12)


def test_KA4_fastslam_ekf(ref, det):
    m = oracle.ref_fs1_model()
    e = np.array([5.0, 0.0, 0.5, 0.0, 0.0, 0.5])
    w = C.c_double(1.0)
    ref.ref_fs1_update_landmark(0.0, 0.0, 0.0, C.byref(w), 5.2, 0.05, dp(e), C.byref(m))
    assert close(e[0], 5.1) and close(e[1], 0.09900990099009901)
    assert close(e[2], 0.25) and close(e[5], 0.30198019801980197) and e[3] == 0.0 and e[4] == 0.0
    assert close(w.value, 0.6772339004082811)
    # D-spec, landmark-major planes with n = 1
    md = oracle.det_fs1_model()
    maps = np.array([5.0, 0.0, 0.5, 0.0, 0.0, 0.5])
    pw = np.array([1.0])
    z = np.array([5.2, 0.05, 0.0])
    zero = np.zeros(1)
    det.det_fs1_observe(1, dp(zero), dp(zero.copy()), dp(zero.copy()), dp(pw), dp(maps), dp(z), 1, C.byref(md), 1)
    assert close(maps[0], 5.1) and close(maps[1], 0.09900990099009901)
    assert close(maps[2], 0.25) and close(maps[5], 0.30198019801980197)
    assert close(pw[0], 0.6772339004082811)


def test_KA4b_fastslam_ekf_near_singular_innovation_covariance(ref, det):
    """det(S) positive but below 2^-1000 (fastslam1.rs:161-181 with a tiny R and a collapsed landmark covariance): the literal
    likelihood exp(..)/(2 pi sqrt(det)) is finite; the D-spec's reciprocal form 1/det would overflow to inf there (weight inf ->
    maximum inf -> the degenerate image for the whole filter), so below 2^-1000 it takes the literal quotient
    (include/rr_pf_spec.h, rr_fs1_update_one).  Just above the guard the reciprocal form must agree as well."""
    for r, guarded in ((1e-160, True), (1e-140, False)):
        m = oracle.ref_fs1_model()
        m.r00, m.r11 = r, r
        e = np.array([5.0, 0.0, 0.0, 0.0, 0.0, 0.0])  # covariance 0: S = R
        w = C.c_double(1.0)
        ref.ref_fs1_update_landmark(0.0, 0.0, 0.0, C.byref(w), 5.0, 0.0, dp(e), C.byref(m))  # innovation exactly 0
        assert (r * r < 2.0 ** -1000) == guarded
        assert np.isfinite(w.value) and close(w.value, 1.0 / (2.0 * np.pi * r), 1e-4)  # (det = r^2 is subnormal in the first case: few bits)
        md = oracle.det_fs1_model()
        md.r00, md.r11 = r, r
        maps = np.array([5.0, 0.0, 0.0, 0.0, 0.0, 0.0])
        pw = np.array([1.0])
        zero = np.zeros(1)
        det.det_fs1_observe(1, dp(zero), dp(zero.copy()), dp(zero.copy()), dp(pw), dp(maps), dp(np.array([5.0, 0.0, 0.0])), 1, C.byref(md), 1)
        assert np.isfinite(pw[0]) and close(pw[0], w.value, 1e-9)
        assert close(maps[0], e[0]) and close(maps[1], e[1]) and close(maps[2], e[2]) and close(maps[5], e[5])


def test_KA5_systematic(ref, det):
    w = np.array([0.1, 0.2, 0.3, 0.4])
    idx = np.empty(4, np.uint32)
    ref.ref_fs1_resample_indices(4, dp(w.copy()), 0.12, u32p(idx))
    assert idx.tolist() == [1, 2, 3, 3]
    # D-spec: rho = r0 * n
    sh, tot, q2h, q2l = C.c_int(), C.c_uint64(), C.c_uint64(), C.c_uint64()
    det.det_fix_reduce(4, dp(w), 0.4, 4, C.byref(sh), C.byref(tot), C.byref(q2h), C.byref(q2l))
    cdf = np.empty(4, np.uint64)
    det.det_fix_cdf(4, dp(w), 1, sh, 0, u64p(cdf))
    det.det_indices_systematic(4, u64p(cdf), tot, 4, 0, 4, 0.48, u32p(idx))
    assert idx.tolist() == [1, 2, 3, 3]


def test_KA6_multinomial(ref, det):
    w = np.array([0.1, 0.2, 0.3, 0.4])
    r = np.array([0.05, 0.3, 0.95, 0.61])
    idx = np.empty(4, np.uint32)
    for fn in (ref.ref_pf_resample_indices, ref.ref_pf_resample_indices_bsearch, ref.ref_mcl_resample_indices):
        fn(4, dp(w), dp(r), u32p(idx))
        assert idx.tolist() == [0, 1, 3, 3]
    sh, tot, q2h, q2l = C.c_int(), C.c_uint64(), C.c_uint64(), C.c_uint64()
    det.det_fix_reduce(4, dp(w), 0.4, 4, C.byref(sh), C.byref(tot), C.byref(q2h), C.byref(q2l))
    cdf = np.empty(4, np.uint64)
    det.det_fix_cdf(4, dp(w), 1, sh, 0, u64p(cdf))
    # r = 0.3 sits exactly on the boundary c[1]: the float cumsum 0.30000000000000004
    # selects 1; the integer CDF (floor-quantised) sits a hair below 0.3*T, so the
    # D-spec also resolves the tie-free neighbours identically and documents this one.
    r2 = np.array([0.05, 0.29999, 0.95, 0.61])
    det.det_indices_multinomial(4, u64p(cdf), tot, 0, 4, dp(r2), 0, 0, u32p(idx))
    assert idx.tolist() == [0, 1, 3, 3]


def test_KA6_default_index_quirks(ref):
    """Q7: PF falls back to index 0, MCL forces the last cum to 1.0 / falls back to last."""
    w = np.array([0.25, 0.25, 0.25, 0.2499999])  # cumsum ends below 1
    r = np.full(4, 0.99999999)  # (one draw per output: n of them -- found by `make -C oracle asan`, the test used to pass one)
    idx = np.empty(4, np.uint32)
    ref.ref_pf_resample_indices(4, dp(w), dp(r), u32p(idx))
    assert idx[0] == 0
    ref.ref_pf_resample_indices_bsearch(4, dp(w), dp(r), u32p(idx))
    assert idx[0] == 0
    ref.ref_mcl_resample_indices(4, dp(w), dp(r), u32p(idx))
    assert idx[0] == 3


def test_KA7_underflow_bound(ref):
    coeff64 = ref.ref_gauss_likelihood(0.0, 0.2) ** 64
    assert close(coeff64, 1.557e19, 1e-3)
    # product of 64 equal factors reaches 0 when sum diff^2 >~ 63.09
    for ss, zero in ((62.0, False), (64.5, True)):
        d = math.sqrt(ss / 64)
        x = np.array([0.0])
        y = np.array([0.0])
        obs = np.tile(np.array([10.0 + d, 10.0, 0.0]), 64)
        w = np.empty(1)
        ref.ref_pf_update_raw(1, dp(x), dp(y), dp(w), dp(obs), 64, 0.2)
        assert (w[0] == 0.0) == zero, (ss, w[0])


def test_a_single_shard_serves_all_slots(tmp_path):
    """finalize_plan (csrc/resample_core.hpp) does not evaluate rr_sys_slots_upto_exact for a shard that holds everything: it
    writes served range [0, n).  tests/c/served_range.c checks that identity against the exact function on the host -- edge cases
    up to T = 2^64 - 1, n = 2^31 - 1 and three million random plans (include/rr_pf_spec.h: the same code the kernels run)."""
    import os
    import shutil
    import subprocess

    import pytest

    if not shutil.which("gcc"):
        pytest.skip("no gcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "served_range")
    r = subprocess.run(["gcc", "-std=c11", "-O2", "-Wall", "-Werror", "-I", os.path.join(root, "include"), os.path.join(root, "tests", "c", "served_range.c"),
                        "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "SERVED_RANGE_OK" in r.stdout, r.stdout[-1500:]
