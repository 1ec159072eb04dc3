"""KLD-adaptive resampling (monte_carlo_localization.rs:322-385) on the CPU: the literal
restatement (float cumsum, linear scan, one draw at a time) against the D-spec (integer CDF) on
identical uniforms, the bound of :367-378 against hand-computed values, and the reference's own
unit test :519-552 re-expressed against the restatement."""
import ctypes as C
import math

import numpy as np
import pytest

import oracle
from oracle import dp, u32p, u64p
from tests import helpers as H


def clouds():
    rng = np.random.default_rng(5)
    n = 700
    yield "spread", rng.uniform(-6, 6, n), rng.uniform(-6, 6, n), rng.uniform(-3, 3, n), rng.random(n) + 0.05
    yield "tight", 1.0 + rng.normal(0, 0.05, n), 1.0 + rng.normal(0, 0.05, n), 0.1 + rng.normal(0, 0.01, n), rng.random(n) + 0.05
    w = np.zeros(n)
    w[[3, 77, 500]] = [0.2, 0.5, 0.3]
    yield "three survivors", rng.uniform(-6, 6, n), rng.uniform(-6, 6, n), rng.uniform(-3, 3, n), w
    yield "on bin edges", np.round(rng.uniform(-4, 4, n) * 2) / 2, np.round(rng.uniform(-4, 4, n) * 2) / 2, np.zeros(n), rng.random(n) + 0.05


def det_adaptive(det, x, y, yaw, w, r, lo, hi, eps=0.05, z=2.326):
    n = x.size
    fx = H.det_fixed(det, w)
    cdf = H.det_cdf(det, w, fx)
    idx = np.empty(hi, np.uint32)
    cnt = det.det_mcl_resample_adaptive(n, dp(x), dp(y), dp(yaw), u64p(cdf), int(cdf[-1]), dp(r), 0, 0, lo, hi, eps, z, u32p(idx))
    return cnt, idx[:cnt]


def test_required_particles_formula(det, ref):
    # k <= 1 -> min; Wilson-Hilferty value by hand for k = 5, eps 0.05, z 2.326
    for f in (det.det_kld_required, ref.ref_kld_required):
        assert f(0, 100, 5000, 0.05, 2.326) == 100 and f(1, 100, 5000, 0.05, 2.326) == 100
        a = 2.0 / (9.0 * 4.0)
        want = math.ceil(4.0 / 0.1 * (1.0 - a + 2.326 * math.sqrt(a)) ** 3)
        assert f(5, 10, 5000, 0.05, 2.326) == want
        assert f(10**6, 100, 5000, 0.05, 2.326) == 5000  # clamped
    for k in range(1, 4000, 7):
        assert det.det_kld_required(k, 100, 5000, 0.05, 2.326) == ref.ref_kld_required(k, 100, 5000, 0.05, 2.326)


@pytest.mark.parametrize("lo,hi", [(100, 1500), (1, 40), (300, 300), (50, 5000)])
def test_literal_and_det_agree(det, ref, lo, hi):
    rng = np.random.default_rng(8)
    for name, x, y, yaw, w in clouds():
        x, y, yaw = (np.ascontiguousarray(a) for a in (x, y, yaw))
        wn = np.ascontiguousarray(w / w.sum())
        r = np.floor(rng.random(hi) * 2**53) / 2**53
        idx_l = np.empty(hi, np.uint32)
        cnt_l = ref.ref_mcl_resample_adaptive(x.size, dp(x), dp(y), dp(yaw), dp(wn), dp(r), lo, hi, 0.05, 2.326, u32p(idx_l))
        cnt_d, idx_d = det_adaptive(det, x, y, yaw, wn, r, lo, hi)
        assert lo <= cnt_l <= hi
        assert cnt_d == cnt_l, name
        assert np.array_equal(idx_d, idx_l[:cnt_l]), name


def test_reference_unit_test_particle_count_adapts(ref):
    """monte_carlo_localization.rs:519-552: a multimodal cloud needs more particles than the
    minimum; a concentrated one needs no more than the multimodal one"""
    lo, hi = 100, 1500
    rng = np.random.default_rng(2)
    n = 800
    i = np.arange(n)
    x = (i % 4) * 3.0 + i * 0.002
    y = (i % 4) * 2.0
    z = np.zeros(n)
    w = np.full(n, 1.0 / n)
    idx = np.empty(hi, np.uint32)
    expanded = ref.ref_mcl_resample_adaptive(n, dp(np.ascontiguousarray(x)), dp(np.ascontiguousarray(y)), dp(z), dp(w),
                                             dp(rng.random(hi)), lo, hi, 0.05, 2.326, u32p(idx))
    assert expanded > lo
    n = 600
    reduced = ref.ref_mcl_resample_adaptive(n, dp(np.full(n, 1.0)), dp(np.full(n, 1.0)), dp(np.full(n, 0.1)), dp(np.full(n, 1.0 / n)),
                                            dp(rng.random(hi)), lo, hi, 0.05, 2.326, u32p(idx))
    assert reduced <= expanded and reduced == lo
