"""include/rr_detmath.h against mpmath (the contract's accuracy claim) and its
structural properties (Philox known-answer vectors, uniform ranges)."""
import ctypes as C
import math

import mpmath as mp
import numpy as np
import pytest

import oracle
from oracle import dp

mp.mp.dps = 40


def ulp_err(got, exact_fn, xs):
    worst = 0.0
    for g, x in zip(got, xs):
        e = exact_fn(x)
        ef = float(e)
        if ef == 0.0 or not np.isfinite(ef):
            continue
        u = np.spacing(abs(ef))
        worst = max(worst, abs(float((mp.mpf(float(g)) - e) / mp.mpf(float(u)))))
    return worst


def test_exp_accuracy(det):
    rng = np.random.default_rng(1)
    x = np.concatenate([rng.uniform(-745, 709, 3000), rng.uniform(-2, 2, 3000), [0.0, -0.0, 1.0, -1.0]])
    o = np.empty_like(x)
    det.det_exp_v(x.size, dp(x), dp(o))
    # skip the subnormal tail where one extra rounding is by design
    keep = x > -708
    assert ulp_err(o[keep], lambda v: mp.exp(mp.mpf(float(v))), x[keep]) <= 1.5
    assert np.all(np.abs(o[~keep] - np.exp(x[~keep])) <= 2 * np.spacing(np.exp(x[~keep])) + 5e-324)


def test_exp_edges(det):
    x = np.array([710.0, 1e308, -746.0, -1e308, np.nan, np.inf, -np.inf])
    o = np.empty_like(x)
    det.det_exp_v(x.size, dp(x), dp(o))
    assert o[0] == np.inf and o[1] == np.inf and o[2] == 0.0 and o[3] == 0.0
    assert np.isnan(o[4]) and o[5] == np.inf and o[6] == 0.0


def test_log_accuracy(det):
    rng = np.random.default_rng(2)
    x = np.concatenate([rng.uniform(0, 1, 4000), 10.0 ** rng.uniform(-300, 300, 2000), [1.0, 0.5, 2.0, 5e-324]])
    x = x[x > 0]
    o = np.empty_like(x)
    det.det_log_v(x.size, dp(x), dp(o))
    assert ulp_err(o, lambda v: mp.log(mp.mpf(float(v))), x) <= 2.5
    z = np.array([0.0, -1.0, np.inf])
    oz = np.empty_like(z)
    det.det_log_v(3, dp(z), dp(oz))
    assert oz[0] == -np.inf and np.isnan(oz[1]) and oz[2] == np.inf


def test_sincos_accuracy(det):
    rng = np.random.default_rng(3)
    x = np.concatenate([rng.uniform(-10, 10, 4000), rng.uniform(-1e4, 1e4, 3000), [0.0, 1e-300, -1e-20]])
    s = np.empty_like(x)
    c = np.empty_like(x)
    det.det_sincos_v(x.size, dp(x), dp(s), dp(c))
    # absolute error is what the pose update sees; <= 1 ulp of 1.0 everywhere,
    # and <= 2 ulp relative away from the zeros
    for got, fn in ((s, mp.sin), (c, mp.cos)):
        ex = np.array([float(fn(mp.mpf(float(v)))) for v in x])
        assert np.max(np.abs(got - ex)) <= 2.3e-16
        far = np.abs(ex) > 1e-3
        assert ulp_err(got[far], lambda v: fn(mp.mpf(float(v))), x[far]) <= 2.0


def test_sincos_nonfinite_and_huge(det):
    x = np.array([np.inf, -np.inf, np.nan, 1e300, 2.0**40 + 0.5])
    s = np.empty_like(x)
    c = np.empty_like(x)
    det.det_sincos_v(x.size, dp(x), dp(s), dp(c))
    assert np.all(np.isnan(s[:3])) and np.all(np.isnan(c[:3]))
    assert np.all(np.abs(s[3:]) <= 1.0) and np.all(np.abs(c[3:]) <= 1.0)


def test_sincos2pi(det):
    rng = np.random.default_rng(4)
    u = np.floor(rng.uniform(0, 1, 5000) * 2**53) / 2**53
    u = np.concatenate([u, [0.0, 0.25, 0.5, 0.75, 1 - 2**-53, 0.125, 0.375]])
    s = np.empty_like(u)
    c = np.empty_like(u)
    det.det_sincos2pi_v(u.size, dp(u), dp(s), dp(c))
    es = np.array([float(mp.sin(2 * mp.pi * mp.mpf(float(v)))) for v in u])
    ec = np.array([float(mp.cos(2 * mp.pi * mp.mpf(float(v)))) for v in u])
    assert np.max(np.abs(s - es)) <= 5e-16 and np.max(np.abs(c - ec)) <= 5e-16
    assert np.max(np.abs(s * s + c * c - 1)) <= 5e-16


def test_atan2_accuracy(det):
    rng = np.random.default_rng(5)
    y = rng.normal(size=6000) * 10.0 ** rng.uniform(-3, 3, 6000)
    x = rng.normal(size=6000) * 10.0 ** rng.uniform(-3, 3, 6000)
    o = np.empty_like(x)
    det.det_atan2_v(x.size, dp(y), dp(x), dp(o))
    exact = lambda pair: mp.atan2(mp.mpf(float(pair[0])), mp.mpf(float(pair[1])))
    assert ulp_err(o, exact, list(zip(y, x))) <= 2.5


def test_atan2_special(det):
    y = np.array([0.0, -0.0, 0.0, -0.0, 1.0, -1.0, np.inf, -np.inf, np.inf, 1.0, 1.0, np.nan])
    x = np.array([1.0, 1.0, -1.0, -1.0, 0.0, 0.0, np.inf, np.inf, -np.inf, np.inf, -np.inf, 1.0])
    o = np.empty_like(x)
    det.det_atan2_v(x.size, dp(y), dp(x), dp(o))
    e = np.arctan2(y, x)
    assert np.isnan(o[-1])
    np.testing.assert_allclose(o[:-1], e[:-1], rtol=0, atol=4.5e-16)
    assert np.array_equal(np.signbit(o[:-1]), np.signbit(e[:-1]))


def test_sqrt_div_are_ieee(det):
    rng = np.random.default_rng(6)
    a = 10.0 ** rng.uniform(-200, 200, 20000)
    b = 10.0 ** rng.uniform(-100, 100, 20000)
    o = np.empty_like(a)
    det.det_sqrt_v(a.size, dp(a), dp(o))
    assert np.array_equal(o, np.sqrt(a))
    det.det_div_v(a.size, dp(a), dp(b), dp(o))
    assert np.array_equal(o, a / b)


def test_philox_known_answers(det):
    """Random123 kat_vectors for philox4x32-10."""
    out = (C.c_uint32 * 4)()
    det.det_philox_raw(0, 0, 0, 0, 0, 0, out)
    assert [hex(v) for v in out] == ["0x6627e8d5", "0xe169c58d", "0xbc57ac4c", "0x9b00dbd8"]
    det.det_philox_raw(0xFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFF, out)
    assert [hex(v) for v in out] == ["0x408f276d", "0x41c83b0e", "0xa20bc7c6", "0x6d5451fd"]
    det.det_philox_raw(0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344, 0xA4093822, 0x299F31D0, out)
    assert [hex(v) for v in out] == ["0xd16cfe09", "0x94fdcceb", "0x5001e420", "0x24126ea1"]


def test_philox_seven_round_known_answers(det):
    """The engine's streams run Philox4x32-7 (RR_PHILOX_ROUNDS): pinned by the Random123 kat_vectors of the 7-round form, and
    by an independent evaluation of the published round function in plain Python integers."""
    assert det.det_philox_rounds() == 7
    out = (C.c_uint32 * 4)()
    kat = [((0, 0, 0, 0), (0, 0), ["0x5f6fb709", "0xd893f64", "0x4f121f81", "0x4f730a48"]),
           ((0xFFFFFFFF,) * 4, (0xFFFFFFFF,) * 2, ["0x5207ddc2", "0x45165e59", "0x4d8ee751", "0x8c52f662"]),
           ((0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344), (0xA4093822, 0x299F31D0), ["0x4dfccaba", "0x190a87f0", "0xc47362ba", "0xb6b5242a"])]
    for c, k, want in kat:
        det.det_philox_raw_n(*c, *k, 7, out)
        assert [hex(v) for v in out] == want
        det.det_philox_raw_n(*c, *k, 0, out)  # the engine's own form
        assert [hex(v) for v in out] == want

    def philox_py(c, k, rounds):
        c, k = list(c), list(k)
        for _ in range(rounds):
            p0, p1 = 0xD2511F53 * c[0], 0xCD9E8D57 * c[2]
            c = [(p1 >> 32) ^ c[1] ^ k[0], p1 & 0xFFFFFFFF, (p0 >> 32) ^ c[3] ^ k[1], p0 & 0xFFFFFFFF]
            k = [(k[0] + 0x9E3779B9) & 0xFFFFFFFF, (k[1] + 0xBB67AE85) & 0xFFFFFFFF]
        return c

    rng = np.random.default_rng(5)
    for _ in range(200):
        c = [int(v) for v in rng.integers(0, 2**32, 4)]
        k = [int(v) for v in rng.integers(0, 2**32, 2)]
        for rounds in (7, 10):
            det.det_philox_raw_n(*c, *k, rounds, out)
            assert list(out) == philox_py(c, k, rounds)


def test_streams_are_uncorrelated_across_index_step_and_stream(det):
    """Seven rounds leave no safety margin by the authors' own account, and the engine's counters are highly structured
    (particle index, step, stream in fixed words): a statistical smoke test over exactly those axes -- neighbouring particle
    indices, consecutive steps, the motion and resample streams of one (index, step), two seeds one bit apart -- on moments,
    lag correlations and a 2-D equidistribution chi-square of the uniforms."""
    n = 1 << 18
    u = {}
    for name, (seed, stream, step) in {"base": (42, 3, 7), "next_step": (42, 3, 8), "other_stream": (42, 4, 7), "seed_bit": (43, 3, 7)}.items():
        a, b = np.empty(n), np.empty(n)
        det.det_uniform2_v(seed, stream, step, 0, n, dp(a), dp(b))
        u[name] = (a, b)
    a, b = u["base"]
    lim = 4.5 / math.sqrt(12 * n) * math.sqrt(12)  # |corr| of n iid pairs: sigma = 1/sqrt(n)
    for lag in (1, 2, 3, 64, 256, 4096):  # neighbouring / wave-strided / tile-strided particle indices
        assert abs(np.corrcoef(a[:-lag], a[lag:])[0, 1]) < lim, lag
        assert abs(np.corrcoef(a[:-lag], b[lag:])[0, 1]) < lim, lag
    assert abs(np.corrcoef(a, b)[0, 1]) < lim
    for other in ("next_step", "other_stream", "seed_bit"):
        for x in u[other]:
            assert abs(np.corrcoef(a, x)[0, 1]) < lim, other
            assert not np.array_equal(a, x)
    # 2-D equidistribution of (u_i, u_{i+1}) on a 32 x 32 grid: chi-square with 1023 degrees of freedom
    h, _, _ = np.histogram2d(a[:-1], a[1:], bins=32, range=[[0, 1], [0, 1]])
    chi2 = ((h - (n - 1) / 1024) ** 2 / ((n - 1) / 1024)).sum()
    assert 1023 - 5 * math.sqrt(2 * 1023) < chi2 < 1023 + 5 * math.sqrt(2 * 1023), chi2
    # bit balance of the 53-bit mantissa image
    bits = (a * 2**53).astype(np.uint64)
    for k in range(0, 53, 4):
        frac = np.mean((bits >> np.uint64(k)) & np.uint64(1))
        assert abs(frac - 0.5) < 4.5 * 0.5 / math.sqrt(n), k


def test_uniform_and_normal_streams(det):
    n = 200000
    u0 = np.empty(n)
    u1 = np.empty(n)
    det.det_uniform2_v(42, 4, 7, 0, n, dp(u0), dp(u1))
    assert u0.min() >= 0 and u0.max() < 1 and u1.min() >= 0 and u1.max() < 1
    assert abs(u0.mean() - 0.5) < 5e-3 and abs(u0.var() - 1 / 12) < 2e-3
    assert np.all(u0 * 2**53 == np.floor(u0 * 2**53))
    z0 = np.empty(n)
    z1 = np.empty(n)
    det.det_normal2_v(42, 3, 7, 0, n, dp(z0), dp(z1))
    assert np.all(np.isfinite(z0)) and np.all(np.isfinite(z1))
    for z in (z0, z1):
        assert abs(z.mean()) < 1e-2 and abs(z.var() - 1) < 1.5e-2
        assert abs(np.mean(z**4) - 3) < 0.1
    assert abs(np.mean(z0 * z1)) < 1e-2
    # counter-based: a sub-range reproduces the same values, another step does not
    w0 = np.empty(10)
    w1 = np.empty(10)
    det.det_normal2_v(42, 3, 7, 1000, 10, dp(w0), dp(w1))
    assert np.array_equal(w0, z0[1000:1010]) and np.array_equal(w1, z1[1000:1010])
    det.det_normal2_v(42, 3, 8, 1000, 10, dp(w0), dp(w1))
    assert not np.array_equal(w0, z0[1000:1010])


def test_normal_stream_ks_moments_tails_and_cross_step_independence(det):
    """the 7-round Philox -> Box-Muller NORMAL stream the filters consume (RR_STREAM_MOTION), beyond bit-identity
    with the host restatement: Kolmogorov-Smirnov against N(0, 1) for both outputs of a block, over the particle index (one step)
    AND over the step counter (one particle: the sequence a single hypothesis sees); the first six moments; tail mass beyond 3 / 4
    sigma; the pair (z0, z1) of one block and the pairs (z_t, z_t+1) of consecutive steps uncorrelated, also in their squares (the
    failure mode of a weak counter mix is dependence, not a wrong marginal); and the radius / angle split of the pair uniform."""
    from scipy import stats

    n = 1 << 19
    z0, z1 = np.empty(n), np.empty(n)
    det.det_normal2_v(1, 3, 11, 0, n, dp(z0), dp(z1))  # seed 1 (bench.py's), the motion stream, one step, 2^19 particle indices
    for z in (z0, z1):
        d, p = stats.kstest(z, "norm")
        assert p > 1e-4, (d, p)
        m = [float(np.mean(z**k)) for k in range(1, 7)]
        se = [math.sqrt(v / n) for v in (1, 2, 15, 96, 945, 10170)]  # sd of z^k's sample mean under N(0, 1): Var(z^k) = E z^2k - (E z^k)^2
        for got, want, s in zip(m, (0, 1, 0, 3, 0, 15), se):
            assert abs(got - want) < 5 * s, (m, want)
        for t, tail in ((3.0, 2 * stats.norm.sf(3.0)), (4.0, 2 * stats.norm.sf(4.0))):
            k = int(np.count_nonzero(np.abs(z) > t))
            assert abs(k - n * tail) < 5 * math.sqrt(n * tail) + 1, (t, k, n * tail)
    lim = 5 / math.sqrt(n)
    assert abs(np.corrcoef(z0, z1)[0, 1]) < lim and abs(np.corrcoef(z0**2, z1**2)[0, 1]) < lim
    # Box-Muller's own structure: radius^2 / 2 is Exp(1), the angle is uniform, and the two are independent
    r2, ang = 0.5 * (z0**2 + z1**2), np.arctan2(z1, z0)
    assert stats.kstest(r2, "expon").pvalue > 1e-4 and stats.kstest((ang + math.pi) / (2 * math.pi), "uniform").pvalue > 1e-4
    assert abs(np.corrcoef(r2, ang)[0, 1]) < lim
    # one particle index over 2^16 consecutive steps (the noise sequence ONE hypothesis integrates), and its neighbour
    T = 1 << 16
    seq = np.empty((2, T))
    a, b = np.empty(1), np.empty(1)
    for k, index in enumerate((123_456, 123_457)):
        for t in range(T):
            det.det_normal2_v(1, 3, t, index, 1, dp(a), dp(b))
            seq[k, t] = a[0]
    for k in range(2):
        assert stats.kstest(seq[k], "norm").pvalue > 1e-4
        for lag in (1, 2, 7):
            assert abs(np.corrcoef(seq[k, :-lag], seq[k, lag:])[0, 1]) < 5 / math.sqrt(T), (k, lag)
            assert abs(np.corrcoef(seq[k, :-lag] ** 2, seq[k, lag:] ** 2)[0, 1]) < 5 / math.sqrt(T), (k, lag)
    assert abs(np.corrcoef(seq[0], seq[1])[0, 1]) < 5 / math.sqrt(T)
