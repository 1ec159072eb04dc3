"""oracle/rand_rs.py -- the restatement of rand 0.9's StdRng (ChaCha12), `seed_from_u64`, `Uniform<f64>` and rand_distr 0.5.1's
`StandardNormal` that replays the reference's seeded tests (fastslam2.rs:443-456, :491-545) -- against every published vector there
is for it: RFC 7539's block, the ChaCha test-vector draft's 8- and 12-round key streams, rand's own `test_stdrng_construction`, the
first entries of rand_distr's shipped ziggurat tables; plus the distributions' moments."""
import math
import struct

import numpy as np

from oracle import rand_rs as R


def hexblock(words):
    return struct.pack("<16I", *words).hex()


def test_chacha_block_function_rfc7539_and_strombergson_vectors():
    # RFC 7539 section 2.3.2: key 00..1f, counter 1, nonce 00:00:00:09 00:00:00:4a 00:00:00:00 (20 rounds; same quarter round and layout)
    key = struct.unpack("<8I", bytes(range(32)))
    assert hexblock(R.chacha_block(key, 1 | (0x09000000 << 32), 0x4A000000, 20)) == (
        "10f1e7e4d13b5915500fdd1fa32071c4c7d1f4c733c068030422aa9ac3d46c4ed2826446079faa0914c2d705d98b02a2b5129cd1de164eb9cbd083e8a2503c4e")
    # draft-strombergson-chacha-test-vectors-01, TC1 (all-zero 256-bit key and IV), first block of the key stream
    zero = (0,) * 8
    assert hexblock(R.chacha_block(zero, 0, 0, 20)).startswith("76b8e0ada0f13d90405d6ae55386bd28bdd219b8a08ded1aa836efcc8b770dc7")
    assert hexblock(R.chacha_block(zero, 0, 0, 12)) == (
        "9bf49a6a0755f953811fce125f2683d50429c3bb49e074147e0089a52eae155f0564f879d27ae3c02ce82834acfa8c793a629f2ca0de6919610be82f411326be")
    assert hexblock(R.chacha_block(zero, 0, 0, 8)).startswith("3e00ef2f895f40d67f5bb8e81f09a5a12c840ec3ce9a7f3b181be188ef711a1e")


def test_stdrng_is_chacha12_rands_own_construction_test():
    """rand 0.9 src/rngs/std.rs `test_stdrng_construction`: from_seed + next_u64, then from_rng (fill_bytes of the next 32 bytes)"""
    seed = bytes([1, 0, 0, 0, 23, 0, 0, 0, 200, 1, 0, 0, 210, 30, 0, 0] + [0] * 16)
    rng0 = R.StdRng(seed)
    x0 = rng0.next_u64()
    rng1 = R.StdRng(rng0.fill_bytes(32))
    assert [x0, rng1.next_u64()] == [10719222850664546238, 14064965282130556830]


def test_seed_from_u64_is_a_pcg32_expansion():
    s = R.seed_from_u64(0)
    # first PCG32 output from state 0: state' = INC; XSH-RR of it
    st = 11634580027462260723
    xs = (((st >> 18) ^ st) >> 27) & 0xFFFFFFFF
    rot = st >> 59
    assert struct.unpack("<I", s[:4])[0] == ((xs >> rot) | (xs << ((32 - rot) & 31))) & 0xFFFFFFFF
    assert len(s) == 32 and R.seed_from_u64(7) != R.seed_from_u64(17)
    a, b = R.StdRng.seed_from_u64(7), R.StdRng.seed_from_u64(7)
    assert [a.next_u64() for _ in range(70)] == [b.next_u64() for _ in range(70)]  # (70 words: across a block boundary)


def test_ziggurat_tables_match_the_shipped_literals():
    """rand_distr 0.5.1 src/ziggurat_tables.rs begins ZIG_NORM_X = [3.910757959537090045, 3.654152885361008796, 3.449278298560964462,
    3.320244733839166074, ...] and ZIG_NORM_F = [0.000477467764586655, 0.001260285930498598, ...] and ends with 0 and 1."""
    assert R.ZIG_NORM_X[:4] == [3.910757959537090045, 3.654152885361008796, 3.449278298560964462, 3.320244733839166074]
    assert abs(R.ZIG_NORM_F[0] - 0.000477467764586655) < 1e-18 and abs(R.ZIG_NORM_F[1] - 0.001260285930498598) < 1e-18
    assert len(R.ZIG_NORM_X) == len(R.ZIG_NORM_F) == 257 and R.ZIG_NORM_X[256] == 0.0 and R.ZIG_NORM_F[256] == 1.0
    assert all(a > b for a, b in zip(R.ZIG_NORM_X, R.ZIG_NORM_X[1:]))
    # every layer has the same area V (what makes it a ziggurat; the topmost one closes to the 12 digits V is given with)
    v = 4.92867323399e-3
    for i in range(1, 256):
        assert abs(R.ZIG_NORM_X[i] * (R.ZIG_NORM_F[i + 1] - R.ZIG_NORM_F[i]) - v) < (1e-13 if i < 255 else 1e-11)


def test_distributions_have_the_right_moments():
    rng = R.StdRng.seed_from_u64(123)
    z = np.array([R.standard_normal(rng) for _ in range(200_000)])
    assert abs(z.mean()) < 0.01 and abs(z.var() - 1.0) < 0.01 and abs((z ** 3).mean()) < 0.03 and abs((z ** 4).mean() - 3.0) < 0.06
    assert (np.abs(z) > R.ZIG_NORM_R).sum() > 20  # the tail branch ran (P = 2.6e-4)
    u = np.array([rng.random_f64() for _ in range(50_000)])
    assert 0.0 <= u.min() and u.max() < 1.0 and abs(u.mean() - 0.5) < 0.005
    uni = R.Uniform(0.0, 1.0 / 120)
    r = np.array([uni.sample(rng) for _ in range(50_000)])
    assert 0.0 <= r.min() and r.max() < 1.0 / 120 and abs(r.mean() * 120 - 0.5) < 0.005
    o = np.array([rng.open01_f64() for _ in range(10_000)])
    assert 0.0 < o.min() and o.max() < 1.0
    assert math.isclose(R.normal(R.StdRng.seed_from_u64(5), 2.0, 3.0), 2.0 + 3.0 * R.standard_normal(R.StdRng.seed_from_u64(5)))
