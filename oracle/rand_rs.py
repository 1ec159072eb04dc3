"""TEST INFRASTRUCTURE (oracle/): a CPU restatement of the random-number machinery the reference's seeded tests use, so that
`StdRng::seed_from_u64(7 | 17)` (rust_robotics_slam/src/fastslam2.rs:449, :496) can be replayed without a Rust toolchain.

The reference pins (Cargo.toml:37-38, Cargo.lock) rand 0.9.x, rand_chacha 0.9.0, rand_core 0.9.x, rand_distr 0.5.1.  None of those
crates is under /root/reference (no vendor directory, no registry, no network), so everything here is restated from their PUBLISHED
algorithms, each function naming what it restates:

  ChaCha            D. J. Bernstein's ChaCha with 12 rounds, 64-bit block counter (words 12-13), 64-bit stream id (words 14-15) = 0:
                    rand_chacha::ChaCha12Rng = rand 0.9's StdRng.  Pinned below by RFC 7539 sec. 2.3.2's block (20 rounds, the same
                    quarter round and layout) and by draft-strombergson-chacha-test-vectors TC1 (12 rounds, 256-bit zero key).
  seed_from_u64     rand_core::SeedableRng::seed_from_u64: a PCG32 (multiplier 6364136223846793005, increment 11634580027462260723,
                    XSH-RR output) fills the 32-byte seed four bytes at a time.
  next_u64 / f64    BlockRng: two consecutive 32-bit words, low word first; StandardUniform f64 = (next_u64 >> 11) * 2^-53;
                    Open01 f64 = float_from(1.0's exponent | next_u64 >> 12) - (1 - 2^-53).
  Uniform<f64>      rand 0.9 UniformFloat: value1_2 = float_from(1.0's exponent | next_u64 >> 12); (value1_2 - 1.0) * scale + low.
  StandardNormal    rand_distr 0.5.1: the 256-layer ziggurat of Marsaglia & Tsang as rand_distr::utils::ziggurat writes it (layer =
                    low 8 bits, u in [-1, 1) from the high 52 bits, tail by Marsaglia's exponential rejection).  The two 257-entry
                    tables are REGENERATED here by the recurrence of the crate's own generator script (R = 3.6541528853610088,
                    V = 4.92867323399e-3, x[0] = V / f(R), x[1] = R, x[i] = f^-1(V / x[i-1] + f(x[i-1])), x[256] = 0): the crate ships
                    them as 18-decimal literals of exactly these values; an entry may differ from the shipped literal in its last
                    bit where the generating machine's libm differs from this one's (a 1e-16 relative change of a sample, a flip of
                    an accept / reject decision with probability ~1e-16 per draw).
  Normal            rand_distr::Normal::sample = mean + std_dev * StandardNormal.

PARITY STAYS UNPINNED: no number computed by the reference's own binary is compared anywhere -- there is none
to compare with.  What this file buys: the reference's two seeded tests run against the literal restatement and against the GPU with
the reference's own seeds, trajectory and thresholds (tests/test_fs2_oracles.py, tests/test_gpu_fs2_parity.py), and the day a
built reference exists its seeded outputs can be laid beside these streams word for word.

Only tests/ may import this module (tests/test_abi_surface.py::test_product_never_imports_the_oracle)."""
import math
import struct

M32 = 0xFFFFFFFF
M64 = 0xFFFFFFFFFFFFFFFF


def _rotl(v, c):
    return ((v << c) & M32) | (v >> (32 - c))


def chacha_block(key_words, counter, stream, rounds):
    """One 64-byte block as 16 little-endian words: constants | key | 64-bit counter | 64-bit stream id."""
    s = [0x61707865, 0x3320646E, 0x79622D32, 0x6B206574] + list(key_words) + [counter & M32, (counter >> 32) & M32, stream & M32, (stream >> 32) & M32]
    x = list(s)

    def qr(a, b, c, d):
        x[a] = (x[a] + x[b]) & M32
        x[d] = _rotl(x[d] ^ x[a], 16)
        x[c] = (x[c] + x[d]) & M32
        x[b] = _rotl(x[b] ^ x[c], 12)
        x[a] = (x[a] + x[b]) & M32
        x[d] = _rotl(x[d] ^ x[a], 8)
        x[c] = (x[c] + x[d]) & M32
        x[b] = _rotl(x[b] ^ x[c], 7)

    for _ in range(rounds // 2):
        qr(0, 4, 8, 12)
        qr(1, 5, 9, 13)
        qr(2, 6, 10, 14)
        qr(3, 7, 11, 15)
        qr(0, 5, 10, 15)
        qr(1, 6, 11, 12)
        qr(2, 7, 8, 13)
        qr(3, 4, 9, 14)
    return [(a + b) & M32 for a, b in zip(x, s)]


class ChaChaRng:
    """rand_chacha::ChaCha{8,12,20}Rng: the key stream as a sequence of 32-bit words (blocks in counter order)."""

    def __init__(self, seed: bytes, rounds: int = 12):
        assert len(seed) == 32
        self.key = struct.unpack("<8I", seed)
        self.rounds, self.counter, self.buf, self.pos = rounds, 0, [], 0

    def next_u32(self) -> int:
        if self.pos == len(self.buf):
            self.buf, self.pos = chacha_block(self.key, self.counter, 0, self.rounds), 0
            self.counter += 1
        v = self.buf[self.pos]
        self.pos += 1
        return v

    def next_u64(self) -> int:
        lo = self.next_u32()
        return lo | (self.next_u32() << 32)

    def fill_bytes(self, n: int) -> bytes:
        """fill_via_u32_chunks: whole words are consumed"""
        words = [self.next_u32() for _ in range((n + 3) // 4)]
        return struct.pack(f"<{len(words)}I", *words)[:n]


def seed_from_u64(state: int) -> bytes:
    """rand_core::SeedableRng::seed_from_u64 for a 32-byte seed"""
    out = b""
    for _ in range(8):
        state = (state * 6364136223846793005 + 11634580027462260723) & M64
        xorshifted = (((state >> 18) ^ state) >> 27) & M32
        rot = state >> 59
        out += struct.pack("<I", ((xorshifted >> rot) | (xorshifted << ((32 - rot) & 31))) & M32)
    return out


class StdRng(ChaChaRng):
    """rand 0.9 `rngs::StdRng` (ChaCha12)"""

    def __init__(self, seed: bytes):
        super().__init__(seed, 12)

    @classmethod
    def seed_from_u64(cls, state: int) -> "StdRng":
        return cls(seed_from_u64(state))

    def random_f64(self) -> float:
        """StandardUniform: 53 random bits in [0, 1)"""
        return (self.next_u64() >> 11) * (1.0 / (1 << 53))

    def open01_f64(self) -> float:
        """Open01: (0, 1)"""
        return _float_with_exponent(self.next_u64() >> 12, 0) - (1.0 - 2.0 ** -53)


def _float_with_exponent(fraction52: int, exponent: int) -> float:
    """IntoFloat::into_float_with_exponent for f64"""
    return struct.unpack("<d", struct.pack("<Q", ((1023 + exponent) << 52) | fraction52))[0]


class Uniform:
    """rand 0.9 `Uniform::<f64>::new(low, high)` (UniformFloat)"""

    def __init__(self, low: float, high: float):
        assert low < high and math.isfinite(low) and math.isfinite(high)
        self.low, self.scale = low, high - low
        max_rand = 1.0 - 2.0 ** -52
        while self.scale * max_rand + low >= high:  # (never taken for the ranges the reference uses)
            self.scale = math.nextafter(self.scale, 0.0)

    def sample(self, rng: StdRng) -> float:
        value1_2 = _float_with_exponent(rng.next_u64() >> 12, 0)
        return (value1_2 - 1.0) * self.scale + self.low


ZIG_NORM_R = 3.654152885361008796


def _zig_tables():
    v = 4.92867323399e-3
    f = lambda x: math.exp(-x * x / 2.0)  # noqa: E731
    x = [v / f(ZIG_NORM_R), ZIG_NORM_R]
    for _ in range(2, 256):
        x.append(math.sqrt(-2.0 * math.log(v / x[-1] + f(x[-1]))))
    x.append(0.0)
    return x, [f(t) for t in x]


ZIG_NORM_X, ZIG_NORM_F = _zig_tables()


def standard_normal(rng: StdRng) -> float:
    """rand_distr 0.5.1 `StandardNormal` for f64 (utils::ziggurat, symmetric)"""
    while True:
        bits = rng.next_u64()
        i = bits & 0xFF
        u = _float_with_exponent(bits >> 12, 1) - 3.0  # [-1, 1)
        x = u * ZIG_NORM_X[i]
        if abs(x) < ZIG_NORM_X[i + 1]:
            return x
        if i == 0:  # the tail
            xx, yy = 1.0, 0.0
            while -2.0 * yy < xx * xx:
                x_ = rng.open01_f64()
                y_ = rng.open01_f64()
                xx = math.log(x_) / ZIG_NORM_R
                yy = math.log(y_)
            return xx - ZIG_NORM_R if u < 0.0 else ZIG_NORM_R - xx
        if ZIG_NORM_F[i + 1] + (ZIG_NORM_F[i] - ZIG_NORM_F[i + 1]) * rng.random_f64() < math.exp(-x * x / 2.0):
            return x


def normal(rng: StdRng, mean: float = 0.0, std_dev: float = 1.0) -> float:
    """rand_distr::Normal::sample"""
    return mean + std_dev * standard_normal(rng)
