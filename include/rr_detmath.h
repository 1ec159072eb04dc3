/* rr_detmath.h -- the deterministic arithmetic contract ("D-spec") of the engine.
 *
 * Every transcendental the hot path needs (exp, log, sin/cos, atan2) and the
 * counter-based noise source (Philox4x32, RR_PHILOX_ROUNDS = 7 rounds -> Box-Muller) are written here as a
 * fixed sequence of IEEE-754 binary64 operations: +, -, *, /, sqrt, fma,
 * round-to-nearest-even and integer bit manipulation.  Those primitives are
 * correctly rounded on both x86-64 and gfx950, so a translation unit compiled
 * with floating-point contraction OFF (-ffp-contract=off) produces bit-identical
 * results on the host CPU and on the MI355X.  The HIP kernels
 * (rust_robotics_amd/csrc) and the deterministic CPU restatement
 * (oracle/det_spec.c) both include this header; the literal restatement of the
 * reference (oracle/ref_literal.c) does NOT -- it uses libm exactly like the
 * reference uses Rust's std f64 methods, so a defect in this header shows up
 * as a tolerance failure between the two oracles (tests/test_detmath.py).
 *
 * Reference call sites these replace (all under /root/reference/crates):
 *   f64::exp   rust_robotics_localization/src/particle_filter.rs:476-479
 *              rust_robotics_slam/src/fastslam1.rs:180
 *   f64::sin/cos  particle_filter.rs:292-293, fastslam1.rs:73-74,146-147
 *   f64::atan2    fastslam1.rs:97
 *   rand_distr::Normal  particle_filter.rs:259-287, fastslam1.rs:124-131
 *     (the reference RNG is the unseedable thread-local ChaCha12; the engine
 *      owns its noise source)
 *
 * Accuracy (measured against mpmath in tests/test_detmath.py): <= 2.5 ulp for
 * exp/sin/cos/atan2/log on the ranges the path uses.
 */
#ifndef RR_DETMATH_H
#define RR_DETMATH_H

#include <stdint.h>
#include "rr_detmath_consts.h"

#if defined(__HIPCC__)
#define RR_HD __host__ __device__ static inline
#else
#define RR_HD static inline
#endif

RR_HD double rr_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
/* One Horner step p * z + C with C a compile-time coefficient: the same correctly rounded fma as rr_fma on both sides (same bits).
 * On the device it is spelled so that the COEFFICIENT never occupies a vector register.  Left to itself the gfx950 code generator
 * (VOP3 cannot hold a 64-bit literal) materialises every literal coefficient in a VECTOR register pair first -- v_mov_b32 x 2 +
 * v_fmac_f64: three vector instructions per step (round 6: 296 v_mov_b32 in k_step_lazy<EST>'s text, ~100 of its 454 per-particle
 * instructions).  Here the two halves go into VCC by scalar moves and ONE v_fma_f64 reads VCC as its addend, all inside one asm
 * statement: the scalar unit issues beside other waves' vector work, and the coefficient lives for exactly these three
 * instructions.  That last point is the whole difficulty -- every form that let the compiler SEE a coefficient in scalar registers
 * lost (read off the ISA / measured on the MI355X, round 6): a v_fma_f64 asm with an "s" operand, or s_mov_b32 asm feeding a
 * builtin fma, or a __constant__ table read by s_load_dwordx16 -- the scheduler hoists the scalar moves / loads to the top of the
 * kernel, ~100 SGPRs spill into VGPR lanes (v_writelane / v_readlane), the 64-VGPR builds spill to scratch, and
 * k_step_lazy<EST> went from 33.8 to 40.4 us; with a dependency token pinning the moves behind the polynomial's argument the
 * compiler copied the SGPRs back into VGPRs to use v_fmac. */
#if defined(__HIP_DEVICE_COMPILE__)
template <unsigned long long BITS>
__device__ static inline double rr_horner_vcc(double p, double z) {
  asm("s_mov_b32 vcc_lo, %2\n\ts_mov_b32 vcc_hi, %3\n\tv_fma_f64 %0, %0, %1, vcc"
      : "+v"(p)
      : "v"(z), "n"((unsigned int)(BITS & 0xffffffffull)), "n"((unsigned int)(BITS >> 32))
      : "vcc");
  return p;
}
#define RR_POLY_TOKEN(tok, z) ((void)0)
#define RR_HORNER_ASM(p, z, C) rr_horner_vcc<__builtin_bit_cast(unsigned long long, (double)(C))>((p), (z))
#else
#define RR_POLY_TOKEN(tok, z) ((void)0)
#define RR_HORNER_ASM(p, z, C) __builtin_fma((p), (z), (C))
#endif
#define RR_HORNER_PLAIN(p, z, C) __builtin_fma((p), (z), (C))
#ifndef RR_ASM_EXP
#define RR_ASM_EXP 1
#endif
#ifndef RR_ASM_LOG
#define RR_ASM_LOG 1
#endif
#ifndef RR_ASM_SIN
#define RR_ASM_SIN 1
#endif
#ifndef RR_ASM_COS
#define RR_ASM_COS 1
#endif
#ifndef RR_ASM_ATAN
#define RR_ASM_ATAN 1
#endif
#if RR_ASM_EXP
#define RR_HORNER_EXP RR_HORNER_ASM
#else
#define RR_HORNER_EXP RR_HORNER_PLAIN
#endif
#if RR_ASM_LOG
#define RR_HORNER_LOG RR_HORNER_ASM
#else
#define RR_HORNER_LOG RR_HORNER_PLAIN
#endif
#if RR_ASM_SIN
#define RR_HORNER_SIN RR_HORNER_ASM
#else
#define RR_HORNER_SIN RR_HORNER_PLAIN
#endif
#if RR_ASM_COS
#define RR_HORNER_COS RR_HORNER_ASM
#else
#define RR_HORNER_COS RR_HORNER_PLAIN
#endif
#if RR_ASM_ATAN
#define RR_HORNER_ATAN RR_HORNER_ASM
#else
#define RR_HORNER_ATAN RR_HORNER_PLAIN
#endif
RR_HD double rr_rint(double x) { return __builtin_rint(x); }
RR_HD double rr_sqrt(double x) { return __builtin_sqrt(x); }
#if defined(__HIP_DEVICE_COMPILE__)
#if !defined(__gfx950__)
#error "rr_sqrt_core's agreement bound rests on the accuracy of gfx950's v_rsq_f64 seed (measured, tools/ubench/rsq_accuracy.hip): build with --offload-arch=gfx950 only"
#endif
/* CONTRACT (read this before relying on "bit-exact"): every GPU evaluates this function identically, so 1 GPU, 8 GPUs and
 * any two runs agree bit for bit, always.  Against the CPU side of the D-spec (oracle/det_spec.c: a correctly rounded sqrt)
 * the agreement is PROBABILISTIC: identical except when sqrt(x) lies within 2^-42.6 ulp of a rounding boundary, i.e. with
 * probability < 2^-41 per evaluation (0 of 6.9e10 measured).  At 1e6 x 32 pairs per step that is about one last-bit
 * difference per 1e5 steps; a difference can move one quantised weight by one unit and, rarely, one resample index.  The
 * CPU <-> GPU parity tests are therefore exact on every case they run but not a proof for arbitrarily long runs; the second
 * residual correction that would make it one costs 2 of the pair loop's 14.4 VALU instructions (~3 us of a 50 us step).
 *
 * Square root for the device's hot loops: v_rsq_f64, one Goldschmidt refinement, ONE residual correction -- the core of
 * LLVM's correctly rounded f64 sqrt lowering for gfx950 without its second residual correction, without the 2^256
 * rescale it wraps around inputs below 2^-767 and without the 0/inf select (for x == 0, inf or NaN this returns NaN;
 * callers -- rr_pf_weight_fused -- redo such particles with rr_sqrt).
 * Agreement with the correctly rounded root (what the CPU side of the D-spec computes): the hardware seed is accurate
 * to e <= 2^-24.17 (measured over 2^36 arguments, tools/ubench/rsq_accuracy.hip); Goldschmidt leaves g and h with the
 * same relative error 1.5 e^2, and the correction g + (x - g^2) h then misses sqrt(x) by 2.25 e^4 <= 2^-95.6 relative =
 * 2^-42.6 ulp before the final rounding.  The result therefore differs from the correctly rounded one only if sqrt(x)
 * lies within 2^-42.6 ulp of a rounding boundary: probability < 2^-41 per evaluation for the worst seed, ~2^-47 for a
 * typical one; measured 0 of 6.9e10.  (Two corrections WITHOUT the Goldschmidt step: 1468 of 6.9e10; the second
 * correction after it: 2 more VALU instructions per (particle, landmark) pair for the last 2^-41.) */
__device__ static inline double rr_sqrt_core(double x) {
  double y = __builtin_amdgcn_rsq(x);
  double g = x * y;
  double h = 0.5 * y;
  double r = __builtin_fma(-h, g, 0.5);
  g = __builtin_fma(g, r, g);
  h = __builtin_fma(h, r, h);
  double d = __builtin_fma(-g, g, x);
  g = __builtin_fma(d, h, g);
  return g;
}
#endif
RR_HD double rr_fabs(double x) { return __builtin_fabs(x); }

RR_HD uint64_t rr_d2u(double x) {
  uint64_t u;
  __builtin_memcpy(&u, &x, 8);
  return u;
}
RR_HD double rr_u2d(uint64_t u) {
  double x;
  __builtin_memcpy(&x, &u, 8);
  return x;
}

/* 2^k for k in [-1022, 1023] */
RR_HD double rr_pow2i(int k) { return rr_u2d((uint64_t)(k + 1023) << 52); }

/* p * 2^k, k in [-2098, 2046]; two exact-power multiplies, the second one may
 * round (subnormal result) or overflow exactly as IEEE prescribes. */
RR_HD double rr_scale2(double p, int k) {
  int k1 = k / 2;
  int k2 = k - k1;
  return (p * rr_pow2i(k1)) * rr_pow2i(k2);
}

/* ------------------------------------------------------------------ exp */
RR_HD double rr_exp(double x) {
  if (x != x) return x;
  if (x > 709.782712893384) return rr_u2d(0x7ff0000000000000ull);
  if (x < -745.1332191019412) return 0.0;
  double k = rr_rint(x * RR_LOG2E);
  double r = rr_fma(-k, RR_LN2_HI, x);
  r = rr_fma(-k, RR_LN2_LO, r);
  RR_POLY_TOKEN(tok, r);
  double p = RR_EXP_C_12;
  p = RR_HORNER_EXP(p, r, RR_EXP_C_11);
  p = RR_HORNER_EXP(p, r, RR_EXP_C_10);
  p = RR_HORNER_EXP(p, r, RR_EXP_C_9);
  p = RR_HORNER_EXP(p, r, RR_EXP_C_8);
  p = RR_HORNER_EXP(p, r, RR_EXP_C_7);
  p = RR_HORNER_EXP(p, r, RR_EXP_C_6);
  p = RR_HORNER_EXP(p, r, RR_EXP_C_5);
  p = RR_HORNER_EXP(p, r, RR_EXP_C_4);
  p = RR_HORNER_EXP(p, r, RR_EXP_C_3);
  p = RR_HORNER_EXP(p, r, RR_EXP_C_2);
  p = RR_HORNER_EXP(p, r, RR_EXP_C_1);
  p = RR_HORNER_EXP(p, r, RR_EXP_C_0);
  return rr_scale2(p, (int)k);
}

/* ------------------------------------------------------------------ log (x > 0) */
/* log of a finite positive NORMAL double with bit pattern u, times 2^e_bias: the branch-free core */
RR_HD double rr_log_core(uint64_t u, int e) {
  e += (int)(u >> 52) - 1023;
  double m = rr_u2d((u & 0x000fffffffffffffull) | 0x3ff0000000000000ull); /* [1,2) */
  if (m > RR_SQRT2) {
    m = m * 0.5;
    e += 1;
  }
  double f = m - 1.0;
  double s = f / (2.0 + f);
  double z = s * s;
  RR_POLY_TOKEN(tok, z);
  double p = RR_LOG_C_9;
  p = RR_HORNER_LOG(p, z, RR_LOG_C_8);
  p = RR_HORNER_LOG(p, z, RR_LOG_C_7);
  p = RR_HORNER_LOG(p, z, RR_LOG_C_6);
  p = RR_HORNER_LOG(p, z, RR_LOG_C_5);
  p = RR_HORNER_LOG(p, z, RR_LOG_C_4);
  p = RR_HORNER_LOG(p, z, RR_LOG_C_3);
  p = RR_HORNER_LOG(p, z, RR_LOG_C_2);
  p = RR_HORNER_LOG(p, z, RR_LOG_C_1);
  p = RR_HORNER_LOG(p, z, RR_LOG_C_0);
  /* atanh(s) = s + s*z*p ; log(m) = 2 atanh(s) */
  double t = s * z;
  double lm = 2.0 * rr_fma(t, p, s);
  double de = (double)e;
  return rr_fma(de, RR_LN2_HI, rr_fma(de, RR_LN2_LO, lm));
}

RR_HD double rr_log(double x) {
  if (x != x || x < 0.0) return rr_u2d(0x7ff8000000000000ull);
  if (x == 0.0) return rr_u2d(0xfff0000000000000ull);
  uint64_t u = rr_d2u(x);
  if (u >= 0x7ff0000000000000ull) return x; /* +inf */
  int e = 0;
  if (u < 0x0010000000000000ull) { /* subnormal */
    x = x * 0x1p54;
    u = rr_d2u(x);
    e = -54;
  }
  return rr_log_core(u, e);
}

/* ------------------------------------------------------------------ sin/cos kernels, |r| <= pi/4 */
RR_HD double rr_sin_kernel(double r) {
  double s = r * r;
  RR_POLY_TOKEN(tok, s);
  double p = RR_SIN_C_7;
  p = RR_HORNER_SIN(p, s, RR_SIN_C_6);
  p = RR_HORNER_SIN(p, s, RR_SIN_C_5);
  p = RR_HORNER_SIN(p, s, RR_SIN_C_4);
  p = RR_HORNER_SIN(p, s, RR_SIN_C_3);
  p = RR_HORNER_SIN(p, s, RR_SIN_C_2);
  p = RR_HORNER_SIN(p, s, RR_SIN_C_1);
  p = RR_HORNER_SIN(p, s, RR_SIN_C_0);
  return rr_fma(r * s, p, r);
}
RR_HD double rr_cos_kernel(double r) {
  double s = r * r;
  RR_POLY_TOKEN(tok, s);
  double p = RR_COS_C_7;
  p = RR_HORNER_COS(p, s, RR_COS_C_6);
  p = RR_HORNER_COS(p, s, RR_COS_C_5);
  p = RR_HORNER_COS(p, s, RR_COS_C_4);
  p = RR_HORNER_COS(p, s, RR_COS_C_3);
  p = RR_HORNER_COS(p, s, RR_COS_C_2);
  p = RR_HORNER_COS(p, s, RR_COS_C_1);
  p = RR_HORNER_COS(p, s, RR_COS_C_0);
  return rr_fma(s * s, p, rr_fma(-0.5, s, 1.0));
}

/* (sin, cos) of r + q pi/2 from (sin r, cos r): q odd swaps the two, then sin changes sign for q = 2, 3 and cos for q = 1, 2.
 * Written without branches (two selects, two sign-bit flips -- a sign-bit flip IS negation, zeros and NaNs included): the four-way
 * switch it replaces compiled to nested exec-mask branches on the device. */
RR_HD void rr_quadrant(int q, double sr, double cr, double* s, double* c) {
  const int odd = q & 1;
  const double s0 = odd ? cr : sr;
  const double c0 = odd ? sr : cr;
  *s = rr_u2d(rr_d2u(s0) ^ ((uint64_t)((unsigned)q & 2u) << 62));
  *c = rr_u2d(rr_d2u(c0) ^ ((uint64_t)((unsigned)(q + 1) & 2u) << 62));
}

/* sin and cos of x; 3-term fma Cody-Waite reduction, intended for |x| < ~1e6
 * (yaw angles); beyond 2^30 the result is defined (deterministic) but loses
 * accuracy.  Non-finite x -> NaN for both. */
RR_HD void rr_sincos(double x, double* s, double* c) {
  if (!(rr_fabs(x) < 0x1p30)) {
    if (x != x || rr_fabs(x) == rr_u2d(0x7ff0000000000000ull)) {
      *s = *c = rr_u2d(0x7ff8000000000000ull);
      return;
    }
    /* deterministic fold of huge finite arguments (|x| >= 2^30 never occurs for
     * yaw angles; the result is defined, not accurate): x -= 2pi*rint(x/2pi),
     * each pass removes ~52 bits of magnitude */
    for (int it = 0; it < 40 && !(rr_fabs(x) < 0x1p30); ++it) {
      double n = rr_rint(x / RR_TWO_PI);
      x = rr_fma(-n, RR_TWO_PI, x);
    }
  }
  double k = rr_rint(x * RR_TWO_OVER_PI);
  double r = rr_fma(-k, RR_PIO2_1, x);
  r = rr_fma(-k, RR_PIO2_2, r);
  r = rr_fma(-k, RR_PIO2_3, r);
  int q = (int)((long long)k & 3);
  rr_quadrant(q, rr_sin_kernel(r), rr_cos_kernel(r), s, c);
}

/* sin(2 pi u), cos(2 pi u) for u in [0,1): exact quadrant split of u. */
RR_HD void rr_sincos2pi(double u, double* s, double* c) {
  double q = rr_rint(4.0 * u);
  double f = rr_fma(-0.25, q, u); /* exact, |f| <= 1/8 */
  double r = f * RR_TWO_PI;
  rr_quadrant((int)q, rr_sin_kernel(r), rr_cos_kernel(r), s, c);
}

/* ------------------------------------------------------------------ atan2 */
/* atan of t in [0, +inf) (t == +inf allowed) */
RR_HD double rr_atan_pos(double y, double x) {
  /* returns atan(y/x) for y >= 0, x >= 0, not both zero/inf handled by caller */
  int swap = y > x;
  double num = swap ? x : y;
  double den = swap ? y : x;
  /* num/den in [0,1]; above tan(pi/8): atan(t) = pi/4 + atan((t - 1)/(t + 1)) with the quotient formed
   * directly as (num - den)/(num + den) -- one division on either branch */
  int big = num > RR_TAN_PIO8 * den;
  double t = (big ? num - den : num) / (big ? num + den : den);
  double base_hi = big ? RR_PIO4_HI : 0.0, base_lo = big ? RR_PIO4_LO : 0.0;
  double z = t * t;
  RR_POLY_TOKEN(tok, z);
  double p = RR_ATAN_C_13;
  p = RR_HORNER_ATAN(p, z, RR_ATAN_C_12);
  p = RR_HORNER_ATAN(p, z, RR_ATAN_C_11);
  p = RR_HORNER_ATAN(p, z, RR_ATAN_C_10);
  p = RR_HORNER_ATAN(p, z, RR_ATAN_C_9);
  p = RR_HORNER_ATAN(p, z, RR_ATAN_C_8);
  p = RR_HORNER_ATAN(p, z, RR_ATAN_C_7);
  p = RR_HORNER_ATAN(p, z, RR_ATAN_C_6);
  p = RR_HORNER_ATAN(p, z, RR_ATAN_C_5);
  p = RR_HORNER_ATAN(p, z, RR_ATAN_C_4);
  p = RR_HORNER_ATAN(p, z, RR_ATAN_C_3);
  p = RR_HORNER_ATAN(p, z, RR_ATAN_C_2);
  p = RR_HORNER_ATAN(p, z, RR_ATAN_C_1);
  p = RR_HORNER_ATAN(p, z, RR_ATAN_C_0);
  double a = base_hi + (rr_fma(t * z, p, t) + base_lo);
  if (swap) a = RR_PIO2_1 - (a - RR_PIO2_2);
  return a;
}

RR_HD double rr_atan2(double y, double x) {
  if (x != x || y != y) return rr_u2d(0x7ff8000000000000ull);
  const double inf = rr_u2d(0x7ff0000000000000ull);
  uint64_t sy = rr_d2u(y) & 0x8000000000000000ull;
  int xneg = (rr_d2u(x) >> 63) != 0;
  double ay = rr_fabs(y), ax = rr_fabs(x);
  double a;
  if (ay == 0.0) {
    a = xneg ? RR_PI_HI : 0.0;
  } else if (ax == 0.0) {
    a = RR_PIO2_1;
  } else if (ax == inf || ay == inf) {
    if (ax == inf && ay == inf) a = xneg ? 3.0 * RR_PIO4_HI : RR_PIO4_HI;
    else if (ax == inf) a = xneg ? RR_PI_HI : 0.0;
    else a = RR_PIO2_1;
  } else {
    a = rr_atan_pos(ay, ax);
    if (xneg) a = RR_PI_HI - (a - RR_PI_LO);
  }
  return rr_u2d(rr_d2u(a) | sy);
}

/* ------------------------------------------------------------------ Philox4x32 (7 rounds in the engine's streams, see below) */
/* Salmon et al., "Parallel random numbers: as easy as 1, 2, 3" (SC'11);
 * constants are the published ones. */
#define RR_PHILOX_M0 0xD2511F53u
#define RR_PHILOX_M1 0xCD9E8D57u
#define RR_PHILOX_W0 0x9E3779B9u
#define RR_PHILOX_W1 0xBB67AE85u

typedef struct rr_philox4 {
  uint32_t v[4];
} rr_philox4;

/* The engine's streams run RR_PHILOX_ROUNDS = 7 rounds: the smallest round count Salmon et al. report as
 * Crush-resistant for Philox4x32 (their Table 2; 10 is the conservative default) -- no safety margin by the authors'
 * own account, taken because the generator is ~150 of k_step_lazy's ~870 instructions per particle.  Both forms are pinned by
 * the published Random123 known answers (7 and 10 rounds) and by an independent integer evaluation of the round function;
 * the streams' moments, lag / step / stream / seed correlations and 2-D equidistribution are smoke-tested over the engine's
 * own (structured) counter layout (tests/test_detmath.py). */
#define RR_PHILOX_ROUNDS 7

RR_HD rr_philox4 rr_philox4x32_n(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, int rounds) {
  for (int i = 0; i < rounds; ++i) {
    uint64_t p0 = (uint64_t)RR_PHILOX_M0 * c0;
    uint64_t p1 = (uint64_t)RR_PHILOX_M1 * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += RR_PHILOX_W0;
    k1 += RR_PHILOX_W1;
  }
  rr_philox4 r;
  r.v[0] = c0; r.v[1] = c1; r.v[2] = c2; r.v[3] = c3;
  return r;
}
RR_HD rr_philox4 rr_philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
  return rr_philox4x32_n(c0, c1, c2, c3, k0, k1, 10);
}
RR_HD rr_philox4 rr_philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
  return rr_philox4x32_n(c0, c1, c2, c3, k0, k1, RR_PHILOX_ROUNDS);
}

/* Streams of the engine: what a (seed, stream, step, index) counter is used for. */
enum {
  RR_STREAM_INIT_XY = 1,   /* initial cloud jitter x,y      (particle_filter.rs:182-183) */
  RR_STREAM_INIT_YV = 2,   /* initial cloud jitter yaw,v    (particle_filter.rs:184-185) */
  RR_STREAM_MOTION = 3,    /* (n_v, n_w) per particle/step  (particle_filter.rs:280-287; fastslam1.rs:129-130) */
  RR_STREAM_RESAMPLE = 4,  /* per-draw uniform / r0         (particle_filter.rs:456; fastslam1.rs:219-220) */
  RR_STREAM_SIM = 5,       /* observation simulator noise   (fastslam1.rs:291-292) */
  RR_STREAM_PROPOSAL = 6   /* third proposal normal per particle/step (fastslam2.rs:248; the first two come from MOTION) */
};

/* two uniforms in [0,1) with 53 random bits each */
RR_HD void rr_uniform2(uint64_t seed, uint32_t stream, uint32_t step, uint64_t index, double* u0, double* u1) {
  rr_philox4 r = rr_philox4x32((uint32_t)index, (uint32_t)(index >> 32), step, stream,
                               (uint32_t)seed, (uint32_t)(seed >> 32));
  uint64_t a = ((uint64_t)r.v[1] << 32) | r.v[0];
  uint64_t b = ((uint64_t)r.v[3] << 32) | r.v[2];
  *u0 = (double)(a >> 11) * 0x1p-53;
  *u1 = (double)(b >> 11) * 0x1p-53;
}

/* Two independent N(0,1) draws: Box-Muller on one Philox block.  The radius uniform v is an odd multiple
 * of 2^-53 in (0,1) -- 52 random bits placed in the mantissa of a double in [1,2), minus (1 - 2^-53),
 * exact -- so log(v) < 0 strictly, the argument of the square root lies in [2^-52, 73.5) and neither the
 * log nor the sqrt needs a special case; the angle uniform is a multiple of 2^-52 in [0,1). */
RR_HD void rr_normal2(uint64_t seed, uint32_t stream, uint32_t step, uint64_t index, double* z0, double* z1) {
  rr_philox4 r = rr_philox4x32((uint32_t)index, (uint32_t)(index >> 32), step, stream,
                               (uint32_t)seed, (uint32_t)(seed >> 32));
  uint64_t a = ((uint64_t)r.v[1] << 32) | r.v[0];
  uint64_t b = ((uint64_t)r.v[3] << 32) | r.v[2];
  double v = rr_u2d(0x3ff0000000000000ull | (a >> 12)) - 0x1.fffffffffffffp-1;
  double u1 = rr_u2d(0x3ff0000000000000ull | (b >> 12)) - 1.0;
  double t = -2.0 * rr_log_core(rr_d2u(v), 0);
#if defined(__HIP_DEVICE_COMPILE__)
  double rad = rr_sqrt_core(t); /* == rr_sqrt(t) on [2^-767, inf) except with probability < 2^-41 (rr_sqrt_core's contract) */
#else
  double rad = rr_sqrt(t);
#endif
  double s, c;
  rr_sincos2pi(u1, &s, &c);
  *z0 = rad * c;
  *z1 = rad * s;
}

#endif /* RR_DETMATH_H */
