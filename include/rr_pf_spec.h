/* rr_pf_spec.h -- per-element arithmetic of the deterministic spec ("D-spec").
 *
 * One particle (or one particle x landmark pair) at a time, as `static inline`
 * functions that compile for the host (gcc/clang) and for gfx950 (hipcc).  The
 * HIP kernels call these from device code; oracle/det_spec.c calls the very same
 * functions from plain serial loops, so "GPU == D-spec" is a statement about the
 * kernels' plumbing (indexing, LDS staging, reductions, scans, gathers), and
 * "D-spec ~= reference" (tolerance 1e-6, the reference's own gate convention,
 * scripts/check_benchmark_gate.py:34-35) is checked against oracle/ref_literal.c,
 * which shares nothing with this file.
 *
 * All citations are /root/reference/crates/... file:line.
 */
#ifndef RR_PF_SPEC_H
#define RR_PF_SPEC_H

#include "rr_detmath.h"

/* ===================================================================== PF / MCL */

/* rust_robotics_localization/src/particle_filter.rs:279-296 (same arithmetic in
 * monte_carlo_localization.rs:236-253).  nv, nw are the already scaled noise
 * samples (sigma * z), exactly 0.0 when the sigma is 0 (Q5).  yaw is NOT wrapped
 * (Q3).  Operation order (v*cos)*dt as the reference (Q4); the final add is fused. */
RR_HD void rr_pf_propagate_one(double* x, double* y, double* yaw, double* v,
                               double u0, double u1, double dt, double nv, double nw) {
  double v_noisy = u0 + nv;
  double w_noisy = u1 + nw;
  double s, c;
  rr_sincos(*yaw, &s, &c);
  *x = rr_fma(v_noisy * c, dt, *x);
  *y = rr_fma(v_noisy * s, dt, *y);
  *yaw = rr_fma(w_noisy, dt, *yaw);
  *v = v_noisy;
}

/* Scaled motion noise of particle `gid` at step `step` (replaces the unseedable
 * rand::rng() draws of particle_filter.rs:280-287). */
RR_HD void rr_pf_motion_noise(uint64_t seed, uint32_t step, uint64_t gid,
                              double sigma_v, double sigma_w, double* nv, double* nw) {
  double z0, z1;
  rr_normal2(seed, RR_STREAM_MOTION, step, gid, &z0, &z1);
  *nv = sigma_v > 0.0 ? sigma_v * z0 : 0.0;
  *nw = sigma_w > 0.0 ? sigma_w * z1 : 0.0;
}

/* Initial cloud jitter, particle_filter.rs:181-185 (MCL: monte_carlo_localization.rs:190-194). */
RR_HD void rr_pf_init_one(uint64_t seed, uint64_t gid, const double st[4],
                          double* x, double* y, double* yaw, double* v) {
  double a, b, c, d;
  rr_uniform2(seed, RR_STREAM_INIT_XY, 0, gid, &a, &b);
  rr_uniform2(seed, RR_STREAM_INIT_YV, 0, gid, &c, &d);
  *x = st[0] + a * 2.0 - 1.0;
  *y = st[1] + b * 2.0 - 1.0;
  *yaw = st[2] + c * 0.5 - 0.25;
  *v = st[3] + d * 1.0 - 0.5;
}

/* Constants of the range likelihood, particle_filter.rs:476-479:
 *   coeff = 1/sqrt(2 pi sigma^2),  two_s2 = 2 sigma^2,  log_coeff = ln(coeff). */
typedef struct rr_pf_lik {
  double coeff;
  double two_s2;
  double inv_two_s2;
  double log_coeff;
} rr_pf_lik;

RR_HD rr_pf_lik rr_pf_lik_make(double sigma) {
  rr_pf_lik k;
  k.two_s2 = 2.0 * (sigma * sigma);
  k.coeff = 1.0 / rr_sqrt(RR_TWO_PI * (sigma * sigma));
  k.inv_two_s2 = 1.0 / k.two_s2;
  k.log_coeff = rr_log(k.coeff);
  return k;
}

/* squared range residual of one (particle, observation) pair,
 * particle_filter.rs:320-325: dx = particle - landmark (Q2). */
RR_HD double rr_pf_residual(double x, double y, double d_obs, double lx, double ly) {
  double dx = x - lx;
  double dy = y - ly;
  double d_pred = rr_sqrt(rr_fma(dy, dy, dx * dx));
  return d_obs - d_pred;
}

/* RR_LIK_PRODUCT: the reference's per-pair form w *= coeff*exp(-diff^2/(2 sigma^2))
 * (particle_filter.rs:317-328); obs = n_obs x (d, lx, ly). */
RR_HD double rr_pf_weight_product(double x, double y, const double* obs, int n_obs, rr_pf_lik k) {
  double w = 1.0;
  for (int l = 0; l < n_obs; ++l) {
    double diff = rr_pf_residual(x, y, obs[3 * l], obs[3 * l + 1], obs[3 * l + 2]);
    w *= k.coeff * rr_exp(-(diff * diff) / k.two_s2);
  }
  return w;
}

/* RR_LIK_FUSED: the same product with the exponentials merged,
 *   w = exp(L ln coeff - sum diff^2 / (2 sigma^2)),
 * one exp per particle instead of one per pair (SURVEY.md section 7 "hard parts").
 * Differs from the product form by ~1e-13 relative, and in the deep-underflow
 * regime (w < 1e-290) where the running product loses bits first. */
/* The squared range carries a floor of 2^-700 m^2 (q = dy^2 + (dx^2 + 2^-700), two fmas): it changes no
 * q above 2^-647 and moves the predicted range of a particle sitting on a landmark by < 2^-350 m, and it keeps
 * every argument of the device's bare square-root core inside its exact range without a per-pair check. */
#define RR_PF_Q_FLOOR 0x1p-700
RR_HD double rr_pf_residual_fused(double x, double y, double d_obs, double lx, double ly) {
  double dx = x - lx;
  double dy = y - ly;
  double q = rr_fma(dy, dy, rr_fma(dx, dx, RR_PF_Q_FLOOR));
  return d_obs - rr_sqrt(q);
}

RR_HD double rr_pf_weight_fused(double x, double y, const double* obs, int n_obs, rr_pf_lik k) {
  double ss = 0.0;
#if defined(__HIP_DEVICE_COMPILE__)
  /* branch-free inner loop on the sqrt core (== rr_sqrt for finite q >= 2^-767 except with probability < 2^-41 per
   * evaluation: rr_sqrt_core's contract in rr_detmath.h); an overflowing q (inf) turns
   * ss into NaN, which sends the particle through the exact form once */
  for (int l = 0; l < n_obs; ++l) {
    double dx = x - obs[3 * l + 1];
    double dy = y - obs[3 * l + 2];
    double q = rr_fma(dy, dy, rr_fma(dx, dx, RR_PF_Q_FLOOR));
    double diff = obs[3 * l] - rr_sqrt_core(q);
    ss = rr_fma(diff, diff, ss);
  }
  if (ss != ss) {
    ss = 0.0;
    for (int l = 0; l < n_obs; ++l) {
      double diff = rr_pf_residual_fused(x, y, obs[3 * l], obs[3 * l + 1], obs[3 * l + 2]);
      ss = rr_fma(diff, diff, ss);
    }
  }
#else
  for (int l = 0; l < n_obs; ++l) {
    double diff = rr_pf_residual_fused(x, y, obs[3 * l], obs[3 * l + 1], obs[3 * l + 2]);
    ss = rr_fma(diff, diff, ss);
  }
#endif
  return rr_exp(rr_fma(-ss, k.inv_two_s2, (double)n_obs * k.log_coeff));
}

#if defined(__cplusplus) && defined(__HIPCC__)
/* rr_pf_weight_fused for R particles of one thread at once -- per particle the very same operations in the same
 * order (so the same bits); what changes is the instruction stream: each observation is read once for all R
 * particles (four observations per LDS address computation), and the R dependency chains (~16 FP64 instructions per
 * pair, every one waiting for its predecessor) are independent of each other, so a wave always has an instruction
 * ready. */
template <int R>
__device__ inline void rr_pf_weight_fused_rows(const double (&x)[R], const double (&y)[R], const double* obs, int n_obs,
                                               rr_pf_lik k, double (&out)[R]) {
  double ss[R];
#pragma unroll
  for (int r = 0; r < R; ++r) ss[r] = 0.0;
  {
#pragma unroll 4
    for (int l = 0; l < n_obs; ++l) {
      const double d = obs[3 * l], lx = obs[3 * l + 1], ly = obs[3 * l + 2];
#pragma unroll
      for (int r = 0; r < R; ++r) {
        double dx = x[r] - lx;
        double dy = y[r] - ly;
        double q = rr_fma(dy, dy, rr_fma(dx, dx, RR_PF_Q_FLOOR));
#if defined(__HIP_DEVICE_COMPILE__)
        double diff = d - rr_sqrt_core(q);
#else
        double diff = d - rr_sqrt(q); /* (host pass of the compiler: never called) */
#endif
        ss[r] = rr_fma(diff, diff, ss[r]);
      }
    }
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    if (ss[r] != ss[r]) { /* an overflowing q: the exact form once (as rr_pf_weight_fused) */
      ss[r] = 0.0;
      for (int l = 0; l < n_obs; ++l) {
        double diff = rr_pf_residual_fused(x[r], y[r], obs[3 * l], obs[3 * l + 1], obs[3 * l + 2]);
        ss[r] = rr_fma(diff, diff, ss[r]);
      }
    }
    out[r] = rr_exp(rr_fma(-ss[r], k.inv_two_s2, (double)n_obs * k.log_coeff));
  }
}
#endif

/* ===================================================================== fixed-point CDF */
/* The resampling CDF is built from integer weights so that its value does not
 * depend on summation order (1 GPU, 8 GPUs and the CPU agree bit for bit):
 *   shift = (62 - ceil_log2(N)) - floor(log2(w_max)),  q_i = floor(w_i * 2^shift),
 * hence q_i < 2^(63-ceil_log2 N) and T = sum q_i < 2^63 exactly. */

RR_HD int rr_ceil_log2_u64(uint64_t n) {
  int b = 0;
  while (b < 63 && ((uint64_t)1 << b) < n) ++b;
  return b;
}

/* floor(log2(w)) for finite w > 0 (subnormals included) */
RR_HD int rr_ilogb_pos(double w) {
  uint64_t u = rr_d2u(w);
  int e = (int)(u >> 52);
  if (e == 0) { /* subnormal: w = m * 2^-1074 */
    uint64_t m = u & 0x000fffffffffffffull;
    int hb = 63 - __builtin_clzll(m);
    return hb - 1074;
  }
  return e - 1023;
}

RR_HD int rr_fix_shift(double w_max, uint64_t n_global) {
  return (62 - rr_ceil_log2_u64(n_global)) - rr_ilogb_pos(w_max);
}

/* floor(w * 2^shift) by mantissa shifting; w < 0, NaN -> 0; caller guarantees
 * w <= w_max so the result fits. */
RR_HD uint64_t rr_fix_quantize(double w, int shift) {
  uint64_t u = rr_d2u(w);
  if (u >= 0x7ff0000000000000ull) return 0; /* negative, inf or NaN */
  int e = (int)(u >> 52);
  uint64_t m = u & 0x000fffffffffffffull;
  int sh;
  if (e == 0) {
    sh = -1074 + shift;
  } else {
    m |= 0x0010000000000000ull;
    sh = e - 1075 + shift;
  }
  /* w <= w_max and shift = (62 - ceil_log2 N) - ilogb(w_max) give sh <= 10 - ceil_log2 N for a NORMAL
   * w_max.  For a subnormal w_max rr_fix_shift grows up to 1136 - ceil_log2 N while m has fewer than 52
   * significant bits; the clamp below then keeps only the top 10 extra bits on purpose: every weight
   * of such a set is itself subnormal (<= 52 - k significant bits), the image stays monotone and
   * < 2^63 / N, and CPU, 1 GPU and 8 GPUs still agree bit for bit.  Precision, not exactness, is what
   * is given up in that regime (weights below 2^-1022, where RR_LIK_PRODUCT's running product has
   * already lost its low bits as well). */
  if (sh >= 0) return m << (sh > 10 ? 10 : sh);
  if (sh <= -64) return 0;
  return m >> (-sh);
}

/* S = T * 2^-shift as a double: the order-independent weight sum used for
 * normalisation in the D-spec (replaces the serial float sum of
 * particle_filter.rs:427 / fastslam1.rs:197). */
RR_HD double rr_fix_total_to_double(uint64_t total, int shift) {
  return rr_scale2((double)total, -shift);
}

/* N_eff = (sum w)^2 / sum w^2 from the exact integer sums (particle_filter.rs:416-423,
 * fastslam1.rs:186-193); q2 = sum q_i^2 as a 128-bit integer (hi, lo). */
RR_HD double rr_fix_neff(uint64_t total, uint64_t q2_hi, uint64_t q2_lo) {
  if (q2_hi == 0 && q2_lo == 0) return 0.0;
  double t = (double)total;
  double q2 = rr_fma((double)q2_hi, 0x1p64, (double)q2_lo);
  return (t * t) / q2;
}

RR_HD void rr_mul64wide(uint64_t a, uint64_t b, uint64_t* hi, uint64_t* lo) {
  unsigned __int128 p = (unsigned __int128)a * b;
  *hi = (uint64_t)(p >> 64);
  *lo = (uint64_t)p;
}

/* Multinomial draw r in [0,1) -> CDF target: smallest P >= 1 with r*T <= P, i.e.
 * index = first i with C_i >= P  <=>  first i with r <= C_i / T
 * (particle_filter.rs:459-465 "r <= cum_w").  Targets are clamped to >= 1 so that every
 * target lies in exactly one shard's interval (base, base + T_local]; the only inputs this
 * changes are r == 0 exactly with a zero-weight first particle (probability 2^-53). */
RR_HD uint64_t rr_fix_target_multinomial(double r, uint64_t total) {
  uint64_t R = (uint64_t)(r * 0x1p53); /* exact: r is a multiple of 2^-53 */
  unsigned __int128 p = (unsigned __int128)R * total + (((unsigned __int128)1 << 53) - 1);
  uint64_t t = (uint64_t)(p >> 53);
  return t ? t : 1; /* r == 0 selects the first particle of non-zero weight */
}

/* ===================================================================== KLD-adaptive particle count */
/* monte_carlo_localization.rs:322-385.  The reference draws particles one at a time and stops at
 * the first count m with m >= min_particles and m >= required, required = max over the draws so
 * far of kld_required_particles(number of distinct occupied bins).  Everything below is a pure
 * function of the draw sequence, so the engine evaluates all max_particles candidate draws in
 * parallel and takes the first count that satisfies the stop rule. */

/* `x as i32` of Rust: saturating, NaN -> 0 */
RR_HD int32_t rr_sat_i32(double v) {
  if (v != v) return 0;
  if (v <= -2147483648.0) return (int32_t)(-2147483647 - 1);
  if (v >= 2147483647.0) return 2147483647;
  return (int32_t)v;
}

/* quantize_particle :380-385: bins of 0.5 m x 0.5 m x 15 degrees */
#define RR_KLD_XY_BIN 0.5
#define RR_KLD_YAW_BIN (15.0 * 3.14159265358979323846 / 180.0)
RR_HD void rr_kld_bin(double x, double y, double yaw, int32_t* xb, int32_t* yb, int32_t* ab) {
  *xb = rr_sat_i32(__builtin_floor(x / RR_KLD_XY_BIN));
  *yb = rr_sat_i32(__builtin_floor(y / RR_KLD_XY_BIN));
  *ab = rr_sat_i32(__builtin_floor(yaw / RR_KLD_YAW_BIN));
}

/* kld_required_particles :367-378 (Wilson-Hilferty bound), clamped to [min_p, max_p] */
RR_HD uint64_t rr_kld_required(uint64_t k_bins, uint64_t min_p, uint64_t max_p, double eps, double z) {
  if (k_bins <= 1) return min_p;
  double km1 = (double)(k_bins - 1);
  double a = 2.0 / (9.0 * km1);
  double term = (1.0 - a) + z * rr_sqrt(a);
  double n = (km1 / (2.0 * eps)) * ((term * term) * term); /* powi(3) */
  double c = __builtin_ceil(n);
  uint64_t r;
  if (!(c > 0.0)) r = 0; /* `as usize`: negative and NaN -> 0 */
  else if (c >= 18446744073709551615.0) r = ~(uint64_t)0;
  else r = (uint64_t)c;
  if (r < min_p) r = min_p;
  if (r > max_p) r = max_p;
  return r;
}

/* stop rule :354-357 for the draw with 0-based index m: count = m + 1 particles so far */
RR_HD int rr_kld_stop(uint64_t m, uint64_t required, uint64_t min_p) {
  return (m + 1 >= min_p) && (m + 1 >= required);
}

/* Systematic resampling (fastslam1.rs:219-231): output i sits at (i + rho)/n,
 * rho = u0 in [0,1); offs = floor(rho*T); target_i = ceil((i*T + offs)/n). */
typedef struct rr_sys_plan {
  uint64_t q;    /* T / n */
  uint64_t rem;  /* T % n */
  uint64_t offs; /* floor(rho * T) */
  uint64_t n;
} rr_sys_plan;

RR_HD rr_sys_plan rr_sys_plan_make(double rho, uint64_t total, uint64_t n) {
  rr_sys_plan p;
  uint64_t R = (uint64_t)(rho * 0x1p53);
  p.offs = (uint64_t)(((unsigned __int128)R * total) >> 53);
  p.q = total / n;
  p.rem = total % n;
  p.n = n;
  return p;
}

RR_HD uint64_t rr_sys_target(rr_sys_plan p, uint64_t i) {
  uint64_t t = i * p.rem + p.offs; /* < n^2 + 2^63 <= 2^64 for n < 2^31 */
  uint64_t r = i * p.q + (t + p.n - 1) / p.n;
  return r ? r : 1; /* position 0 selects the first particle of non-zero weight */
}

/* Number of output slots i in [0, n) whose target rr_sys_target(p, i) is <= c, for a CDF value
 * c <= T: the inverse of the target map, used to turn "which source does slot i copy" into
 * "which slots does source j feed": source j with inclusive CDF C_j and predecessor C_{j-1}
 * feeds exactly the slots [slots_upto(C_{j-1}), slots_upto(C_j)).
 *   target_i <= c  <=>  i*T + offs <= c*n  (and c >= 1 because targets are clamped to >= 1)
 *   count = floor((c*n - offs) / T) + 1, capped at n.
 * The 128-by-64 division is done as a double estimate corrected with exact 128-bit products, so
 * the same code runs on the host and on gfx950 (no __udivti3 on the device). */
#if defined(__HIPCC__)
#define RR_HD_NOINLINE __host__ __device__ __attribute__((noinline)) static
#else
#define RR_HD_NOINLINE static __attribute__((noinline, unused))
#endif
RR_HD_NOINLINE uint64_t rr_sys_slots_upto_exact(rr_sys_plan p, uint64_t total, uint64_t c) {
  if (c == 0 || total == 0) return 0;
  unsigned __int128 a = (unsigned __int128)c * p.n;
  if (a < p.offs) return 0;
  unsigned __int128 d = a - p.offs;
  double dd = (double)(uint64_t)(d >> 64) * 0x1p64 + (double)(uint64_t)d;
  double est = dd / (double)total;
  uint64_t q = est >= 0x1p63 ? ~(uint64_t)0 >> 1 : (uint64_t)est;
  if (q > p.n) q = p.n; /* the count is capped at n anyway; keeps q*total inside 128 bits */
  while ((unsigned __int128)q * total > d) --q;
  while (q < p.n && (unsigned __int128)(q + 1) * total <= d) ++q;
  uint64_t cnt = q + 1;
  return cnt < p.n ? cnt : p.n;
}

/* Fast path of the same integer function: x = (c*n - offs)/T evaluated in doubles is within
 * 2^-50 * n of the exact quotient, so floor(x) is the exact integer whenever x is at least
 * `guard` away from the neighbouring integers; only then is the cheap answer taken, otherwise
 * the exact 128-bit evaluation decides.  Identical results by construction. */
typedef struct rr_sys_inv {
  double n, offs, inv_total, guard;
} rr_sys_inv;

RR_HD rr_sys_inv rr_sys_inv_make(rr_sys_plan p, uint64_t total) {
  rr_sys_inv v;
  v.n = (double)p.n;
  v.offs = (double)p.offs;
  v.inv_total = total ? 1.0 / (double)total : 0.0;
  v.guard = 0x1p-48 * (double)p.n + 0x1p-30; /* >> accumulated rounding error of the 4 operations */
  return v;
}

RR_HD uint64_t rr_sys_slots_upto(rr_sys_plan p, rr_sys_inv v, uint64_t total, uint64_t c) {
  if (c == 0 || total == 0) return 0;
  double x = rr_fma((double)c, v.n, -v.offs) * v.inv_total; /* ~ (c*n - offs)/T */
  double f = __builtin_floor(x);
  double frac = x - f;
  if (x >= 0.0 && frac > v.guard && frac < 1.0 - v.guard && x < 0x1p52) {
    uint64_t cnt = (uint64_t)f + 1;
    return cnt < p.n ? cnt : p.n;
  }
  return rr_sys_slots_upto_exact(p, total, c);
}

/* first i in [0,n) with c[i] >= target (c inclusive, non-decreasing); n-1 if none */
RR_HD uint64_t rr_lower_bound_u64(const uint64_t* c, uint64_t n, uint64_t target) {
  uint64_t lo = 0, hi = n;
  while (lo < hi) {
    uint64_t mid = lo + ((hi - lo) >> 1);
    if (c[mid] >= target) hi = mid; else lo = mid + 1;
  }
  return lo < n ? lo : n - 1;
}

/* ===================================================================== FastSLAM 1.0 */

/* model constants of rust_robotics_slam/src/fastslam1.rs:13-23 as run-time fields */
typedef struct rr_fs1_model {
  double dt;            /* DT = 0.1 */
  double q_sqrt0;       /* sqrt(Q_SIM[0][0]) = sqrt(0.3) */
  double q_sqrt1;       /* sqrt(Q_SIM[1][1]) = sqrt(0.0305) */
  double r00, r11;      /* R_SIM diagonal 0.5, 0.0305 */
  double init_threshold;/* 100.0: cov[(0,0)] > threshold => first observation (fastslam1.rs:143) */
  double init_cov;      /* NaN => reference-faithful "leave cov untouched" (Q11); else cov := init_cov * I */
  /* the two places where fastslam2.rs:255-291 differs from fastslam1.rs:140-183 */
  double init_test_lt;  /* 0 => first observation iff cov00 > threshold (fastslam1.rs:143);
                           1 => iff !(cov00 < threshold) (fastslam2.rs:49-51,262) */
  double nonpos_det_w;  /* likelihood factor when det S <= 0: 1.0 (fastslam1.rs:182), 1e-10 (fastslam2.rs:289) */
} rr_fs1_model;

/* fastslam1.rs:80-89 */
RR_HD double rr_normalize_angle(double a) {
  while (a > RR_PI_HI) a -= 2.0 * RR_PI_HI;
  while (a < -RR_PI_HI) a += 2.0 * RR_PI_HI;
  return a;
}

/* fastslam1.rs:123-137 + 70-77; z0,z1 unit normals.  (u0*DT)*cos(yaw) order (Q4). */
RR_HD void rr_fs1_predict_one(double* x, double* y, double* yaw, double u0, double u1,
                              double z0, double z1, rr_fs1_model m) {
  double un0 = rr_fma(z0, m.q_sqrt0, u0);
  double un1 = rr_fma(z1, m.q_sqrt1, u1);
  double s, c;
  rr_sincos(*yaw, &s, &c);
  *x = rr_fma(un0 * m.dt, c, *x);
  *y = rr_fma(un0 * m.dt, s, *y);
  *yaw = rr_normalize_angle(rr_fma(un1, m.dt, *yaw));
}

/* One (particle, observation) EKF update, fastslam1.rs:140-183.
 * lm = {x, y, c00, c10, c01, c11} (nalgebra Matrix2 is column-major).
 * Returns the likelihood factor to multiply into the particle weight
 * (1.0 on the first-observation branch and when det S <= 0, Q13). */
RR_HD double rr_fs1_update_one(double px, double py, double pyaw, double zd, double za,
                               double* lm, rr_fs1_model m) {
  double lx = lm[0], ly = lm[1];
  double p00 = lm[2], p10 = lm[3], p01 = lm[4], p11 = lm[5];
  if (m.init_test_lt != 0.0 ? !(p00 < m.init_threshold) : (p00 > m.init_threshold)) { /* :143-149 */
    double s, c;
    rr_sincos(pyaw + za, &s, &c);
    lm[0] = rr_fma(zd, c, px);
    lm[1] = rr_fma(zd, s, py);
    if (m.init_cov == m.init_cov) {
      lm[2] = m.init_cov; lm[3] = 0.0; lm[4] = 0.0; lm[5] = m.init_cov;
    }
    return 1.0;
  }
  /* observation model :92-99 (landmark - particle, Q2) and Jacobian :102-110 */
  double dx = lx - px;
  double dy = ly - py;
  /* squared range with the 2^-700 m^2 floor of the MCL likelihood (RR_PF_Q_FLOOR): no division by zero when a
   * particle sits exactly on its landmark estimate (the reference divides by 0 there), and every argument of the
   * device's bare square-root core is inside its exact range */
  double d2 = rr_fma(dy, dy, rr_fma(dx, dx, RR_PF_Q_FLOOR));
#if defined(__HIP_DEVICE_COMPILE__)
  double d = rr_sqrt_core(d2); /* == rr_sqrt(d2) for finite d2 >= 2^-767, up to rr_sqrt_core's 2^-41 contract */
#else
  double d = rr_sqrt(d2);
#endif
  double zp_a = rr_normalize_angle(rr_atan2(dy, dx) - pyaw);
  double y0 = zd - d;
  double y1 = rr_normalize_angle(za - zp_a);
  /* D-spec: the Jacobian's four quotients (:105-108: dx/d, dy/d, -dy/d2, dx/d2) and the four of S^-1
   * below (:164) share two reciprocals -- 2 divisions instead of 8 (+1 instead of 2 in the likelihood);
   * each entry differs from the literal quotient by <= 1.5 ulp (tests/test_oracle_agreement.py, 1e-6). */
  double rd = 1.0 / d, rd2 = rd * rd;
  double h00 = dx * rd, h01 = dy * rd, h10 = -dy * rd2, h11 = dx * rd2;
  /* HP = H * P */
  double hp00 = rr_fma(h01, p10, h00 * p00);
  double hp01 = rr_fma(h01, p11, h00 * p01);
  double hp10 = rr_fma(h11, p10, h10 * p00);
  double hp11 = rr_fma(h11, p11, h10 * p01);
  /* S = HP * H^T + R :161 */
  double s00 = rr_fma(hp01, h01, hp00 * h00) + m.r00;
  double s01 = rr_fma(hp01, h11, hp00 * h10);
  double s10 = rr_fma(hp11, h01, hp10 * h00);
  double s11 = rr_fma(hp11, h11, hp10 * h10) + m.r11;
  /* S^-1 :164 -- nalgebra try_inverse for 2x2: det == 0 => None => identity */
  double det = rr_fma(s00, s11, -(s10 * s01));
  double i00, i01, i10, i11;
  double rdet = 1.0 / det; /* inf when det == 0: not used then */
  if (det == 0.0) {
    i00 = 1.0; i01 = 0.0; i10 = 0.0; i11 = 1.0;
  } else {
    i00 = s11 * rdet; i01 = -s01 * rdet; i10 = -s10 * rdet; i11 = s00 * rdet;
    if (rr_fabs(det) < 0x1p-1000) {
      /* the reciprocal of det is on its way to overflow (inf from 2^-1024 down) while the reference's four quotients are
       * still finite (a collapsed landmark covariance under a tiny R; never seen in a tracking filter): det and the entries
       * of S scaled by 2^512 first -- exact -- then entry * reciprocal as above */
      const double rs = 1.0 / (det * 0x1p512);
      i00 = (s11 * 0x1p512) * rs; i01 = (-s01 * 0x1p512) * rs; i10 = (-s10 * 0x1p512) * rs; i11 = (s00 * 0x1p512) * rs;
    }
  }
  /* K = P * H^T * S^-1 :165 */
  double pht00 = rr_fma(p01, h01, p00 * h00);
  double pht01 = rr_fma(p01, h11, p00 * h10);
  double pht10 = rr_fma(p11, h01, p10 * h00);
  double pht11 = rr_fma(p11, h11, p10 * h10);
  double k00 = rr_fma(pht01, i10, pht00 * i00);
  double k01 = rr_fma(pht01, i11, pht00 * i01);
  double k10 = rr_fma(pht11, i10, pht10 * i00);
  double k11 = rr_fma(pht11, i11, pht10 * i01);
  /* landmark += K y :168-170 */
  lm[0] = lx + rr_fma(k01, y1, k00 * y0);
  lm[1] = ly + rr_fma(k11, y1, k10 * y0);
  /* P = (I - K H) P :173-174, no symmetrisation (Q12) */
  double a00 = 1.0 - rr_fma(k01, h10, k00 * h00);
  double a01 = -rr_fma(k01, h11, k00 * h01);
  double a10 = -rr_fma(k11, h10, k10 * h00);
  double a11 = 1.0 - rr_fma(k11, h11, k10 * h01);
  lm[2] = rr_fma(a01, p10, a00 * p00);
  lm[3] = rr_fma(a11, p10, a10 * p00);
  lm[4] = rr_fma(a01, p11, a00 * p01);
  lm[5] = rr_fma(a11, p11, a10 * p01);
  /* likelihood :177-182, uses s.determinant() = s00*s11 - s10*s01 */
  if (det > 0.0) {
    /* (y^T S^-1) y, row vector first as nalgebra evaluates it */
    double t0 = rr_fma(y1, i10, y0 * i00);
    double t1 = rr_fma(y1, i11, y0 * i01);
    double mahal = rr_fma(t1, y1, t0 * y0);
    /* exp(-m/2) / (2 pi sqrt(det)) = exp(-m/2) * sqrt(det) * (1/det) * (1/(2 pi)); below 2^-1000 the reciprocal of det
     * is on its way to overflow (inf from 2^-1024 down) while the reference's quotient is still finite: its literal form there */
    if (det < 0x1p-1000) return rr_exp(-0.5 * mahal) / (RR_TWO_PI * rr_sqrt(det));
    return ((rr_exp(-0.5 * mahal) * rr_sqrt(det)) * rdet) * RR_INV_TWO_PI;
  }
  return m.nonpos_det_w;
}

/* unit motion noise for FastSLAM particle gid at step (fastslam1.rs:129-130) */
RR_HD void rr_fs1_motion_noise(uint64_t seed, uint32_t step, uint64_t gid, double* z0, double* z1) {
  rr_normal2(seed, RR_STREAM_MOTION, step, gid, z0, z1);
}

/* ===================================================================== FastSLAM 2.0 proposal */
/* rust_robotics_slam/src/fastslam2.rs.  Everything after the pose has been sampled is the
 * FastSLAM 1.0 observation loop with the two model switches above (init_test_lt = 1,
 * nonpos_det_w = 1e-10, init_cov = 10).  3x3 matrices are row-major arrays of 9. */
typedef struct rr_fs2_model {
  rr_fs1_model base;
  double m0, m1, m2; /* MOTION_COV diagonal 0.1, 0.1, 0.01 (fastslam2.rs:30) */
} rr_fs2_model;

/* nalgebra 0.33 Matrix3::try_inverse (cofactor formula, None iff det == 0); returns 0 for None */
RR_HD int rr_inv3(const double* a, double* o) {
  double m11 = a[0], m12 = a[1], m13 = a[2], m21 = a[3], m22 = a[4], m23 = a[5], m31 = a[6], m32 = a[7], m33 = a[8];
  double minor_m12_m23 = rr_fma(m22, m33, -(m32 * m23));
  double minor_m11_m23 = rr_fma(m21, m33, -(m31 * m23));
  double minor_m11_m22 = rr_fma(m21, m32, -(m31 * m22));
  double det = rr_fma(m13, minor_m11_m22, rr_fma(m11, minor_m12_m23, -(m12 * minor_m11_m23)));
  if (det == 0.0) return 0;
  o[0] = minor_m12_m23 / det;
  o[1] = rr_fma(m13, m32, -(m33 * m12)) / det;
  o[2] = rr_fma(m12, m23, -(m22 * m13)) / det;
  o[3] = -minor_m11_m23 / det;
  o[4] = rr_fma(m11, m33, -(m31 * m13)) / det;
  o[5] = rr_fma(m13, m21, -(m23 * m11)) / det;
  o[6] = minor_m11_m22 / det;
  o[7] = rr_fma(m12, m31, -(m32 * m11)) / det;
  o[8] = rr_fma(m11, m22, -(m21 * m12)) / det;
  return 1;
}

/* motion_model :92-99 */
RR_HD void rr_fs2_motion(const double x[3], double u0, double u1, double dt, double o[3]) {
  double s, c;
  rr_sincos(x[2], &s, &c);
  o[0] = rr_fma(u0 * dt, c, x[0]);
  o[1] = rr_fma(u0 * dt, s, x[1]);
  o[2] = rr_normalize_angle(rr_fma(u1, dt, x[2]));
}

/* compute_proposal :173-216: mean and covariance of the proposal for one particle given the
 * FIRST observation (zd, za) of landmark lm = {x, y, c00, c10, c01, c11}. */
RR_HD void rr_fs2_proposal(const double pose[3], double u0, double u1, double zd, double za, const double* lm,
                           rr_fs2_model m, double mean[3], double cov[9]) {
  double xp[3];
  rr_fs2_motion(pose, u0, u1, m.base.dt, xp);
  /* g * MOTION_COV * g^T, g = motion_jacobian :103-118 */
  double s, c;
  rr_sincos(pose[2], &s, &c);
  double a = -u0 * m.base.dt * s; /* g[0][2] */
  double b = u0 * m.base.dt * c;  /* g[1][2] */
  double P[9];
  P[0] = rr_fma(a * m.m2, a, m.m0); P[1] = (a * m.m2) * b;           P[2] = a * m.m2;
  P[3] = (b * m.m2) * a;           P[4] = rr_fma(b * m.m2, b, m.m1); P[5] = b * m.m2;
  P[6] = m.m2 * a;                 P[7] = m.m2 * b;                  P[8] = m.m2;
  double p00 = lm[2], p10 = lm[3], p01 = lm[4], p11 = lm[5];
  if (!(p00 < m.base.init_threshold)) { /* :186-189 landmark not initialised */
    for (int i = 0; i < 3; ++i) mean[i] = xp[i];
    for (int i = 0; i < 9; ++i) cov[i] = P[i];
    return;
  }
  double dx = lm[0] - xp[0], dy = lm[1] - xp[1];
  double d2 = rr_fma(dy, dy, dx * dx);
  double d = rr_sqrt(d2);
  /* h_pose :139-147 (2x3) and h_lm :131-137 (2x2) */
  double h[6] = {-dx / d, -dy / d, 0.0, dy / d2, -dx / d2, -1.0};
  double l00 = dx / d, l01 = dy / d, l10 = -dy / d2, l11 = dx / d2;
  /* q_obs = h_lm * C * h_lm^T + R :195 */
  double hc00 = rr_fma(l01, p10, l00 * p00), hc01 = rr_fma(l01, p11, l00 * p01);
  double hc10 = rr_fma(l11, p10, l10 * p00), hc11 = rr_fma(l11, p11, l10 * p01);
  double q00 = rr_fma(hc01, l01, hc00 * l00) + m.base.r00, q01 = rr_fma(hc01, l11, hc00 * l10);
  double q10 = rr_fma(hc11, l01, hc10 * l00), q11 = rr_fma(hc11, l11, hc10 * l10) + m.base.r11;
  double qdet = rr_fma(q00, q11, -(q10 * q01));
  double i00, i01, i10, i11; /* :200 try_inverse or identity */
  if (qdet == 0.0) { i00 = 1.0; i01 = 0.0; i10 = 0.0; i11 = 1.0; }
  else { i00 = q11 / qdet; i01 = -q01 / qdet; i10 = -q10 / qdet; i11 = q00 / qdet; }
  double Pinv[9]; /* :202 */
  if (!rr_inv3(P, Pinv)) {
    for (int i = 0; i < 9; ++i) Pinv[i] = 0.0;
    Pinv[0] = Pinv[4] = Pinv[8] = 1e-6;
  }
  /* (H^T * Qinv) 3x2, then * H :203 */
  double t[6];
  for (int r = 0; r < 3; ++r) {
    t[2 * r] = rr_fma(h[3 + r], i10, h[r] * i00);
    t[2 * r + 1] = rr_fma(h[3 + r], i11, h[r] * i01);
  }
  double Ppi[9];
  for (int r = 0; r < 3; ++r)
    for (int q = 0; q < 3; ++q) Ppi[3 * r + q] = Pinv[3 * r + q] + rr_fma(t[2 * r + 1], h[3 + q], t[2 * r] * h[q]);
  double Pp[9]; /* :204 */
  if (!rr_inv3(Ppi, Pp))
    for (int i = 0; i < 9; ++i) Pp[i] = P[i];
  /* innovation :207-208 */
  double zp_a = rr_normalize_angle(rr_atan2(dy, dx) - xp[2]);
  double y0 = zd - d, y1 = rr_normalize_angle(za - zp_a);
  /* x_post = x_pred + ((P_post * H^T) * Qinv) * innovation :210 */
  for (int r = 0; r < 3; ++r) {
    double ph0 = rr_fma(Pp[3 * r + 2], h[2], rr_fma(Pp[3 * r + 1], h[1], Pp[3 * r] * h[0]));
    double ph1 = rr_fma(Pp[3 * r + 2], h[5], rr_fma(Pp[3 * r + 1], h[4], Pp[3 * r] * h[3]));
    double k0 = rr_fma(ph1, i10, ph0 * i00), k1 = rr_fma(ph1, i11, ph0 * i01);
    mean[r] = xp[r] + rr_fma(k1, y1, k0 * y0);
  }
  for (int i = 0; i < 9; ++i) cov[i] = Pp[i];
}

/* sample_pose_with_rng :219-239 + set_pose :77-81: mean + L * noise with L the lower Cholesky
 * factor (nalgebra Cholesky::new, lower triangle of cov) or, when the matrix is not positive
 * definite, diag(sqrt(max(c_ii, 0))) */
RR_HD void rr_fs2_sample(const double mean[3], const double c[9], const double z[3], double pose[3]) {
  double L[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  int ok = 0;
  double d0 = c[0];
  if (d0 != 0.0 && d0 >= 0.0) {
    double l00 = rr_sqrt(d0), l10 = c[3] / l00, l20 = c[6] / l00;
    double d1 = rr_fma(-l10, l10, c[4]);
    double c21 = rr_fma(-l10, l20, c[7]);
    if (d1 != 0.0 && d1 >= 0.0) {
      double l11 = rr_sqrt(d1), l21 = c21 / l11;
      double d2 = rr_fma(-l21, l21, rr_fma(-l20, l20, c[8]));
      if (d2 != 0.0 && d2 >= 0.0) {
        L[0] = l00; L[3] = l10; L[4] = l11; L[6] = l20; L[7] = l21; L[8] = rr_sqrt(d2);
        ok = 1;
      }
    }
  }
  if (!ok) {
    L[0] = rr_sqrt(c[0] > 0.0 ? c[0] : 0.0);
    L[4] = rr_sqrt(c[4] > 0.0 ? c[4] : 0.0);
    L[8] = rr_sqrt(c[8] > 0.0 ? c[8] : 0.0);
  }
  pose[0] = mean[0] + L[0] * z[0];
  pose[1] = mean[1] + rr_fma(L[4], z[1], L[3] * z[0]);
  pose[2] = rr_normalize_angle(mean[2] + rr_fma(L[8], z[2], rr_fma(L[7], z[1], L[6] * z[0])));
}

/* the sampling step of fastslam2_update_with_rng :339-358 for one particle: proposal from the
 * first observation, or the noisy motion model when there is none */
RR_HD void rr_fs2_predict_one(double pose[3], double u0, double u1, int has_obs, double zd, double za,
                              const double* lm, const double z[3], rr_fs2_model m) {
  if (has_obs) {
    double mean[3], cov[9], np[3];
    rr_fs2_proposal(pose, u0, u1, zd, za, lm, m, mean, cov);
    rr_fs2_sample(mean, cov, z, np);
    pose[0] = np[0]; pose[1] = np[1]; pose[2] = np[2];
  } else { /* :349-357 */
    double un0 = rr_fma(z[0], m.base.q_sqrt0, u0), un1 = rr_fma(z[1], m.base.q_sqrt1, u1);
    double np[3];
    rr_fs2_motion(pose, un0, un1, m.base.dt, np);
    pose[0] = np[0]; pose[1] = np[1]; pose[2] = rr_normalize_angle(np[2]);
  }
}

/* three unit normals of particle gid at step (fastslam2.rs:248) */
RR_HD void rr_fs2_noise(uint64_t seed, uint32_t step, uint64_t gid, double z[3]) {
  double spare;
  rr_normal2(seed, RR_STREAM_MOTION, step, gid, &z[0], &z[1]);
  rr_normal2(seed, RR_STREAM_PROPOSAL, step, gid, &z[2], &spare);
}

#endif /* RR_PF_SPEC_H */
