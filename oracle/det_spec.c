/* det_spec.c -- TEST INFRASTRUCTURE, not product code.
 *
 * Serial CPU evaluation of the engine's deterministic spec (D-spec): the
 * per-element functions of include/rr_pf_spec.h driven by plain loops, with
 * every cross-particle quantity that feeds back into the particle state built
 * from exact integer sums (include/rr_pf_spec.h "fixed-point CDF").  The HIP
 * kernels must agree with this file BIT FOR BIT (states, raw weights, integer
 * totals, resample indices); tests/test_parity_*.py assert exactly that.
 *
 * This file restates the same reference lines as oracle/ref_literal.c (see the
 * citations in include/rr_pf_spec.h); where the two differ (fused multiply-adds,
 * polynomial transcendentals, integer CDF instead of the serial float cumsum of
 * particle_filter.rs:448-453 / fastslam1.rs:213-216, closed-form systematic
 * positions instead of the serially accumulated r += 1/n of fastslam1.rs:230)
 * the difference is bounded and tested in tests/test_oracle_agreement.py.
 * PARITY PINNING: as ref_literal.c (RNG-level parity unpinned by construction).
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "rr_pf_spec.h"

/* ---- array shims over rr_detmath.h (tests/test_detmath.py) ---- */
void det_exp_v(size_t n, const double* x, double* o) { for (size_t i = 0; i < n; ++i) o[i] = rr_exp(x[i]); }
void det_log_v(size_t n, const double* x, double* o) { for (size_t i = 0; i < n; ++i) o[i] = rr_log(x[i]); }
void det_sincos_v(size_t n, const double* x, double* s, double* c) { for (size_t i = 0; i < n; ++i) rr_sincos(x[i], &s[i], &c[i]); }
void det_sincos2pi_v(size_t n, const double* x, double* s, double* c) { for (size_t i = 0; i < n; ++i) rr_sincos2pi(x[i], &s[i], &c[i]); }
void det_atan2_v(size_t n, const double* y, const double* x, double* o) { for (size_t i = 0; i < n; ++i) o[i] = rr_atan2(y[i], x[i]); }
void det_sqrt_v(size_t n, const double* x, double* o) { for (size_t i = 0; i < n; ++i) o[i] = rr_sqrt(x[i]); }
void det_div_v(size_t n, const double* a, const double* b, double* o) { for (size_t i = 0; i < n; ++i) o[i] = a[i] / b[i]; }
/* a * b + a in one rounding: the explicit-FMA contract of the spec headers (compiled to vfmadd here, v_fma_f64 on gfx950) */
void det_fma_v(size_t n, const double* a, const double* b, double* o) { for (size_t i = 0; i < n; ++i) o[i] = rr_fma(a[i], b[i], a[i]); }
void det_uniform2_v(uint64_t seed, uint32_t stream, uint32_t step, uint64_t first, size_t n, double* u0, double* u1) {
  for (size_t i = 0; i < n; ++i) rr_uniform2(seed, stream, step, first + i, &u0[i], &u1[i]);
}
void det_normal2_v(uint64_t seed, uint32_t stream, uint32_t step, uint64_t first, size_t n, double* z0, double* z1) {
  for (size_t i = 0; i < n; ++i) rr_normal2(seed, stream, step, first + i, &z0[i], &z1[i]);
}
void det_philox_raw(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t out[4]) {
  rr_philox4 r = rr_philox4x32_10(c0, c1, c2, c3, k0, k1);
  memcpy(out, r.v, 16);
}
/* the same with a given number of rounds; rounds == 0: the engine's own form (RR_PHILOX_ROUNDS) */
void det_philox_raw_n(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, int rounds, uint32_t out[4]) {
  rr_philox4 r = rounds ? rr_philox4x32_n(c0, c1, c2, c3, k0, k1, rounds) : rr_philox4x32(c0, c1, c2, c3, k0, k1);
  memcpy(out, r.v, 16);
}
int det_philox_rounds(void) { return RR_PHILOX_ROUNDS; }

/* ------------------------------------------------------------------ PF / MCL */

void det_pf_init(size_t n, uint64_t seed, uint64_t first_gid, const double st[4],
                 double* x, double* y, double* yaw, double* v) {
  for (size_t i = 0; i < n; ++i) rr_pf_init_one(seed, first_gid + i, st, &x[i], &y[i], &yaw[i], &v[i]);
}

/* nv/nw explicit (NULL => Philox noise keyed by (seed, step, first_gid + i)) */
void det_pf_predict(size_t n, double* x, double* y, double* yaw, double* v, double u0, double u1,
                    double dt, const double* nv, const double* nw, uint64_t seed, uint32_t step,
                    uint64_t first_gid, double sigma_v, double sigma_w) {
  for (size_t i = 0; i < n; ++i) {
    double a, b;
    if (nv && nw) {
      a = nv[i];
      b = nw[i];
    } else {
      rr_pf_motion_noise(seed, step, first_gid + i, sigma_v, sigma_w, &a, &b);
    }
    rr_pf_propagate_one(&x[i], &y[i], &yaw[i], &v[i], u0, u1, dt, a, b);
  }
}

/* mode 0 = RR_LIK_FUSED, 1 = RR_LIK_PRODUCT */
void det_pf_weights(size_t n, const double* x, const double* y, double* w, const double* obs,
                    size_t n_obs, double sigma, int mode) {
  rr_pf_lik k = rr_pf_lik_make(sigma);
  for (size_t i = 0; i < n; ++i)
    w[i] = mode ? rr_pf_weight_product(x[i], y[i], obs, (int)n_obs, k)
                : rr_pf_weight_fused(x[i], y[i], obs, (int)n_obs, k);
}

/* max over the non-NaN weights (order independent) */
double det_wmax(size_t n, const double* w) {
  double m = 0.0;
  for (size_t i = 0; i < n; ++i)
    if (w[i] > m) m = w[i];
  return m;
}

/* Integer image of the weights.  Returns 1 if usable (0 < w_max < inf), else 0
 * ("degenerate": PF falls back to uniform, particle_filter.rs:433-438).
 * out: shift, total T, q2 = sum q^2 (hi, lo).  n_global sizes the headroom. */
int det_fix_reduce(size_t n, const double* w, double w_max, uint64_t n_global, int* shift,
                   uint64_t* total, uint64_t* q2_hi, uint64_t* q2_lo) {
  *shift = 0; *total = 0; *q2_hi = 0; *q2_lo = 0;
  if (!(w_max > 0.0) || !(w_max < INFINITY)) return 0;
  int sh = rr_fix_shift(w_max, n_global);
  unsigned __int128 q2 = 0;
  uint64_t t = 0;
  for (size_t i = 0; i < n; ++i) {
    uint64_t q = rr_fix_quantize(w[i], sh);
    t += q;
    q2 += (unsigned __int128)q * q;
  }
  *shift = sh; *total = t; *q2_hi = (uint64_t)(q2 >> 64); *q2_lo = (uint64_t)q2;
  return 1;
}

/* inclusive integer CDF; degenerate (usable == 0) => q_i = 1 */
void det_fix_cdf(size_t n, const double* w, int usable, int shift, uint64_t base, uint64_t* cdf) {
  uint64_t c = base;
  for (size_t i = 0; i < n; ++i) {
    c += usable ? rr_fix_quantize(w[i], shift) : 1;
    cdf[i] = c;
  }
}

double det_fix_total_to_double(uint64_t total, int shift) { return rr_fix_total_to_double(total, shift); }
double det_fix_neff(uint64_t total, uint64_t q2_hi, uint64_t q2_lo) { return rr_fix_neff(total, q2_hi, q2_lo); }

/* multinomial indices for output slots [first_slot, first_slot + n_out) over a CDF of n
 * entries with grand total `total`; r explicit or Philox (seed, RESAMPLE stream, rstep, slot). */
void det_indices_multinomial(size_t n, const uint64_t* cdf, uint64_t total, size_t first_slot,
                             size_t n_out, const double* r, uint64_t seed, uint32_t rstep, uint32_t* idx) {
  for (size_t k = 0; k < n_out; ++k) {
    double rk, dummy;
    if (r) rk = r[k]; else rr_uniform2(seed, RR_STREAM_RESAMPLE, rstep, first_slot + k, &rk, &dummy);
    idx[k] = (uint32_t)rr_lower_bound_u64(cdf, n, rr_fix_target_multinomial(rk, total));
  }
}

/* CDF targets of the multinomial draws of output slots [first_slot, first_slot + n_out) (Philox RESAMPLE stream):
 * what a shard compares with its CDF interval (base, base + T_local] to find the slots it serves */
void det_targets_multinomial(uint64_t total, size_t first_slot, size_t n_out, uint64_t seed, uint32_t rstep, uint64_t* out) {
  for (size_t k = 0; k < n_out; ++k) {
    double rk, dummy;
    rr_uniform2(seed, RR_STREAM_RESAMPLE, rstep, first_slot + k, &rk, &dummy);
    out[k] = rr_fix_target_multinomial(rk, total);
  }
}

/* KLD-adaptive resample of the D-spec: multinomial draws over the integer CDF (n entries, grand
 * total `total`), r explicit or Philox (seed, RESAMPLE stream, rstep, draw index); sequential
 * evaluation of the stop rule.  Returns the new particle count, idx[0..count) = sources. */
uint64_t det_kld_required(uint64_t k, uint64_t min_p, uint64_t max_p, double eps, double z) {
  return rr_kld_required(k, min_p, max_p, eps, z);
}

size_t det_mcl_resample_adaptive(size_t n, const double* x, const double* y, const double* yaw, const uint64_t* cdf,
                                 uint64_t total, const double* r, uint64_t seed, uint32_t rstep, uint64_t min_p,
                                 uint64_t max_p, double eps, double z, uint32_t* idx) {
  int32_t* bins = (int32_t*)malloc(3 * max_p * sizeof(int32_t));
  size_t k = 0, count = 0;
  uint64_t required = min_p;
  while (count < max_p) {
    double rk, dummy;
    if (r) rk = r[count]; else rr_uniform2(seed, RR_STREAM_RESAMPLE, rstep, count, &rk, &dummy);
    size_t j = rr_lower_bound_u64(cdf, n, rr_fix_target_multinomial(rk, total));
    if (j >= n) j = n - 1;
    int32_t b0, b1, b2;
    rr_kld_bin(x[j], y[j], yaw[j], &b0, &b1, &b2);
    int seen = 0;
    for (size_t q = 0; q < k && !seen; ++q) seen = bins[3 * q] == b0 && bins[3 * q + 1] == b1 && bins[3 * q + 2] == b2;
    if (!seen) {
      bins[3 * k] = b0; bins[3 * k + 1] = b1; bins[3 * k + 2] = b2;
      ++k;
    }
    uint64_t need = rr_kld_required(k, min_p, max_p, eps, z);
    if (need > required) required = need;
    idx[count] = (uint32_t)j;
    const int stop = rr_kld_stop(count, required, min_p);
    ++count;
    if (stop) break;
  }
  free(bins);
  return count;
}

double det_resample_rho(uint64_t seed, uint32_t rstep) {
  double a, b;
  rr_uniform2(seed, RR_STREAM_RESAMPLE, rstep, 0, &a, &b);
  return a;
}

void det_indices_systematic(size_t n, const uint64_t* cdf, uint64_t total, uint64_t n_global,
                            size_t first_slot, size_t n_out, double rho, uint32_t* idx) {
  rr_sys_plan p = rr_sys_plan_make(rho, total, n_global);
  for (size_t k = 0; k < n_out; ++k)
    idx[k] = (uint32_t)rr_lower_bound_u64(cdf, n, rr_sys_target(p, first_slot + k));
}

static void gather4(size_t n, double* x, double* y, double* yaw, double* v, const uint32_t* idx) {
  double* t = (double*)malloc(4 * n * sizeof(double));
  for (size_t k = 0; k < n; ++k) {
    t[k] = x[idx[k]]; t[n + k] = y[idx[k]]; t[2 * n + k] = yaw[idx[k]]; t[3 * n + k] = v[idx[k]];
  }
  memcpy(x, t, n * sizeof(double));
  memcpy(y, t + n, n * sizeof(double));
  memcpy(yaw, t + 2 * n, n * sizeof(double));
  memcpy(v, t + 3 * n, n * sizeof(double));
  free(t);
}

/* weighted mean and covariance, particle_filter.rs:382-413, on (w_i / s) */
void det_pf_moments(size_t n, const double* x, const double* y, const double* yaw, const double* v,
                    const double* w, double s, double est[4], double cov[16]) {
  double e[4] = {0, 0, 0, 0};
  for (size_t i = 0; i < n; ++i) {
    double wi = w[i] / s;
    e[0] += wi * x[i]; e[1] += wi * y[i]; e[2] += wi * yaw[i]; e[3] += wi * v[i];
  }
  memcpy(est, e, sizeof e);
  if (!cov) return;
  for (int k = 0; k < 16; ++k) cov[k] = 0.0;
  for (size_t i = 0; i < n; ++i) {
    double wi = w[i] / s;
    double d[4] = {x[i] - e[0], y[i] - e[1], yaw[i] - e[2], v[i] - e[3]};
    for (int r = 0; r < 4; ++r)
      for (int c = 0; c < 4; ++c) cov[4 * r + c] += wi * d[r] * d[c];
  }
}

/* One D-spec PF/MCL step on a single shard holding all n particles.
 *   gate: 0 = N_eff < n*threshold (particle_filter.rs:337-345), 1 = every step
 *         (monte_carlo_localization.rs:298)
 *   scheme: 0 = multinomial (particle_filter.rs:441-473), 1 = systematic (fastslam1.rs:205-234)
 * w holds RAW weights on return when no resample fired (normalise with *s_out),
 * and exactly 1/n after a resample (then *s_out = 1).  Returns 1 if resampled. */
int det_pf_step(size_t n, double* x, double* y, double* yaw, double* v, double* w,
                double u0, double u1, double dt, double sigma_v, double sigma_w,
                const double* obs, size_t n_obs, double sigma, int lik_mode,
                double threshold, int gate, int scheme,
                uint64_t seed, uint32_t step, uint32_t rstep,
                uint32_t* idx_out, double* s_out) {
  det_pf_predict(n, x, y, yaw, v, u0, u1, dt, NULL, NULL, seed, step, 0, sigma_v, sigma_w);
  det_pf_weights(n, x, y, w, obs, n_obs, sigma, lik_mode);
  double wmax = det_wmax(n, w);
  int shift;
  uint64_t total, q2h, q2l;
  int usable = det_fix_reduce(n, w, wmax, n, &shift, &total, &q2h, &q2l);
  double s;
  double neff;
  if (usable && total > 0) {
    s = rr_fix_total_to_double(total, shift);
    neff = rr_fix_neff(total, q2h, q2l);
  } else { /* uniform fallback, particle_filter.rs:433-438 */
    usable = 0;
    for (size_t i = 0; i < n; ++i) w[i] = 1.0 / (double)n;
    s = 1.0;
    total = n;
    neff = (double)n;
  }
  *s_out = s;
  int fire = gate ? 1 : (neff < (double)n * threshold);
  if (!fire) return 0;
  uint64_t* cdf = (uint64_t*)malloc(n * sizeof(uint64_t));
  det_fix_cdf(n, w, usable, shift, 0, cdf);
  if (scheme == 0) det_indices_multinomial(n, cdf, total, 0, n, NULL, seed, rstep, idx_out);
  else det_indices_systematic(n, cdf, total, n, 0, n, det_resample_rho(seed, rstep), idx_out);
  free(cdf);
  gather4(n, x, y, yaw, v, idx_out);
  for (size_t i = 0; i < n; ++i) w[i] = 1.0 / (double)n;
  *s_out = 1.0;
  return 1;
}

/* ------------------------------------------------------------------ FastSLAM 1.0 */
/* D-spec storage is landmark-major planes, as on the device:
 *   maps[(l*6 + f) * n + p],  f in {x, y, c00, c10, c01, c11}. */

void det_fs1_model_default(rr_fs1_model* m) {
  m->dt = 0.1;
  m->q_sqrt0 = rr_sqrt(0.3);
  m->q_sqrt1 = rr_sqrt(0.0305);
  m->r00 = 0.5;
  m->r11 = 0.0305;
  m->init_threshold = 100.0;
  m->init_cov = NAN;
  m->init_test_lt = 0.0;
  m->nonpos_det_w = 1.0;
}

/* fastslam2.rs:18-30 */
void det_fs2_model_default(rr_fs2_model* m) {
  det_fs1_model_default(&m->base);
  m->base.init_cov = 10.0;      /* :255 */
  m->base.init_test_lt = 1.0;   /* :49-51 */
  m->base.nonpos_det_w = 1e-10; /* :289 */
  m->m0 = 0.1;
  m->m1 = 0.1;
  m->m2 = 0.01;
}

void det_fs2_proposal(const double pose[3], double u0, double u1, double zd, double za, const double* lm,
                      const rr_fs2_model* m, double mean[3], double cov[9]) {
  rr_fs2_proposal(pose, u0, u1, zd, za, lm, *m, mean, cov);
}

void det_fs2_sample(const double mean[3], const double cov[9], const double z[3], double pose[3]) {
  rr_fs2_sample(mean, cov, z, pose);
}

/* the sampling step for n particles; maps in plane layout [l][6][n]; noise: explicit 3 normals
 * per particle or NULL for the Philox streams */
void det_fs2_predict(size_t n, double* px, double* py, double* pyaw, const double* maps, double u0, double u1,
                     const double* z, size_t n_z, const double* noise, uint64_t seed, uint32_t step, uint64_t first_gid,
                     const rr_fs2_model* m) {
  for (size_t p = 0; p < n; ++p) {
    double zn[3];
    if (noise) { zn[0] = noise[3 * p]; zn[1] = noise[3 * p + 1]; zn[2] = noise[3 * p + 2]; }
    else rr_fs2_noise(seed, step, first_gid + p, zn);
    double pose[3] = {px[p], py[p], pyaw[p]};
    double lm[6] = {0, 0, 0, 0, 0, 0};
    if (n_z) {
      size_t id = (size_t)z[2];
      for (int f = 0; f < 6; ++f) lm[f] = maps[(id * 6 + (size_t)f) * n + p];
    }
    rr_fs2_predict_one(pose, u0, u1, n_z > 0, n_z ? z[0] : 0.0, n_z ? z[1] : 0.0, lm, zn, *m);
    px[p] = pose[0]; py[p] = pose[1]; pyaw[p] = pose[2];
  }
}

void det_fs1_predict(size_t n, double* px, double* py, double* pyaw, double u0, double u1,
                     const double* z0, const double* z1, uint64_t seed, uint32_t step,
                     uint64_t first_gid, const rr_fs1_model* m) {
  for (size_t p = 0; p < n; ++p) {
    double a, b;
    if (z0 && z1) { a = z0[p]; b = z1[p]; } else rr_fs1_motion_noise(seed, step, first_gid + p, &a, &b);
    rr_fs1_predict_one(&px[p], &py[p], &pyaw[p], u0, u1, a, b, *m);
  }
}

/* per particle, observations in order (the per-particle order of fastslam1.rs:250-256).
 * n_chunks > 1 mirrors the engine's observation chunking: chunk c covers observations
 * [c*len, (c+1)*len), len = ceil(n_z / n_chunks); its partial product starts from the old weight
 * (c == 0) or from 1.0, and the partials are multiplied together in chunk order.  With one chunk
 * this is exactly the reference's left-to-right weight *= likelihood. */
void det_fs1_observe(size_t n, const double* px, const double* py, const double* pyaw, double* pw,
                     double* maps, const double* z, size_t n_z, const rr_fs1_model* m, int n_chunks) {
  if (n_z == 0) return;
  if (n_chunks < 1) n_chunks = 1;
  size_t len = (n_z + (size_t)n_chunks - 1) / (size_t)n_chunks;
  for (size_t p = 0; p < n; ++p) {
    double w = 0.0;
    for (size_t c = 0; c * len < n_z; ++c) {
      double acc = c == 0 ? pw[p] : 1.0;
      size_t k1 = (c + 1) * len < n_z ? (c + 1) * len : n_z;
      for (size_t k = c * len; k < k1; ++k) {
        size_t id = (size_t)z[3 * k + 2];
        double e[6];
        for (int f = 0; f < 6; ++f) e[f] = maps[(id * 6 + f) * n + p];
        acc *= rr_fs1_update_one(px[p], py[p], pyaw[p], z[3 * k], z[3 * k + 1], e, *m);
        for (int f = 0; f < 6; ++f) maps[(id * 6 + f) * n + p] = e[f];
      }
      w = c == 0 ? acc : w * acc;
    }
    pw[p] = w;
  }
}

/* out-of-place gather of poses and landmark-major maps */
static void fs1_gather(size_t n, size_t L, double* px, double* py, double* pyaw, double* maps,
                       const uint32_t* idx) {
  double* t = (double*)malloc(n * sizeof(double));
  double* arrs[3] = {px, py, pyaw};
  for (int a = 0; a < 3; ++a) {
    for (size_t k = 0; k < n; ++k) t[k] = arrs[a][idx[k]];
    memcpy(arrs[a], t, n * sizeof(double));
  }
  for (size_t pl = 0; pl < L * 6; ++pl) {
    double* plane = maps + pl * n;
    for (size_t k = 0; k < n; ++k) t[k] = plane[idx[k]];
    memcpy(plane, t, n * sizeof(double));
  }
  free(t);
}

/* fastslam1.rs:237-266 in the D-spec.  Returns 1 if resampled.
 * The integer image (shift, T, sum q^2, CDF) is taken from the accumulated RAW
 * weights; the normalised weight is w / (T * 2^-shift). */
static int det_fs_observe_resample(size_t n, size_t L, double* px, double* py, double* pyaw, double* pw, double* maps,
                                   const double* z, size_t n_z, const rr_fs1_model* m, double nth, uint64_t seed,
                                   uint32_t rstep, int n_chunks, uint32_t* idx_out);

int det_fs1_update(size_t n, size_t L, double* px, double* py, double* pyaw, double* pw, double* maps,
                   double u0, double u1, const double* z, size_t n_z, const rr_fs1_model* m,
                   double nth, uint64_t seed, uint32_t step, uint32_t rstep, int n_chunks, uint32_t* idx_out) {
  det_fs1_predict(n, px, py, pyaw, u0, u1, NULL, NULL, seed, step, 0, m);
  return det_fs_observe_resample(n, L, px, py, pyaw, pw, maps, z, n_z, m, nth, seed, rstep, n_chunks, idx_out);
}

/* fastslam2.rs:331-374 in the D-spec: proposal sampling, then the FastSLAM 1.0 observation loop
 * with the FastSLAM 2.0 model switches, normalise / N_eff gate / systematic resample */
int det_fs2_update(size_t n, size_t L, double* px, double* py, double* pyaw, double* pw, double* maps,
                   double u0, double u1, const double* z, size_t n_z, const rr_fs2_model* m, const double* noise,
                   double nth, uint64_t seed, uint32_t step, uint32_t rstep, int n_chunks, uint32_t* idx_out) {
  det_fs2_predict(n, px, py, pyaw, maps, u0, u1, z, n_z, noise, seed, step, 0, m);
  return det_fs_observe_resample(n, L, px, py, pyaw, pw, maps, z, n_z, &m->base, nth, seed, rstep, n_chunks, idx_out);
}

static int det_fs_observe_resample(size_t n, size_t L, double* px, double* py, double* pyaw, double* pw, double* maps,
                                   const double* z, size_t n_z, const rr_fs1_model* m, double nth, uint64_t seed,
                                   uint32_t rstep, int n_chunks, uint32_t* idx_out) {
  det_fs1_observe(n, px, py, pyaw, pw, maps, z, n_z, m, n_chunks);
  double wmax = det_wmax(n, pw);
  int shift;
  uint64_t total, q2h, q2l;
  int usable = det_fix_reduce(n, pw, wmax, n, &shift, &total, &q2h, &q2l);
  usable = usable && total > 0;
  double neff = usable ? rr_fix_neff(total, q2h, q2l) : 0.0;
  int fire = neff < nth;
  if (fire) {
    if (usable) {
      uint64_t* cdf = (uint64_t*)malloc(n * sizeof(uint64_t));
      det_fix_cdf(n, pw, 1, shift, 0, cdf);
      det_indices_systematic(n, cdf, total, n, 0, n, det_resample_rho(seed, rstep), idx_out);
      free(cdf);
    } else { /* all-zero weights: the walk of fastslam1.rs:224-226 runs to the last particle */
      for (size_t k = 0; k < n; ++k) idx_out[k] = (uint32_t)(n - 1);
    }
    fs1_gather(n, L, px, py, pyaw, maps, idx_out);
    for (size_t p = 0; p < n; ++p) pw[p] = 1.0 / (double)n;
    return 1;
  }
  if (usable) { /* fastslam1.rs:196-203: normalise only if the sum is positive */
    double s = rr_fix_total_to_double(total, shift);
    for (size_t p = 0; p < n; ++p) pw[p] = pw[p] / s;
  }
  return 0;
}

/* observation simulator of the engine (rr_fs1_get_observations): fastslam1.rs:277-299 with the
 * Philox SIM stream; out rows (d, angle, id); returns the count */
size_t det_fs1_get_observations(const double xt[3], const double* lms, size_t L, double max_range, double r00,
                                double r11, uint64_t seed, uint32_t step, double* out) {
  size_t cnt = 0;
  double sr0 = rr_sqrt(r00), sr1 = rr_sqrt(r11);
  for (size_t l = 0; l < L; ++l) {
    double dx = lms[2 * l] - xt[0], dy = lms[2 * l + 1] - xt[1];
    double d = rr_sqrt(rr_fma(dy, dy, dx * dx));
    if (d <= max_range) {
      double angle = rr_normalize_angle(rr_atan2(dy, dx) - xt[2]);
      double z0, z1;
      rr_normal2(seed, RR_STREAM_SIM, step, l, &z0, &z1);
      out[3 * cnt] = rr_fma(z0, sr0, d);
      out[3 * cnt + 1] = rr_fma(z1, sr1, angle);
      out[3 * cnt + 2] = (double)l;
      ++cnt;
    }
  }
  return cnt;
}

/* fastslam1.rs:269-274: arg max of the weight, ties -> last */
size_t det_fs1_best_particle(size_t n, const double* pw) {
  size_t best = 0;
  for (size_t p = 1; p < n; ++p)
    if (!(pw[p] < pw[best])) best = p;
  return best;
}
