/* ref_literal.c -- TEST INFRASTRUCTURE, not product code.
 *
 * A plain-C restatement of the reference's CPU algorithm for
 * the sampling-based localization hot path, following the reference line by
 * line: same loop order, same operation order, serial left-to-right sums,
 * libm transcendentals (Rust's f64::{sin,cos,exp,sqrt,atan2} are the platform
 * libm on Linux), no fused multiply-add (rustc never contracts).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 *
 * Reference files restated (all under /root/reference/crates):
 *   rust_robotics_localization/src/particle_filter.rs
 *   rust_robotics_localization/src/monte_carlo_localization.rs  (fixed-N and KLD-adaptive resample)
 *   rust_robotics_slam/src/fastslam1.rs
 *   rust_robotics_slam/src/fastslam2.rs  (with nalgebra 0.33.2's Matrix3::try_inverse and Cholesky restated
 *                                        from that crate's published source: it is a crates.io dependency,
 *                                        not part of /root/reference -- that part is parity-unpinned too)
 *
 * PARITY PINNING: the reference cannot be compiled here (no rustc/cargo in the
 * image) and its hot path is not seedable (rand::rng() at particle_filter.rs:258,443;
 * fastslam1.rs:129-130,220), and its own tests pin invariants, never values
 * (SURVEY.md section 4).  This restatement is therefore pinned by (a) every
 * reference-owned invariant re-expressed in tests/test_reference_invariants.py and
 * (b) the hand-derivable known answers KA1-KA7 of SURVEY.md Appendix B
 * (tests/test_known_answers.py).  RNG-level parity with rand/rand_distr is
 * UNPINNED by construction: all random draws enter as explicit arrays.
 *
 * Threads: the reference is single-threaded.  For bench.py's cpu_baseline ("embarrassingly
 * parallel stages run under OpenMP over particles on all host cores; serial stages -- cumsum, resample
 * walk -- stay serial") the per-particle loops carry `omp parallel for`; every particle's arithmetic and
 * every sum's order is unchanged, so results do not depend on the thread count.  ref_set_threads(1)
 * (the default) is the reference as written.
 *
 * Build: see oracle/Makefile (compiled with -ffp-contract=off -fopenmp).
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define REF_PI 3.14159265358979323846 /* std::f64::consts::PI */

/* ------------------------------------------------------------------ threads (bench.py cpu_baseline only) */
#ifdef _OPENMP
#include <omp.h>
#endif
static int g_ref_threads = 1;
/* n <= 0 => all host cores; returns the thread count now in use */
int ref_set_threads(int n) {
#ifdef _OPENMP
  if (n <= 0) n = omp_get_num_procs();
  g_ref_threads = n;
#else
  (void)n;
  g_ref_threads = 1;
#endif
  return g_ref_threads;
}
int ref_get_threads(void) { return g_ref_threads; }
#define REF_PARALLEL_FOR _Pragma("omp parallel for schedule(static) num_threads(g_ref_threads)")

/* ------------------------------------------------------------------ PF */

/* particle_filter.rs:476-479 */
double ref_gauss_likelihood(double x, double sigma) {
  double coeff = 1.0 / sqrt(2.0 * REF_PI * (sigma * sigma));
  return coeff * exp(-(x * x) / (2.0 * (sigma * sigma)));
}

/* particle_filter.rs:279-296.  nv/nw: per-particle noise samples already scaled
 * by sigma (Normal::new(0, sigma).sample), NULL => the `None` arm (exactly 0.0). */
void ref_pf_predict(size_t n, double* x, double* y, double* yaw, double* v,
                    double u0, double u1, double dt, const double* nv, const double* nw) {
  REF_PARALLEL_FOR
  for (size_t i = 0; i < n; ++i) {
    double v_noise = nv ? nv[i] : 0.0;
    double yaw_noise = nw ? nw[i] : 0.0;
    double v_noisy = u0 + v_noise;
    double yaw_rate_noisy = u1 + yaw_noise;
    x[i] += v_noisy * cos(yaw[i]) * dt;
    y[i] += v_noisy * sin(yaw[i]) * dt;
    yaw[i] += yaw_rate_noisy * dt;
    v[i] = v_noisy;
  }
}

/* particle_filter.rs:316-329: weight overwritten with the product of likelihoods */
void ref_pf_update_raw(size_t n, const double* x, const double* y, double* w,
                       const double* obs, size_t n_obs, double sigma) {
  REF_PARALLEL_FOR
  for (size_t i = 0; i < n; ++i) {
    double wi = 1.0;
    for (size_t l = 0; l < n_obs; ++l) {
      double d_obs = obs[3 * l], lx = obs[3 * l + 1], ly = obs[3 * l + 2];
      double dx = x[i] - lx;
      double dy = y[i] - ly;
      double d_pred = sqrt(dx * dx + dy * dy);
      double diff = d_obs - d_pred;
      wi *= ref_gauss_likelihood(diff, sigma);
    }
    w[i] = wi;
  }
}

/* particle_filter.rs:426-439; returns the serial sum it divided by */
double ref_pf_normalize(size_t n, double* w) {
  double sum_w = 0.0;
  for (size_t i = 0; i < n; ++i) sum_w += w[i];
  if (sum_w > 0.0) {
    for (size_t i = 0; i < n; ++i) w[i] /= sum_w;
  } else {
    double uniform = 1.0 / (double)n;
    for (size_t i = 0; i < n; ++i) w[i] = uniform;
  }
  return sum_w;
}

/* particle_filter.rs:416-423 */
double ref_pf_neff(size_t n, const double* w) {
  double s2 = 0.0;
  for (size_t i = 0; i < n; ++i) s2 += w[i] * w[i];
  return s2 > 0.0 ? 1.0 / s2 : 0.0;
}

/* particle_filter.rs:382-396 */
void ref_pf_estimate(size_t n, const double* x, const double* y, const double* yaw,
                     const double* v, const double* w, double out[4]) {
  double xe = 0.0, ye = 0.0, yawe = 0.0, ve = 0.0;
  for (size_t i = 0; i < n; ++i) {
    xe += w[i] * x[i];
    ye += w[i] * y[i];
    yawe += w[i] * yaw[i];
    ve += w[i] * v[i];
  }
  out[0] = xe; out[1] = ye; out[2] = yawe; out[3] = ve;
}

/* particle_filter.rs:398-413: cov += w * dx * dx^T (nalgebra evaluates (w*dx) * dx^T);
 * out row-major 4x4 */
void ref_pf_covariance(size_t n, const double* x, const double* y, const double* yaw,
                       const double* v, const double* w, const double est[4], double out[16]) {
  for (int k = 0; k < 16; ++k) out[k] = 0.0;
  for (size_t i = 0; i < n; ++i) {
    double d[4] = {x[i] - est[0], y[i] - est[1], yaw[i] - est[2], v[i] - est[3]};
    double wd[4] = {w[i] * d[0], w[i] * d[1], w[i] * d[2], w[i] * d[3]};
    for (int r = 0; r < 4; ++r)
      for (int c = 0; c < 4; ++c) out[4 * r + c] += wd[r] * d[c];
  }
}

/* particle_filter.rs:441-473 -- multinomial despite its doc comment: serial
 * inclusive cumsum, one uniform per output, first i with r <= c[i], DEFAULT 0.
 * Writes the chosen source indices; the caller gathers and sets w = 1/n. */
void ref_pf_resample_indices(size_t n, const double* w, const double* r, uint32_t* idx) {
  double* cum = (double*)malloc(n * sizeof(double));
  double cs = 0.0;
  for (size_t i = 0; i < n; ++i) {
    cs += w[i];
    cum[i] = cs;
  }
  for (size_t k = 0; k < n; ++k) {
    size_t index = 0;
    for (size_t i = 0; i < n; ++i) {
      if (r[k] <= cum[i]) {
        index = i;
        break;
      }
    }
    idx[k] = (uint32_t)index;
  }
  free(cum);
}

/* The same selection rule found by binary search (identical indices: cum is
 * non-decreasing) -- the "algorithmically equal" CPU variant timed at full N
 * because the literal O(N^2) scan above is infeasible at 1e6. */
void ref_pf_resample_indices_bsearch(size_t n, const double* w, const double* r, uint32_t* idx) {
  double* cum = (double*)malloc(n * sizeof(double));
  double cs = 0.0;
  for (size_t i = 0; i < n; ++i) {
    cs += w[i];
    cum[i] = cs;
  }
  for (size_t k = 0; k < n; ++k) {
    size_t lo = 0, hi = n;
    while (lo < hi) {
      size_t mid = lo + (hi - lo) / 2;
      if (r[k] <= cum[mid]) hi = mid; else lo = mid + 1;
    }
    idx[k] = (uint32_t)(lo < n ? lo : 0); /* default 0 */
  }
  free(cum);
}

/* monte_carlo_localization.rs:328-336,387-392 with min_particles == max_particles == n:
 * cumsum with the LAST entry forced to 1.0, first i with r <= c[i], fallback LAST. */
void ref_mcl_resample_indices(size_t n, const double* w, const double* r, uint32_t* idx) {
  double* cum = (double*)malloc(n * sizeof(double));
  double cs = 0.0;
  for (size_t i = 0; i < n; ++i) {
    cs += w[i];
    cum[i] = cs;
  }
  if (n) cum[n - 1] = 1.0;
  for (size_t k = 0; k < n; ++k) {
    size_t lo = 0, hi = n;
    while (lo < hi) {
      size_t mid = lo + (hi - lo) / 2;
      if (r[k] <= cum[mid]) hi = mid; else lo = mid + 1;
    }
    idx[k] = (uint32_t)(lo < n ? lo : n - 1);
  }
  free(cum);
}

/* monte_carlo_localization.rs:322-385, resample_adaptive with min_particles <= max_particles.
 * w = normalised weights of the n current particles, r = the uniforms the reference would have
 * drawn (at most max_p of them are consumed).  Writes the source index of every new particle to
 * idx (capacity max_p) and returns the new particle count.  The bin set is a plain list (the
 * reference's HashSet only answers "seen before?"). */
static int32_t ref_sat_i32(double v) {
  if (v != v) return 0;
  if (v <= -2147483648.0) return (int32_t)(-2147483647 - 1);
  if (v >= 2147483647.0) return 2147483647;
  return (int32_t)v;
}

size_t ref_kld_required(size_t k_bins, size_t min_p, size_t max_p, double eps, double z) {
  if (k_bins <= 1) return min_p; /* :368-370 */
  double k_minus_one = (double)(k_bins - 1);
  double term = 1.0 - 2.0 / (9.0 * k_minus_one) + z * sqrt(2.0 / (9.0 * k_minus_one));
  double n = (k_minus_one / (2.0 * eps)) * (term * term * term);
  double c = ceil(n);
  size_t r = !(c > 0.0) ? 0 : (c >= 18446744073709551615.0 ? (size_t)-1 : (size_t)c);
  if (r < min_p) r = min_p;
  if (r > max_p) r = max_p;
  return r;
}

size_t ref_mcl_resample_adaptive(size_t n, const double* x, const double* y, const double* yaw, const double* w,
                                 const double* r, size_t min_p, size_t max_p, double eps, double z, uint32_t* idx) {
  if (n == 0) return 0;
  double* cum = (double*)malloc(n * sizeof(double));
  double cs = 0.0;
  for (size_t i = 0; i < n; ++i) {
    cs += w[i];
    cum[i] = cs;
  }
  cum[n - 1] = 1.0; /* :334-336 */
  int32_t* bins = (int32_t*)malloc(3 * max_p * sizeof(int32_t));
  size_t k = 0, count = 0, required = min_p;
  const double yaw_bin = 15.0 * M_PI / 180.0;
  while (count < max_p) {
    size_t j = n - 1; /* sample_index :387-392: first i with r <= c[i], else last */
    for (size_t i = 0; i < n; ++i)
      if (r[count] <= cum[i]) { j = i; break; }
    int32_t b0 = ref_sat_i32(floor(x[j] / 0.5)), b1 = ref_sat_i32(floor(y[j] / 0.5)), b2 = ref_sat_i32(floor(yaw[j] / yaw_bin));
    int seen = 0;
    for (size_t q = 0; q < k && !seen; ++q) seen = bins[3 * q] == b0 && bins[3 * q + 1] == b1 && bins[3 * q + 2] == b2;
    if (!seen) {
      bins[3 * k] = b0; bins[3 * k + 1] = b1; bins[3 * k + 2] = b2;
      ++k;
    }
    size_t need = ref_kld_required(k, min_p, max_p, eps, z);
    if (need > required) required = need;
    idx[count++] = (uint32_t)j;
    if (count >= min_p && count >= required) break;
  }
  free(cum);
  free(bins);
  return count;
}

/* gather + uniform weights, particle_filter.rs:467-469 */
void ref_pf_gather(size_t n, double* x, double* y, double* yaw, double* v, double* w,
                   const uint32_t* idx) {
  double* t = (double*)malloc(4 * n * sizeof(double));
  REF_PARALLEL_FOR
  for (size_t k = 0; k < n; ++k) {
    t[k] = x[idx[k]];
    t[n + k] = y[idx[k]];
    t[2 * n + k] = yaw[idx[k]];
    t[3 * n + k] = v[idx[k]];
  }
  memcpy(x, t, n * sizeof(double));
  memcpy(y, t + n, n * sizeof(double));
  memcpy(yaw, t + 2 * n, n * sizeof(double));
  memcpy(v, t + 3 * n, n * sizeof(double));
  for (size_t k = 0; k < n; ++k) w[k] = 1.0 / (double)n;
  free(t);
}

/* One full PF step, particle_filter.rs:488-497: predict, update (+normalise),
 * N_eff-gated resample (:337-345).  r_draws: n uniforms used iff the gate fires.
 * scheme: 0 = PF (:441-473, gate n_eff < n*threshold, default index 0)
 *         1 = fixed-N MCL (monte_carlo_localization.rs:298, every step, fallback last)
 *         2 = the MCL step with the reference's SYSTEMATIC walk (fastslam1.rs:205-234; r0 = r_draws[0] / n) instead of
 *             its multinomial draws: the CPU counterpart of the engine's systematic option (bench.py's like-for-like baseline)
 * Returns 1 if it resampled.  est_out (4) = estimate after the step (Q15). */
int ref_pf_step_ex(size_t n, double* x, double* y, double* yaw, double* v, double* w,
                   double u0, double u1, double dt, const double* nv, const double* nw,
                   const double* obs, size_t n_obs, double sigma, double resample_threshold,
                   int scheme, const double* r_draws, uint32_t* idx_scratch, double* est_out, int literal_scan);

void ref_fs1_resample_indices(size_t n, double* pw, double r0, uint32_t* idx);

int ref_pf_step(size_t n, double* x, double* y, double* yaw, double* v, double* w,
                double u0, double u1, double dt, const double* nv, const double* nw,
                const double* obs, size_t n_obs, double sigma, double resample_threshold,
                int scheme, const double* r_draws, uint32_t* idx_scratch, double* est_out) {
  return ref_pf_step_ex(n, x, y, yaw, v, w, u0, u1, dt, nv, nw, obs, n_obs, sigma, resample_threshold, scheme, r_draws,
                        idx_scratch, est_out, 0);
}

/* literal_scan != 0: the PF resample walks the cumulative weights linearly for every draw exactly as
 * particle_filter.rs:455-470 does (O(N^2)); 0: the binary search that returns the same indices. */
int ref_pf_step_ex(size_t n, double* x, double* y, double* yaw, double* v, double* w,
                   double u0, double u1, double dt, const double* nv, const double* nw,
                   const double* obs, size_t n_obs, double sigma, double resample_threshold,
                   int scheme, const double* r_draws, uint32_t* idx_scratch, double* est_out, int literal_scan) {
  ref_pf_predict(n, x, y, yaw, v, u0, u1, dt, nv, nw);
  ref_pf_update_raw(n, x, y, w, obs, n_obs, sigma);
  ref_pf_normalize(n, w);
  int fired = 0;
  if (scheme == 1) {
    ref_mcl_resample_indices(n, w, r_draws, idx_scratch);
    ref_pf_gather(n, x, y, yaw, v, w, idx_scratch);
    fired = 1;
  } else if (scheme == 2) {
    ref_fs1_resample_indices(n, w, r_draws[0] / (double)n, idx_scratch);
    ref_pf_gather(n, x, y, yaw, v, w, idx_scratch);
    fired = 1;
  } else {
    double n_eff = ref_pf_neff(n, w);
    if (n_eff < (double)n * resample_threshold) {
      if (literal_scan) ref_pf_resample_indices(n, w, r_draws, idx_scratch);
      else ref_pf_resample_indices_bsearch(n, w, r_draws, idx_scratch);
      ref_pf_gather(n, x, y, yaw, v, w, idx_scratch);
      fired = 1;
    }
  }
  if (est_out) ref_pf_estimate(n, x, y, yaw, v, w, est_out);
  return fired;
}

/* K calls of try_step as the reference's callers make them (headless_localizers.rs:52-66, render_gif_particle_filter.rs:56-83),
 * timed INSIDE C with the monotonic clock: at the reference's own sizes (100 - 200 particles) a step costs ~10 us and a
 * foreign-function call per step would be a quarter of it.  Inputs per step k: controls[2k..], nv/nw[k n ..] (the scaled
 * normal samples), obs[k 3 n_obs ..], r_draws[k n ..].  with_cache_refresh != 0 also does what try_step really does between
 * the stages -- refresh_cache() = compute_estimate + compute_covariance after predict (particle_filter.rs:299), after
 * update (:332) and after a fired resample (:343); est_out (K x 4) receives the mean try_step returns (:496).  The RNG
 * (rand::rng() + Normal::sample per particle, :259-287) is NOT in the timed loop -- the samples are pre-drawn -- so this is a
 * lower bound of the reference's own cost.  Returns seconds. */
#include <time.h>
double ref_pf_try_step_loop(size_t n, double* x, double* y, double* yaw, double* v, double* w, const double* controls, double dt,
                            const double* nv, const double* nw, const double* obs, size_t n_obs, double sigma,
                            double resample_threshold, int scheme, const double* r_draws, uint32_t* idx_scratch, size_t K,
                            double* est_out, int with_cache_refresh, int literal_scan) {
  struct timespec t0, t1;
  double est[4], cov[16];
  volatile double sink = 0.0;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  for (size_t k = 0; k < K; ++k) {
    const double* o = obs + 3 * n_obs * k;
    ref_pf_predict(n, x, y, yaw, v, controls[2 * k], controls[2 * k + 1], dt, nv + n * k, nw + n * k);
    if (with_cache_refresh) {
      ref_pf_estimate(n, x, y, yaw, v, w, est);
      ref_pf_covariance(n, x, y, yaw, v, w, est, cov);
      sink += cov[0];
    }
    ref_pf_update_raw(n, x, y, w, o, n_obs, sigma);
    ref_pf_normalize(n, w);
    if (with_cache_refresh) {
      ref_pf_estimate(n, x, y, yaw, v, w, est);
      ref_pf_covariance(n, x, y, yaw, v, w, est, cov);
      sink += cov[0];
    }
    int fired = 0;
    if (scheme == 1) {
      ref_mcl_resample_indices(n, w, r_draws + n * k, idx_scratch);
      fired = 1;
    } else if (ref_pf_neff(n, w) < (double)n * resample_threshold) {
      if (literal_scan) ref_pf_resample_indices(n, w, r_draws + n * k, idx_scratch);
      else ref_pf_resample_indices_bsearch(n, w, r_draws + n * k, idx_scratch);
      fired = 1;
    }
    if (fired) ref_pf_gather(n, x, y, yaw, v, w, idx_scratch);
    if (fired || !with_cache_refresh) {
      ref_pf_estimate(n, x, y, yaw, v, w, est);
      if (with_cache_refresh) {
        ref_pf_covariance(n, x, y, yaw, v, w, est, cov);
        sink += cov[0];
      }
    }
    if (est_out)
      for (int q = 0; q < 4; ++q) est_out[4 * k + q] = est[q];
  }
  clock_gettime(CLOCK_MONOTONIC, &t1);
  (void)sink;
  return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}

/* ------------------------------------------------------------------ FastSLAM 1.0 */
/* Particle poses px,py,pyaw and weights pw are arrays of n; the per-particle
 * maps are particle-major AoS lm[(p*L + l)*6 + {x,y,c00,c10,c01,c11}], the memory
 * order of the reference's Vec<Landmark> with a column-major nalgebra Matrix2
 * (fastslam1.rs:26-31). */

typedef struct ref_fs1_model {
  double dt, q00, q11, r00, r11, init_threshold, init_cov;
} ref_fs1_model;

/* fastslam1.rs:13-23 */
void ref_fs1_model_default(ref_fs1_model* m) {
  m->dt = 0.1;
  m->q00 = 0.3;
  m->q11 = 0.0305;
  m->r00 = 0.5;
  m->r11 = 0.0305;
  m->init_threshold = 100.0;
  m->init_cov = NAN; /* reference leaves cov untouched on first observation (Q11) */
}

/* fastslam1.rs:80-89 */
double ref_normalize_angle(double a) {
  while (a > REF_PI) a -= 2.0 * REF_PI;
  while (a < -REF_PI) a += 2.0 * REF_PI;
  return a;
}

/* fastslam1.rs:302-306 with Particle::new :54-61 and Landmark::new :34-40 */
void ref_fs1_create(size_t n, size_t L, double* px, double* py, double* pyaw, double* pw, double* lm) {
  for (size_t p = 0; p < n; ++p) {
    px[p] = py[p] = pyaw[p] = 0.0;
    pw[p] = 1.0 / 100.0; /* 1/N_PARTICLE regardless of n (Q10) */
    for (size_t l = 0; l < L; ++l) {
      double* e = lm + (p * L + l) * 6;
      e[0] = 0.0; e[1] = 0.0;
      e[2] = 1000.0; e[3] = 0.0; e[4] = 0.0; e[5] = 1000.0;
    }
  }
}

/* fastslam1.rs:123-137 + 70-77; z0,z1 = unit normal draws per particle */
void ref_fs1_predict(size_t n, double* px, double* py, double* pyaw, double u0, double u1,
                     const double* z0, const double* z1, const ref_fs1_model* m) {
  REF_PARALLEL_FOR
  for (size_t p = 0; p < n; ++p) {
    double un0 = u0 + z0[p] * sqrt(m->q00);
    double un1 = u1 + z1[p] * sqrt(m->q11);
    double yaw = pyaw[p];
    px[p] = px[p] + un0 * m->dt * cos(yaw);
    py[p] = py[p] + un0 * m->dt * sin(yaw);
    pyaw[p] = ref_normalize_angle(yaw + un1 * m->dt);
  }
}

/* fastslam1.rs:140-183 for one particle and one observation; returns nothing,
 * multiplies *weight in place. */
void ref_fs1_update_landmark(double px, double py, double pyaw, double* weight,
                             double zd, double za, double* e, const ref_fs1_model* m) {
  if (e[2] > m->init_threshold) { /* :143-149 */
    e[0] = px + zd * cos(pyaw + za);
    e[1] = py + zd * sin(pyaw + za);
    if (!isnan(m->init_cov)) {
      e[2] = m->init_cov; e[3] = 0.0; e[4] = 0.0; e[5] = m->init_cov;
    }
    return;
  }
  double p00 = e[2], p10 = e[3], p01 = e[4], p11 = e[5];
  /* observation_model :92-99 */
  double dx = e[0] - px;
  double dy = e[1] - py;
  double d = sqrt(dx * dx + dy * dy);
  double zp_a = ref_normalize_angle(atan2(dy, dx) - pyaw);
  /* innovation :155 */
  double y0 = zd - d;
  double y1 = ref_normalize_angle(za - zp_a);
  /* compute_jacobian :102-110 */
  double d2 = dx * dx + dy * dy;
  double dd = sqrt(d2);
  double h00 = dx / dd, h01 = dy / dd, h10 = -dy / d2, h11 = dx / d2;
  /* s = h * cov * h^T + r :161, evaluated (h*cov)*h^T, entries summed in k order */
  double hp00 = h00 * p00 + h01 * p10, hp01 = h00 * p01 + h01 * p11;
  double hp10 = h10 * p00 + h11 * p10, hp11 = h10 * p01 + h11 * p11;
  double s00 = hp00 * h00 + hp01 * h01 + m->r00;
  double s01 = hp00 * h10 + hp01 * h11 + 0.0;
  double s10 = hp10 * h00 + hp11 * h01 + 0.0;
  double s11 = hp10 * h10 + hp11 * h11 + m->r11;
  /* s.try_inverse().unwrap_or(identity) :164 -- nalgebra 0.33 2x2: det = m11*m22 - m21*m12;
   * det == 0 => None */
  double det_inv = s00 * s11 - s10 * s01;
  double i00, i01, i10, i11;
  if (det_inv == 0.0) {
    i00 = 1.0; i01 = 0.0; i10 = 0.0; i11 = 1.0;
  } else {
    i00 = s11 / det_inv; i01 = -s01 / det_inv; i10 = -s10 / det_inv; i11 = s00 / det_inv;
  }
  /* k = cov * h^T * s_inv :165 */
  double pht00 = p00 * h00 + p01 * h01, pht01 = p00 * h10 + p01 * h11;
  double pht10 = p10 * h00 + p11 * h01, pht11 = p10 * h10 + p11 * h11;
  double k00 = pht00 * i00 + pht01 * i10, k01 = pht00 * i01 + pht01 * i11;
  double k10 = pht10 * i00 + pht11 * i10, k11 = pht10 * i01 + pht11 * i11;
  /* :168-170 */
  e[0] += k00 * y0 + k01 * y1;
  e[1] += k10 * y0 + k11 * y1;
  /* cov = (I - k*h) * cov :173-174 */
  double kh00 = k00 * h00 + k01 * h10, kh01 = k00 * h01 + k01 * h11;
  double kh10 = k10 * h00 + k11 * h10, kh11 = k10 * h01 + k11 * h11;
  double a00 = 1.0 - kh00, a01 = 0.0 - kh01, a10 = 0.0 - kh10, a11 = 1.0 - kh11;
  e[2] = a00 * p00 + a01 * p10;
  e[3] = a10 * p00 + a11 * p10;
  e[4] = a00 * p01 + a01 * p11;
  e[5] = a10 * p01 + a11 * p11;
  /* :177-182 */
  double det_s = s00 * s11 - s10 * s01;
  if (det_s > 0.0) {
    double t0 = y0 * i00 + y1 * i10; /* y^T * s_inv */
    double t1 = y0 * i01 + y1 * i11;
    double mahal = t0 * y0 + t1 * y1;
    double likelihood = exp(-0.5 * mahal) / (2.0 * REF_PI * sqrt(det_s));
    *weight *= likelihood;
  }
}

/* fastslam1.rs:196-203 (no fallback) */
double ref_fs1_normalize(size_t n, double* pw) {
  double s = 0.0;
  for (size_t p = 0; p < n; ++p) s += pw[p];
  if (s > 0.0)
    for (size_t p = 0; p < n; ++p) pw[p] /= s;
  return s;
}

/* fastslam1.rs:186-193 */
double ref_fs1_neff(size_t n, const double* pw) {
  double s2 = 0.0;
  for (size_t p = 0; p < n; ++p) s2 += pw[p] * pw[p];
  return s2 > 0.0 ? 1.0 / s2 : 0.0;
}

/* fastslam1.rs:205-234: indices of the systematic walk; r0 in [0, 1/n) */
void ref_fs1_resample_indices(size_t n, double* pw, double r0, uint32_t* idx) {
  ref_fs1_normalize(n, pw); /* :207 */
  double* cum = (double*)malloc((n + 1) * sizeof(double));
  cum[0] = 0.0;
  for (size_t i = 0; i < n; ++i) cum[i + 1] = cum[i] + pw[i];
  double r = r0;
  size_t j = 0;
  for (size_t k = 0; k < n; ++k) {
    while (r > cum[j + 1] && j < n - 1) j += 1;
    idx[k] = (uint32_t)j;
    r += 1.0 / (double)n;
  }
  free(cum);
}

/* clone step of fastslam1.rs:227-229 */
void ref_fs1_gather(size_t n, size_t L, double* px, double* py, double* pyaw, double* pw,
                    double* lm, const uint32_t* idx) {
  double* t = (double*)malloc((3 * n + n * L * 6) * sizeof(double));
  double* tl = t + 3 * n;
  REF_PARALLEL_FOR
  for (size_t k = 0; k < n; ++k) {
    size_t j = idx[k];
    t[k] = px[j]; t[n + k] = py[j]; t[2 * n + k] = pyaw[j];
    memcpy(tl + k * L * 6, lm + j * L * 6, L * 6 * sizeof(double));
  }
  memcpy(px, t, n * sizeof(double));
  memcpy(py, t + n, n * sizeof(double));
  memcpy(pyaw, t + 2 * n, n * sizeof(double));
  memcpy(lm, tl, n * L * 6 * sizeof(double));
  for (size_t k = 0; k < n; ++k) pw[k] = 1.0 / (double)n;
  free(t);
}

/* the observation loop of fastslam_update, fastslam1.rs:250-256: observation outer, particle inner (the particles
 * are independent of each other, so the inner loop may run under OpenMP) */
void ref_fs1_observe(size_t n, size_t L, const double* px, const double* py, const double* pyaw, double* pw, double* lm,
                     const double* z, size_t n_z, const ref_fs1_model* m) {
  for (size_t k = 0; k < n_z; ++k) {
    double zd = z[3 * k], za = z[3 * k + 1];
    size_t id = (size_t)z[3 * k + 2];
    REF_PARALLEL_FOR
    for (size_t p = 0; p < n; ++p)
      ref_fs1_update_landmark(px[p], py[p], pyaw[p], &pw[p], zd, za, lm + (p * L + id) * 6, m);
  }
}

/* fastslam1.rs:237-266.  z = n_z x (d, angle, id as double); nth = NTH (66.67 in the
 * reference); r0 used iff the gate fires.  Returns 1 if it resampled. */
int ref_fs1_update(size_t n, size_t L, double* px, double* py, double* pyaw, double* pw, double* lm,
                   double u0, double u1, const double* z0, const double* z1,
                   const double* z, size_t n_z, const ref_fs1_model* m, double nth, double r0,
                   uint32_t* idx_scratch) {
  ref_fs1_predict(n, px, py, pyaw, u0, u1, z0, z1, m);
  ref_fs1_observe(n, L, px, py, pyaw, pw, lm, z, n_z, m);
  ref_fs1_normalize(n, pw);
  double neff = ref_fs1_neff(n, pw);
  if (neff < nth) {
    ref_fs1_resample_indices(n, pw, r0, idx_scratch);
    ref_fs1_gather(n, L, px, py, pyaw, pw, lm, idx_scratch);
    return 1;
  }
  return 0;
}

/* fastslam1.rs:269-274: Iterator::max_by returns the LAST maximal element (Q14) */
size_t ref_fs1_best_particle(size_t n, const double* pw) {
  size_t best = 0;
  for (size_t p = 1; p < n; ++p)
    if (!(pw[p] < pw[best])) best = p; /* partial_cmp: >= keeps the later one */
  return best;
}

/* fastslam1.rs:277-299: observation simulator; zn = 2 unit normals per landmark;
 * out = (d, angle, id) rows; returns the count (range gate MAX_RANGE). */
size_t ref_fs1_get_observations(const double xt[3], const double* lms, size_t L, double max_range,
                                const double* zn, const ref_fs1_model* m, double* out) {
  size_t cnt = 0;
  for (size_t l = 0; l < L; ++l) {
    double dx = lms[2 * l] - xt[0];
    double dy = lms[2 * l + 1] - xt[1];
    double d = sqrt(dx * dx + dy * dy);
    if (d <= max_range) {
      double angle = ref_normalize_angle(atan2(dy, dx) - xt[2]);
      out[3 * cnt] = d + zn[2 * l] * sqrt(m->r00);
      out[3 * cnt + 1] = angle + zn[2 * l + 1] * sqrt(m->r11);
      out[3 * cnt + 2] = (double)l;
      ++cnt;
    }
  }
  return cnt;
}

/* ------------------------------------------------------------------ FastSLAM 2.0 */
/* rust_robotics_slam/src/fastslam2.rs.  3x3 matrices are row-major arrays of 9; matrix products
 * are evaluated as nalgebra 0.33 does for small static matrices (each entry summed over k in
 * ascending order, no FMA); Matrix3::try_inverse is nalgebra's cofactor formula (linalg/inverse.rs),
 * Cholesky is linalg/cholesky.rs (lower triangle, None when a pivot is zero or negative).
 * nalgebra itself is a crates.io dependency and not part of /root/reference. */

static void ref_mul33(const double* a, const double* b, double* o) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) o[3 * i + j] = a[3 * i] * b[j] + a[3 * i + 1] * b[3 + j] + a[3 * i + 2] * b[6 + j];
}

static int ref_inv3(const double* a, double* o) {
  double m11 = a[0], m12 = a[1], m13 = a[2], m21 = a[3], m22 = a[4], m23 = a[5], m31 = a[6], m32 = a[7], m33 = a[8];
  double minor_m12_m23 = m22 * m33 - m32 * m23;
  double minor_m11_m23 = m21 * m33 - m31 * m23;
  double minor_m11_m22 = m21 * m32 - m31 * m22;
  double det = m11 * minor_m12_m23 - m12 * minor_m11_m23 + m13 * minor_m11_m22;
  if (det == 0.0) return 0;
  o[0] = minor_m12_m23 / det;
  o[1] = (m13 * m32 - m33 * m12) / det;
  o[2] = (m12 * m23 - m22 * m13) / det;
  o[3] = -minor_m11_m23 / det;
  o[4] = (m11 * m33 - m31 * m13) / det;
  o[5] = (m13 * m21 - m23 * m11) / det;
  o[6] = minor_m11_m22 / det;
  o[7] = (m12 * m31 - m32 * m11) / det;
  o[8] = (m11 * m22 - m21 * m12) / det;
  return 1;
}

/* motion_model :92-99 */
static void ref_fs2_motion_model(const double x[3], double u0, double u1, double o[3]) {
  double yaw = x[2];
  o[0] = x[0] + u0 * 0.1 * cos(yaw);
  o[1] = x[1] + u0 * 0.1 * sin(yaw);
  o[2] = ref_normalize_angle(x[2] + u1 * 0.1);
}

/* compute_proposal :173-216; lm = {x, y, c00, c10, c01, c11}; r = diag(r00, r11) */
void ref_fs2_proposal(const double pose[3], double u0, double u1, double zd, double za, const double* lm, double r00,
                      double r11, double mean[3], double cov[9]) {
  const double DT = 0.1;
  double xp[3];
  ref_fs2_motion_model(pose, u0, u1, xp);
  double yaw = pose[2], v = u0;
  double g[9] = {1.0, 0.0, -v * DT * sin(yaw), 0.0, 1.0, v * DT * cos(yaw), 0.0, 0.0, 1.0}; /* :103-118 */
  double mc[9] = {0.1, 0.0, 0.0, 0.0, 0.1, 0.0, 0.0, 0.0, 0.01};                              /* :30 */
  double gt[9] = {g[0], g[3], g[6], g[1], g[4], g[7], g[2], g[5], g[8]};
  double gm[9], P[9];
  ref_mul33(g, mc, gm);
  ref_mul33(gm, gt, P); /* :183 */
  double c00 = lm[2], c10 = lm[3], c01 = lm[4], c11 = lm[5];
  if (!(c00 < 100.0)) { /* is_initialized :49-51 */
    memcpy(mean, xp, 3 * sizeof(double));
    memcpy(cov, P, 9 * sizeof(double));
    return;
  }
  double dx = lm[0] - xp[0], dy = lm[1] - xp[1];
  double d2 = dx * dx + dy * dy;
  double d = sqrt(d2);
  double h[6] = {-dx / d, -dy / d, 0.0, dy / d2, -dx / d2, -1.0}; /* :139-147 */
  double l00 = dx / d, l01 = dy / d, l10 = -dy / d2, l11 = dx / d2; /* :131-137 */
  double hc00 = l00 * c00 + l01 * c10, hc01 = l00 * c01 + l01 * c11;
  double hc10 = l10 * c00 + l11 * c10, hc11 = l10 * c01 + l11 * c11;
  double q00 = hc00 * l00 + hc01 * l01 + r00, q01 = hc00 * l10 + hc01 * l11 + 0.0;
  double q10 = hc10 * l00 + hc11 * l01 + 0.0, q11 = hc10 * l10 + hc11 * l11 + r11; /* :195 */
  double qdet = q00 * q11 - q10 * q01;
  double i00, i01, i10, i11;
  if (qdet == 0.0) { i00 = 1.0; i01 = 0.0; i10 = 0.0; i11 = 1.0; }
  else { i00 = q11 / qdet; i01 = -q01 / qdet; i10 = -q10 / qdet; i11 = q00 / qdet; } /* :200 */
  double Pinv[9];
  if (!ref_inv3(P, Pinv)) { /* :202 */
    memset(Pinv, 0, sizeof Pinv);
    Pinv[0] = Pinv[4] = Pinv[8] = 1.0 * 1e-6;
  }
  double t[6]; /* h_pose^T * q_obs_inv, 3x2 */
  for (int r = 0; r < 3; ++r) {
    t[2 * r] = h[r] * i00 + h[3 + r] * i10;
    t[2 * r + 1] = h[r] * i01 + h[3 + r] * i11;
  }
  double Ppi[9];
  for (int r = 0; r < 3; ++r)
    for (int q = 0; q < 3; ++q) Ppi[3 * r + q] = Pinv[3 * r + q] + (t[2 * r] * h[q] + t[2 * r + 1] * h[3 + q]); /* :203 */
  double Pp[9];
  if (!ref_inv3(Ppi, Pp)) memcpy(Pp, P, sizeof Pp); /* :204 */
  double zp_a = ref_normalize_angle(atan2(dy, dx) - xp[2]); /* :207, observation_model :122-128 */
  double y0 = zd - d, y1 = ref_normalize_angle(za - zp_a);
  for (int r = 0; r < 3; ++r) { /* :210, ((p_post * h^T) * q_inv) * innovation */
    double ph0 = Pp[3 * r] * h[0] + Pp[3 * r + 1] * h[1] + Pp[3 * r + 2] * h[2];
    double ph1 = Pp[3 * r] * h[3] + Pp[3 * r + 1] * h[4] + Pp[3 * r + 2] * h[5];
    double k0 = ph0 * i00 + ph1 * i10, k1 = ph0 * i01 + ph1 * i11;
    mean[r] = xp[r] + (k0 * y0 + k1 * y1);
  }
  memcpy(cov, Pp, 9 * sizeof(double));
}

/* sample_pose_with_rng :219-239 followed by set_pose :77-81; z = three unit normals */
void ref_fs2_sample(const double mean[3], const double c[9], const double z[3], double pose[3]) {
  double L[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  int ok = 0;
  if (c[0] != 0.0 && c[0] >= 0.0) {
    double l00 = sqrt(c[0]), l10 = c[3] / l00, l20 = c[6] / l00;
    double d1 = -l10 * l10 + c[4];
    double c21 = -l10 * l20 + c[7];
    if (d1 != 0.0 && d1 >= 0.0) {
      double l11 = sqrt(d1), l21 = c21 / l11;
      double d2 = -l21 * l21 + (-l20 * l20 + c[8]);
      if (d2 != 0.0 && d2 >= 0.0) {
        L[0] = l00; L[3] = l10; L[4] = l11; L[6] = l20; L[7] = l21; L[8] = sqrt(d2);
        ok = 1;
      }
    }
  }
  if (!ok) { /* :227-233 */
    L[0] = sqrt(fmax(c[0], 0.0));
    L[4] = sqrt(fmax(c[4], 0.0));
    L[8] = sqrt(fmax(c[8], 0.0));
  }
  pose[0] = mean[0] + (L[0] * z[0] + L[1] * z[1] + L[2] * z[2]);
  pose[1] = mean[1] + (L[3] * z[0] + L[4] * z[1] + L[5] * z[2]);
  pose[2] = ref_normalize_angle(mean[2] + (L[6] * z[0] + L[7] * z[1] + L[8] * z[2]));
}

/* update_landmark_and_weight :242-291; returns the factor the particle weight is multiplied by */
double ref_fs2_update_landmark(double px, double py, double pyaw, double zd, double za, double* e, double r00, double r11) {
  if (!(e[2] < 100.0)) { /* :251-257 */
    e[0] = px + zd * cos(pyaw + za);
    e[1] = py + zd * sin(pyaw + za);
    e[2] = 10.0; e[3] = 0.0; e[4] = 0.0; e[5] = 10.0;
    return 1.0;
  }
  double w = 1.0;
  ref_fs1_model m;
  ref_fs1_model_default(&m);
  m.r00 = r00;
  m.r11 = r11;
  m.init_threshold = INFINITY; /* the branch above already decided */
  /* :259-283 are line for line fastslam1.rs:152-182 (same Jacobian, S, K, (I - K H) P, likelihood) */
  /* det S is needed here because the weight of the det <= 0 case differs (1e-10, :288-290) */
  double p00 = e[2], p10 = e[3], p01 = e[4], p11 = e[5];
  double dx = e[0] - px, dy = e[1] - py;
  double d2 = dx * dx + dy * dy, dd = sqrt(d2);
  double h00 = dx / dd, h01 = dy / dd, h10 = -dy / d2, h11 = dx / d2;
  double hp00 = h00 * p00 + h01 * p10, hp01 = h00 * p01 + h01 * p11;
  double hp10 = h10 * p00 + h11 * p10, hp11 = h10 * p01 + h11 * p11;
  double s00 = hp00 * h00 + hp01 * h01 + r00, s01 = hp00 * h10 + hp01 * h11 + 0.0;
  double s10 = hp10 * h00 + hp11 * h01 + 0.0, s11 = hp10 * h10 + hp11 * h11 + r11;
  double det_s = s00 * s11 - s10 * s01;
  ref_fs1_update_landmark(px, py, pyaw, &w, zd, za, e, &m);
  if (!(det_s > 0.0)) return 1e-10; /* :288-290 */
  return w;
}

/* fastslam2_update_with_rng :331-374.  noise = 3 unit normals per particle (the proposal sample,
 * or the two motion normals when z is empty); r0 = the resample offset in [0, 1/n). */
int ref_fs2_update(size_t n, size_t L, double* px, double* py, double* pyaw, double* pw, double* lm, double u0, double u1,
                   const double* noise, const double* z, size_t n_z, double nth, double r0, uint32_t* idx_scratch) {
  const double r00 = 0.5, r11 = 0.0305;
  REF_PARALLEL_FOR
  for (size_t p = 0; p < n; ++p) {
    double pose[3] = {px[p], py[p], pyaw[p]};
    if (n_z > 0) { /* :341-347 */
      double mean[3], cov[9], np[3];
      size_t id = (size_t)z[2];
      ref_fs2_proposal(pose, u0, u1, z[0], z[1], lm + (p * L + id) * 6, r00, r11, mean, cov);
      ref_fs2_sample(mean, cov, noise + 3 * p, np);
      px[p] = np[0]; py[p] = np[1]; pyaw[p] = np[2];
    } else { /* :349-357 */
      double un0 = u0 + noise[3 * p] * sqrt(0.3), un1 = u1 + noise[3 * p + 1] * sqrt(0.0305);
      double np[3];
      ref_fs2_motion_model(pose, un0, un1, np);
      px[p] = np[0]; py[p] = np[1]; pyaw[p] = ref_normalize_angle(np[2]);
    }
    for (size_t k = 0; k < n_z; ++k) /* :361-365 */
      pw[p] *= ref_fs2_update_landmark(px[p], py[p], pyaw[p], z[3 * k], z[3 * k + 1], lm + (p * L + (size_t)z[3 * k + 2]) * 6, r00, r11);
  }
  ref_fs1_normalize(n, pw); /* :368, same as fastslam1.rs:196-203 */
  double neff = ref_fs1_neff(n, pw);
  if (neff < nth) { /* :370-373; resample_with_rng :303-328 is fastslam1.rs:205-234 */
    ref_fs1_resample_indices(n, pw, r0, idx_scratch);
    ref_fs1_gather(n, L, px, py, pyaw, pw, lm, idx_scratch);
    return 1;
  }
  return 0;
}
