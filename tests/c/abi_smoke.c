/* Plain-C consumer of the boundary: what a cgo/JNI/Rust `-sys` binding sees.  Built by
 * tests/test_gpu_c_abi.py with gcc against include/ and librust_robotics_amd.so, run on the GPU. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "rr_fastslam1.h"
#include "rr_pf.h"

#define CHECK(call)                                                            \
  do {                                                                         \
    rr_status s_ = (call);                                                     \
    if (s_ != RR_OK) {                                                         \
      fprintf(stderr, "%s -> %d: %s\n", #call, (int)s_, rr_last_error());      \
      return 1;                                                                \
    }                                                                          \
  } while (0)

int main(void) {
  printf("%s devices=%d\n", rr_version(), rr_device_count());
  rr_pf_config cfg;
  rr_pf_config_default(&cfg);
  cfg.n_particles = 5000;
  cfg.range_noise = 0.5;
  cfg.velocity_noise = 0.3;
  cfg.yaw_rate_noise = 5.0 * 3.14159265358979323846 / 180.0;
  rr_pf_options opt;
  rr_pf_options_default(&opt);
  opt.seed = 42;
  rr_pf* pf = NULL;
  CHECK(rr_pf_create(&cfg, &opt, &pf));
  const double lm[8] = {10, 0, 0, 15, -5, 20, 10, 10};
  CHECK(rr_pf_set_landmarks(pf, lm, 4));
  double truth[3] = {0, 0, 0}, u[2] = {1.0, 0.1}, est[4];
  for (int t = 0; t < 60; ++t) {
    truth[0] += u[0] * cos(truth[2]) * cfg.dt;
    truth[1] += u[0] * sin(truth[2]) * cfg.dt;
    truth[2] += u[1] * cfg.dt;
    double obs[12];
    for (int l = 0; l < 4; ++l) {
      obs[3 * l] = hypot(lm[2 * l] - truth[0], lm[2 * l + 1] - truth[1]);
      obs[3 * l + 1] = lm[2 * l];
      obs[3 * l + 2] = lm[2 * l + 1];
    }
    CHECK(rr_pf_step(pf, u, obs, 4, est));
  }
  double err = hypot(est[0] - truth[0], est[1] - truth[1]);
  printf("pf estimate (%.3f, %.3f) truth (%.3f, %.3f) err %.3f\n", est[0], est[1], truth[0], truth[1], err);
  if (!(err < 1.0)) return 2;
  double cov[16];
  CHECK(rr_pf_covariance(pf, cov));
  if (!(cov[0] >= 0 && cov[5] >= 0)) return 3;
  /* error behaviour: the reference's InvalidParameter message */
  double bad[3] = {-1.0, 0.0, 0.0};
  if (rr_pf_update(pf, bad, 1) != RR_INVALID_PARAMETER) return 4;
  if (strstr(rr_last_error(), "non-negative distances") == NULL) return 5;
  rr_pf_destroy(pf);

  rr_fs1_params prm;
  rr_fs1_params_default(&prm);
  prm.first_obs_cov = 2.0;
  rr_fs1_options fo;
  rr_fs1_options_default(&fo);
  fo.seed = 7;
  rr_fs1* fs = NULL;
  CHECK(rr_fs1_create(300, 3, &prm, &fo, &fs));
  const double fl[6] = {10, 0, 0, 10, 10, 10};
  double xt[3] = {0, 0, 0};
  for (uint32_t t = 0; t < 10; ++t) {
    double z[9];
    size_t nz = rr_fs1_get_observations(xt, fl, 3, &prm, 7, t, z, 3);
    CHECK(rr_fs1_update(fs, u, z, nz));
  }
  double pose[3], w;
  uint64_t idx;
  CHECK(rr_fs1_best_particle(fs, pose, &w, &idx));
  printf("fastslam best particle %llu weight %.4g pose (%.3f, %.3f, %.3f)\n", (unsigned long long)idx, w, pose[0], pose[1], pose[2]);
  if (!isfinite(pose[0]) || !(w > 0)) return 6;
  rr_fs1_destroy(fs);
  printf("C_ABI_OK\n");
  return 0;
}
