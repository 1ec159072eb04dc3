// C++ consumer of include/rust_robotics.hpp: the reference-shaped wrapper classes driven the way the
// reference's own tests drive its types (particle_filter.rs:575-707, monte_carlo_localization.rs:489-577,
// fastslam1.rs:308-401).  Built by tests/test_gpu_c_abi.py with g++ and run on the GPU.
#include <cmath>
#include <cstdio>
#include <random>

#include "rust_robotics.hpp"

#define REQUIRE(cond)                                                  \
  do {                                                                 \
    if (!(cond)) {                                                     \
      std::fprintf(stderr, "%s:%d: %s failed\n", __FILE__, __LINE__, #cond); \
      return 1;                                                        \
    }                                                                  \
  } while (0)

static rr::PFMeasurement observe(const double truth[3], std::mt19937_64& rng, double sigma) {
  static const double lm[4][2] = {{10, 0}, {0, 15}, {-5, 20}, {10, 10}};  // unified_filter_comparison.rs:43
  std::normal_distribution<double> noise(0.0, sigma);
  rr::PFMeasurement z;
  for (auto& l : lm) z.emplace_back(std::max(0.0, std::hypot(l[0] - truth[0], l[1] - truth[1]) + noise(rng)), l[0], l[1]);
  return z;
}

int main() {
  std::mt19937_64 rng(7);
  // ---- ParticleFilterLocalizer: config validation, the step loop, estimate / covariance / particles
  rr::ParticleFilterConfig bad;
  bad.n_particles = 0;
  bool threw = false;
  try {
    rr::ParticleFilterLocalizer pf(bad);
  } catch (const rr::RoboticsError& e) {
    threw = e.kind == rr::RoboticsError::InvalidParameter && std::string(e.what()).find("at least one particle") != std::string::npos;
  }
  REQUIRE(threw);  // particle_filter.rs:649-660
  rr::ParticleFilterConfig cfg;
  cfg.n_particles = 4000;
  cfg.range_noise = 0.5;
  cfg.velocity_noise = 0.3;
  cfg.yaw_rate_noise = 5.0 * 3.14159265358979323846 / 180.0;
  rr::ParticleFilterLocalizer pf(cfg, 42);
  pf.try_set_landmarks({{10, 0}, {0, 15}});
  REQUIRE(pf.get_landmarks().size() == 2);
  double truth[3] = {0, 0, 0};
  rr::PFState est{};
  for (int t = 0; t < 60; ++t) {
    truth[0] += 1.0 * std::cos(truth[2]) * cfg.dt;
    truth[1] += 1.0 * std::sin(truth[2]) * cfg.dt;
    truth[2] += 0.1 * cfg.dt;
    est = pf.try_step({1.0, 0.1}, observe(truth, rng, 0.5));
    for (double v : est) REQUIRE(std::isfinite(v));
  }
  REQUIRE(std::hypot(est[0] - truth[0], est[1] - truth[1]) < 1.0);
  auto cov = pf.calc_covariance();
  for (int k = 0; k < 4; ++k) REQUIRE(cov[5 * k] >= 0.0);  // :639-646
  auto parts = pf.get_particles();
  double wsum = 0.0;
  for (auto& p : parts) wsum += p.w;
  REQUIRE(parts.size() == 4000 && std::fabs(wsum - 1.0) < 1e-3);  // :611-623
  threw = false;
  try {
    pf.try_update_with_observations({{-1.0, 0.0, 0.0}});  // negative distance, :538-549
  } catch (const rr::RoboticsError& e) {
    threw = e.kind == rr::RoboticsError::InvalidParameter;
  }
  REQUIRE(threw);
  // StateEstimator shape: predict(dt ignored) + update
  pf.predict({1.0, 0.1}, 123.0);
  pf.update(observe(truth, rng, 0.5));
  REQUIRE(std::isfinite(pf.get_state()[0]));

  // ---- MonteCarloLocalizer: fixed N and KLD-adaptive
  rr::MonteCarloLocalizationConfig mc;
  mc.min_particles = mc.max_particles = 3000;
  mc.range_noise = 0.5;
  rr::MonteCarloLocalizer mcl(mc, 5);
  double tm[3] = {0, 0, 0};
  for (int t = 0; t < 40; ++t) {
    tm[0] += std::cos(tm[2]) * mc.dt;
    tm[1] += std::sin(tm[2]) * mc.dt;
    tm[2] += 0.1 * mc.dt;
    est = mcl.try_step({1.0, 0.1}, observe(tm, rng, 0.5));
  }
  REQUIRE(mcl.particle_count() == 3000 && std::hypot(est[0] - tm[0], est[1] - tm[1]) < 1.0);  // monte_carlo_localization.rs:489-516
  rr::MonteCarloLocalizationConfig ad;
  ad.min_particles = 200;
  ad.max_particles = 3000;
  ad.range_noise = 0.5;
  rr::MonteCarloLocalizer amcl(ad, 6);
  double ta[3] = {0, 0, 0};
  for (int t = 0; t < 20; ++t) {
    ta[0] += std::cos(ta[2]) * ad.dt;
    ta[1] += std::sin(ta[2]) * ad.dt;
    ta[2] += 0.1 * ad.dt;
    amcl.try_step({1.0, 0.1}, observe(ta, rng, 0.5));
    REQUIRE(amcl.particle_count() >= 200 && amcl.particle_count() <= 3000);  // :518-545
  }

  // ---- the reference's own sizes (150 particles, render_gif_particle_filter.rs:77-79): a batch of steps in one launch must
  // return exactly what single synchronous steps return
  {
    rr::ParticleFilterConfig sc;
    sc.n_particles = 150;
    sc.range_noise = 0.5;
    rr::ParticleFilterLocalizer a(sc, 11), b(sc, 11);
    std::vector<rr::PFControl> us;
    std::vector<rr::PFMeasurement> zs;
    double tr[3] = {0, 0, 0};
    for (int t = 0; t < 30; ++t) {
      tr[0] += std::cos(tr[2]) * sc.dt;
      tr[1] += std::sin(tr[2]) * sc.dt;
      tr[2] += 0.1 * sc.dt;
      us.push_back({1.0, 0.1});
      zs.push_back(observe(tr, rng, 0.5));
    }
    const std::vector<rr::PFState> many = a.try_step_many(us, zs);
    REQUIRE(many.size() == 30);
    for (int t = 0; t < 30; ++t) {
      const rr::PFState one = b.try_step(us[t], zs[t]);
      for (int k = 0; k < 4; ++k) REQUIRE(many[t][k] == one[k]);
    }
  }

  // ---- engine extensions through the wrapper: the resident service answers try_step with the launched steps' bits, and the
  // asynchronous step + last_step_estimate return what try_step returns
  {
    rr::ParticleFilterConfig sc;
    sc.n_particles = 150;
    sc.range_noise = 0.5;
    rr::ParticleFilterLocalizer a(sc, 12), b(sc, 12), c(sc, 12);
    b.set_resident(20000.0);
    double tr[3] = {0, 0, 0};
    for (int t = 0; t < 40; ++t) {
      tr[0] += std::cos(tr[2]) * sc.dt;
      tr[1] += std::sin(tr[2]) * sc.dt;
      tr[2] += 0.1 * sc.dt;
      const rr::PFMeasurement z = observe(tr, rng, 0.5);
      const rr::PFState ea = a.try_step({1.0, 0.1}, z), eb = b.try_step({1.0, 0.1}, z);
      c.try_step_async({1.0, 0.1}, z, /*with_estimate=*/true);
      const rr::PFState ec = c.last_step_estimate();
      for (int k = 0; k < 4; ++k) REQUIRE(ea[k] == eb[k] && ea[k] == ec[k]);
    }
    c.synchronize();
  }

  // ---- fastslam1 / fastslam2
  rr::fastslam1::Params prm;
  prm.first_obs_cov = 10.0;
  rr::fastslam1::FastSlam1 fs(500, 3, prm, 9);
  for (int t = 0; t < 10; ++t) fs.update({1.0, 0.1}, {{5.0, 0.1, 0}, {7.0, -0.4, 2}});
  auto [pose, w, i] = fs.best_particle();
  REQUIRE(i < 500 && std::isfinite(pose[0]) && w > 0.0);
  {  // resident service + asynchronous update through the wrapper: the same best particle as the launched updates
    rr::fastslam1::FastSlam1 fa(100, 3, prm, 10), fb(100, 3, prm, 10);
    fb.set_resident(20000.0);
    for (int t = 0; t < 12; ++t) {
      fa.update_async({1.0, 0.1}, {{5.0, 0.1, 0}, {7.0, -0.4, 2}});
      fb.update({1.0, 0.1}, {{5.0, 0.1, 0}, {7.0, -0.4, 2}});
      auto [pa, wa, ia] = fa.best_particle();
      auto [pb, wb, ib] = fb.best_particle();
      REQUIRE(ia == ib && wa == wb && pa[0] == pb[0] && pa[1] == pb[1] && pa[2] == pb[2]);
    }
    fa.synchronize();
  }
  auto lms = fs.landmarks_of(i);
  REQUIRE(lms.size() == 18 && lms[2] < 100.0 && lms[6 + 2] == 1000.0);  // landmark 1 never observed: cov stays 1000 I (fastslam1.rs:34-40)
  rr::fastslam2::FastSlam2 f2(400, 2, {}, 3);
  for (int t = 0; t < 5; ++t) f2.update({1.0, 0.1}, {{5.0, 0.1, 0}, {6.0, 0.5, 1}});
  REQUIRE(std::isfinite(std::get<1>(f2.best_particle())));
  std::printf("HPP_OK est=(%.3f, %.3f) truth=(%.3f, %.3f) adaptive_n=%llu\n", est[0], est[1], tm[0], tm[1],
              (unsigned long long)amcl.particle_count());
  return 0;
}
