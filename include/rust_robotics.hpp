// rust_robotics.hpp -- header-only C++17 wrapper over the C ABI (include/rr_pf.h, include/rr_fastslam1.h)
// with the reference's names: ParticleFilterLocalizer / MonteCarloLocalizer
// (rust_robotics_localization/src/particle_filter.rs:121-573, monte_carlo_localization.rs:136-462) and
// fastslam1 (rust_robotics_slam/src/fastslam1.rs).  try_* methods throw rr::RoboticsError where the
// reference returns Err(RoboticsError::InvalidParameter(..)).  Link with -lrust_robotics_amd.
#pragma once

#include <array>
#include <cmath>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <tuple>
#include <utility>
#include <vector>

#include "rr_fastslam2.h"
#include "rr_pf.h"

namespace rr {

struct RoboticsError : std::runtime_error {
  enum Kind { InvalidParameter, NumericalError } kind;
  RoboticsError(Kind k, const std::string& m) : std::runtime_error(m), kind(k) {}
};

inline void check(rr_status s) {
  if (s == RR_OK) return;
  throw RoboticsError(s == RR_INVALID_PARAMETER ? RoboticsError::InvalidParameter : RoboticsError::NumericalError,
                      rr_last_error());
}

struct Point2D { double x = 0, y = 0; };                       // core/src/types.rs:17-20
struct State2D { double x = 0, y = 0, yaw = 0, v = 0; };       // core/src/types.rs:141-146
struct ControlInput { double v = 0, omega = 0; };              // core/src/types.rs:189-192
struct Particle { double x, y, yaw, v, w; };                   // particle_filter.rs:25-32
struct Obstacles {                                             // core/src/types.rs:344-346
  std::vector<Point2D> points;
  static Obstacles from_points(std::vector<Point2D> p) { return Obstacles{std::move(p)}; }
};
using PFState = std::array<double, 4>;
using PFControl = std::array<double, 2>;
using PFMeasurement = std::vector<std::tuple<double, double, double>>;  // (d, landmark_x, landmark_y)

struct ParticleFilterConfig : rr_pf_config {                   // particle_filter.rs:51-78
  ParticleFilterConfig() { rr_pf_config_default(this); }
  void validate() const { check(rr_pf_config_validate(this)); }
};

class ParticleFilterLocalizer {
 public:
  explicit ParticleFilterLocalizer(const ParticleFilterConfig& cfg = {}, uint64_t seed = 0, int device = 0) {
    rr_pf_options o;
    options(&o);
    o.seed = seed;
    o.device = device;
    check(rr_pf_create(&cfg, &o, &h_));
  }
  ParticleFilterLocalizer(const PFState& initial, const ParticleFilterConfig& cfg, uint64_t seed = 0, int device = 0) {
    rr_pf_options o;
    options(&o);
    o.seed = seed;
    o.device = device;
    check(rr_pf_create_with_state(&cfg, &o, initial.data(), &h_));
  }
  static ParticleFilterLocalizer with_defaults() { return ParticleFilterLocalizer(); }
  // with_initial_state_2d, particle_filter.rs:202-207
  static ParticleFilterLocalizer with_initial_state_2d(const State2D& s, const ParticleFilterConfig& cfg, uint64_t seed = 0, int device = 0) {
    return ParticleFilterLocalizer(PFState{s.x, s.y, s.yaw, s.v}, cfg, seed, device);
  }
  // set_landmarks_from_obstacles, :223-225
  void set_landmarks_from_obstacles(const Obstacles& o) { try_set_landmarks(o.points); }
  ParticleFilterLocalizer(ParticleFilterLocalizer&& o) noexcept : h_(std::exchange(o.h_, nullptr)) {}
  ParticleFilterLocalizer(const ParticleFilterLocalizer&) = delete;
  virtual ~ParticleFilterLocalizer() { rr_pf_destroy(h_); }

  void try_set_landmarks(const std::vector<Point2D>& lm) {
    check(rr_pf_set_landmarks(h_, lm.empty() ? nullptr : &lm[0].x, lm.size()));
  }
  std::vector<Point2D> get_landmarks() const {
    std::vector<Point2D> out(rr_pf_landmark_count(h_));
    if (!out.empty()) rr_pf_get_landmarks(h_, &out[0].x, out.size());
    return out;
  }
  void set_range_noise(double s) { check(rr_pf_set_range_noise(h_, s)); }
  void try_predict_with_control(const PFControl& u) { check(rr_pf_predict(h_, u.data())); }
  void try_predict_input(const ControlInput& c) { try_predict_with_control({c.v, c.omega}); }
  void try_update_with_observations(const PFMeasurement& obs) {
    std::vector<double> flat = flatten(obs);
    check(rr_pf_update(h_, flat.data(), obs.size()));
  }
  void resample() { check(rr_pf_resample(h_)); }
  PFState try_step(const PFControl& u, const PFMeasurement& obs) {
    std::vector<double> flat = flatten(obs);
    PFState out;
    check(rr_pf_step(h_, u.data(), flat.data(), obs.size(), out.data()));
    return out;
  }
  // engine extension: controls.size() steps in one call (rr_pf_step_many; up to 2048 particles: ONE kernel launch for all of
  // them); obs[k] = the observations of step k, the same number every step; returns what try_step returns after each step
  std::vector<PFState> try_step_many(const std::vector<PFControl>& controls, const std::vector<PFMeasurement>& obs) {
    if (controls.size() != obs.size()) throw RoboticsError(RoboticsError::InvalidParameter, "one observation list per control");
    const size_t n_obs = obs.empty() ? 0 : obs[0].size();
    std::vector<double> u, flat;
    for (size_t k = 0; k < controls.size(); ++k) {
      if (obs[k].size() != n_obs) throw RoboticsError(RoboticsError::InvalidParameter, "every step of a batch needs the same number of observations");
      u.push_back(controls[k][0]);
      u.push_back(controls[k][1]);
      const std::vector<double> f = flatten(obs[k]);
      flat.insert(flat.end(), f.begin(), f.end());
    }
    std::vector<double> out(4 * controls.size());
    check(rr_pf_step_many(h_, u.data(), flat.data(), n_obs, controls.size(), out.data()));
    std::vector<PFState> r(controls.size());
    for (size_t k = 0; k < controls.size(); ++k) r[k] = {out[4 * k], out[4 * k + 1], out[4 * k + 2], out[4 * k + 3]};
    return r;
  }
  State2D try_step_state(const ControlInput& c, const PFMeasurement& obs) {
    PFState e = try_step({c.v, c.omega}, obs);
    return {e[0], e[1], e[2], e[3]};
  }
  // engine extensions: the step enqueued without waiting for it (rr_pf_step_async), with the mean try_step returns left on the
  // device (rr_pf_step_async_estimate; read it with last_step_estimate), and the resident service of small filters
  // (rr_pf_set_resident: try_step then launches nothing)
  void try_step_async(const PFControl& u, const PFMeasurement& obs, bool with_estimate = false) {
    std::vector<double> flat = flatten(obs);
    check(with_estimate ? rr_pf_step_async_estimate(h_, u.data(), flat.data(), obs.size()) : rr_pf_step_async(h_, u.data(), flat.data(), obs.size()));
  }
  PFState last_step_estimate() {
    PFState e;
    check(rr_pf_last_step_estimate(h_, e.data()));
    return e;
  }
  void synchronize() { check(rr_pf_synchronize(h_)); }
  // engine extension: `ms` milliseconds (0: 50) of step-shaped work, so that the first step runs at the steady rate (rr_pf_warm)
  void warm(double ms = 0.0) { check(rr_pf_warm(h_, ms)); }
  void set_resident(double idle_us) { check(rr_pf_set_resident(h_, idle_us)); }
  PFState estimate() {
    PFState e;
    check(rr_pf_estimate(h_, e.data()));
    return e;
  }
  State2D state_2d() {
    PFState e = estimate();
    return {e[0], e[1], e[2], e[3]};
  }
  std::array<double, 16> calc_covariance() {
    std::array<double, 16> c;
    check(rr_pf_covariance(h_, c.data()));
    return c;
  }
  std::vector<Particle> get_particles() {
    std::vector<Particle> p(rr_pf_particle_count(h_));
    check(rr_pf_get_particles(h_, &p[0].x));
    return p;
  }
  // StateEstimator (core/src/traits.rs:31-52): dt is ignored as in particle_filter.rs:557-559
  void predict(const PFControl& u, double /*dt*/) { try_predict_with_control(u); }
  void update(const PFMeasurement& m) {
    try_update_with_observations(m);
    resample();
  }
  PFState get_state() { return estimate(); }
  rr_pf* handle() { return h_; }

 protected:
  explicit ParticleFilterLocalizer(rr_pf* adopted) : h_(adopted) {}
  static void options(rr_pf_options* o) { rr_pf_options_default(o); }
  static std::vector<double> flatten(const PFMeasurement& obs) {
    std::vector<double> f;
    f.reserve(3 * obs.size());
    for (auto& [d, x, y] : obs) {
      f.push_back(d);
      f.push_back(x);
      f.push_back(y);
    }
    return f;
  }
  rr_pf* h_ = nullptr;
};

// MonteCarloLocalizationConfig, monte_carlo_localization.rs:50-82
struct MonteCarloLocalizationConfig {
  uint64_t min_particles = 100, max_particles = 5000;
  double kld_epsilon = 0.05, kld_z = 2.326;
  double range_noise = 0.2, velocity_noise = 2.0, yaw_rate_noise = 40.0 * 3.14159265358979323846 / 180.0, dt = 0.1;
};

// Monte Carlo localization: resample every step (monte_carlo_localization.rs:291-300); the particle
// count is fixed when min_particles == max_particles and KLD-adaptive otherwise (:322-385)
class MonteCarloLocalizer : public ParticleFilterLocalizer {
 public:
  explicit MonteCarloLocalizer(const MonteCarloLocalizationConfig& c = {}, uint64_t seed = 0, int device = 0)
      : ParticleFilterLocalizer(make(c, nullptr, seed, device)) {}
  MonteCarloLocalizer(const PFState& initial, const MonteCarloLocalizationConfig& c, uint64_t seed = 0, int device = 0)
      : ParticleFilterLocalizer(make(c, initial.data(), seed, device)) {}
  uint64_t particle_count() const { return rr_pf_particle_count(h_); }  // :318-320

 private:
  static rr_pf* make(const MonteCarloLocalizationConfig& c, const double* state, uint64_t seed, int device) {
    rr_pf_config cfg;
    rr_pf_config_default(&cfg);
    cfg.n_particles = c.min_particles;
    cfg.resample_threshold = 1.0;
    cfg.range_noise = c.range_noise;
    cfg.velocity_noise = c.velocity_noise;
    cfg.yaw_rate_noise = c.yaw_rate_noise;
    cfg.dt = c.dt;
    rr_pf_options o;
    rr_pf_options_mcl(&o);
    o.seed = seed;
    o.device = device;
    rr_pf* h = nullptr;
    if (c.min_particles == c.max_particles) {
      check(state ? rr_pf_create_with_state(&cfg, &o, state, &h) : rr_pf_create(&cfg, &o, &h));
    } else {
      rr_mcl_adaptive k{c.min_particles, c.max_particles, c.kld_epsilon, c.kld_z};
      check(rr_pf_create_adaptive(&cfg, &o, &k, state, &h));
    }
    return h;
  }
};

namespace fastslam1 {
struct Params : rr_fs1_params {
  Params() { rr_fs1_params_default(this); }
};

class FastSlam1 {
 public:
  FastSlam1(uint64_t n_particles, uint64_t n_landmarks, const Params& p = {}, uint64_t seed = 0, int device = 0) {
    rr_fs1_options o;
    rr_fs1_options_default(&o);
    o.seed = seed;
    o.device = device;
    check(rr_fs1_create(n_particles, n_landmarks, &p, &o, &h_));
  }
  FastSlam1(const FastSlam1&) = delete;
  ~FastSlam1() { rr_fs1_destroy(h_); }
  // fastslam_update, fastslam1.rs:237-266; z rows = (distance, angle, landmark id)
  void update(const std::array<double, 2>& u, const std::vector<std::tuple<double, double, size_t>>& z) {
    std::vector<double> f;
    for (auto& [d, a, id] : z) {
      f.push_back(d);
      f.push_back(a);
      f.push_back((double)id);
    }
    check(rr_fs1_update(h_, u.data(), f.data(), z.size()));
  }
  // engine extensions: the update enqueued without waiting (rr_fs1_update_async + synchronize), and the resident service of small
  // maps (rr_fs1_set_resident: update launches nothing and is answered with the best particle, so best_particle() after it is free)
  void update_async(const std::array<double, 2>& u, const std::vector<std::tuple<double, double, size_t>>& z) {
    std::vector<double> f;
    for (auto& [d, a, id] : z) {
      f.push_back(d);
      f.push_back(a);
      f.push_back((double)id);
    }
    check(rr_fs1_update_async(h_, u.data(), f.data(), z.size()));
  }
  void synchronize() { check(rr_fs1_synchronize(h_)); }
  void warm(double ms = 0.0) { check(rr_fs1_warm(h_, ms)); }
  void set_resident(double idle_us) { check(rr_fs1_set_resident(h_, idle_us)); }
  // get_best_particle, fastslam1.rs:269-274
  std::tuple<std::array<double, 3>, double, uint64_t> best_particle() {
    std::array<double, 3> pose;
    double w;
    uint64_t i;
    check(rr_fs1_best_particle(h_, pose.data(), &w, &i));
    return {pose, w, i};
  }
  std::vector<double> landmarks_of(uint64_t particle) {
    std::vector<double> out(6 * rr_fs1_landmark_count(h_));
    check(rr_fs1_get_landmarks(h_, particle, out.data()));
    return out;
  }

 protected:
  explicit FastSlam1(rr_fs1* adopted) : h_(adopted) {}
  rr_fs1* h_ = nullptr;
};
}  // namespace fastslam1

namespace fastslam2 {
struct Params : rr_fs2_params {
  Params() { rr_fs2_params_default(this); }
};

// fastslam2.rs: the FastSLAM 1.0 engine with the proposal-sampling step; update() is fastslam2_update :376-383
class FastSlam2 : public fastslam1::FastSlam1 {
 public:
  FastSlam2(uint64_t n_particles, uint64_t n_landmarks, const Params& p = {}, uint64_t seed = 0, int device = 0)
      : fastslam1::FastSlam1(make(n_particles, n_landmarks, p, seed, device)) {}

 private:
  static rr_fs2* make(uint64_t n, uint64_t L, const Params& p, uint64_t seed, int device) {
    rr_fs1_options o;
    rr_fs1_options_default(&o);
    o.seed = seed;
    o.device = device;
    rr_fs2* h = nullptr;
    check(rr_fs2_create(n, L, &p, &o, &h));
    return h;
  }
};
}  // namespace fastslam2
}  // namespace rr
