// node_logic.hpp -- the PF localizer node's logic, transport-free: what ros2_nodes/ekf_localizer_node/src/main.rs does around
// its EKF, done around rr::ParticleFilterLocalizer (include/rust_robotics.hpp -> the C ABI -> the HIP engine).
//
//   * configuration from environment variables with defaults (main.rs:19-26,41-46,174-176);
//   * ONE localizer behind a mutex, driven by one selector loop (main.rs:181-186,204-218,299-301);
//   * the first odometry message initialises the filter at its pose and is published as is (main.rs:220-256);
//   * every later odometry message: dt bookkeeping (sanitize_dt, main.rs:63-76), control = (twist.linear.x, twist.angular.z)
//     (main.rs:262), one try_step_state with the range observations that arrived since the last step (none: a pure
//     prediction with uniform weights, particle_filter.rs:317-331), pose + odometry out with the source's stamp and frames
//     (main.rs:116-162), a line in the log every 5 s (main.rs:163-171,283-292) -- here with the step latency;
//   * errors are logged and the message dropped, never fatal (main.rs:266-272).
// The particle filter's dt is a configuration constant in the reference (ParticleFilterConfig.dt; try_step takes none,
// StateEstimator::predict ignores its argument, particle_filter.rs:557-559): the node logs when the observed message
// interval strays from it instead of feeding it to the filter.
#pragma once

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <optional>
#include <string>

#include "messages.hpp"
#include "rust_robotics.hpp"

namespace pfnode {

constexpr const char* kDefaultInputOdomTopic = "/odom";             // main.rs:19
constexpr const char* kDefaultInputRangesTopic = "/landmark_ranges";
constexpr const char* kDefaultOutputOdomTopic = "/pf_odom";         // main.rs:20 ("/ekf_odom")
constexpr const char* kDefaultOutputPoseTopic = "/pf_pose";         // main.rs:21 ("/ekf_pose")
constexpr const char* kDefaultFrameId = "odom";                     // main.rs:22
constexpr double kLogIntervalSeconds = 5.0;                         // main.rs:23
constexpr double kFallbackDt = 0.1, kMinDt = 1e-3, kMaxDt = 0.5;    // main.rs:24-26

// main.rs:41-46
inline std::string topic_from_env(const char* key, const char* fallback) {
  const char* v = std::getenv(key);
  if (!v) return fallback;
  std::string s(v);
  const size_t a = s.find_first_not_of(" \t\r\n");
  if (a == std::string::npos) return fallback;
  return s;
}
// a whole number from the environment: NaN, negative or absurdly large values (a cast of which is undefined) fall back
inline uint64_t count_from_env(const char* key, uint64_t fallback, uint64_t max = (uint64_t)1 << 40) {
  const char* v = std::getenv(key);
  if (!v || !*v) return fallback;
  char* end = nullptr;
  const double x = std::strtod(v, &end);
  if (end == v || !(x >= 0.0) || !(x <= (double)max)) return fallback;
  return (uint64_t)x;
}
inline double number_from_env(const char* key, double fallback) {
  const char* v = std::getenv(key);
  if (!v || !*v) return fallback;
  char* end = nullptr;
  const double x = std::strtod(v, &end);
  return end == v ? fallback : x;
}

// main.rs:48-52
inline double yaw_from_quaternion(double x, double y, double z, double w) {
  const double siny_cosp = 2.0 * (w * z + x * y);
  const double cosy_cosp = 1.0 - 2.0 * (y * y + z * z);
  return std::atan2(siny_cosp, cosy_cosp);
}
// main.rs:54-60
inline void apply_yaw_to_pose(double orientation[4], double yaw) {
  const double half = 0.5 * yaw;
  orientation[0] = 0.0;
  orientation[1] = 0.0;
  orientation[2] = std::sin(half);
  orientation[3] = std::cos(half);
}
// main.rs:105-107 (i128 there; 96 bits are plenty)
inline __int128 stamp_to_nanos(Stamp s) { return (__int128)s.sec * 1000000000 + (__int128)s.nanosec; }
// main.rs:63-76
inline double sanitize_dt(const std::optional<Stamp>& last, Stamp now) {
  if (!last) return kFallbackDt;
  const __int128 delta = stamp_to_nanos(now) - stamp_to_nanos(*last);
  if (delta <= 0) return kMinDt;
  const double dt = (double)delta / 1e9;
  return dt < kMinDt ? kMinDt : (dt > kMaxDt ? kMaxDt : dt);
}
// main.rs:109-116
inline std::string output_frame_id(const Odometry& source) { return source.frame_id.empty() ? kDefaultFrameId : source.frame_id; }
// main.rs:78-86
inline rr::State2D initial_state_from_odom(const Odometry& m) {
  return {m.position[0], m.position[1], yaw_from_quaternion(m.orientation[0], m.orientation[1], m.orientation[2], m.orientation[3]),
          m.linear[0]};
}

struct Settings {
  Topics topics;
  rr::ParticleFilterConfig filter;
  // PF_LOCALIZER=mcl: rust_robotics_localization's MonteCarloLocalizer instead (resample at every step; the particle count follows the
  // KLD bound between PF_MIN_PARTICLES and PF_MAX_PARTICLES, monte_carlo_localization.rs:50-82) -- same node, same messages
  bool use_mcl = false;
  rr::MonteCarloLocalizationConfig mcl;
  uint64_t seed = 0;
  int device = 0;
  // rr_pf_set_resident: the step kernel stays on the device between messages and leaves after this long without one (0: a launch
  // per step).  250 ms keeps it there for any odometry rate above 4 Hz; the price is one workgroup's slots of one compute unit.
  double resident_idle_us = 250000.0;
  double log_interval_s = kLogIntervalSeconds;

  static Settings from_env() {
    Settings s;
    s.topics.input_odom = topic_from_env("PF_INPUT_ODOM_TOPIC", kDefaultInputOdomTopic);
    s.topics.input_ranges = topic_from_env("PF_INPUT_RANGES_TOPIC", kDefaultInputRangesTopic);
    s.topics.output_odom = topic_from_env("PF_OUTPUT_ODOM_TOPIC", kDefaultOutputOdomTopic);
    s.topics.output_pose = topic_from_env("PF_OUTPUT_POSE_TOPIC", kDefaultOutputPoseTopic);
    // ParticleFilterConfig::default() (particle_filter.rs:67-78) unless the environment says otherwise
    s.filter.n_particles = count_from_env("PF_PARTICLES", s.filter.n_particles);
    s.filter.resample_threshold = number_from_env("PF_RESAMPLE_THRESHOLD", s.filter.resample_threshold);
    s.filter.range_noise = number_from_env("PF_RANGE_NOISE", s.filter.range_noise);
    s.filter.velocity_noise = number_from_env("PF_VELOCITY_NOISE", s.filter.velocity_noise);
    s.filter.yaw_rate_noise = number_from_env("PF_YAW_RATE_NOISE", s.filter.yaw_rate_noise);
    s.filter.dt = number_from_env("PF_DT", s.filter.dt);
    s.use_mcl = topic_from_env("PF_LOCALIZER", "pf") == "mcl";
    s.mcl.min_particles = count_from_env("PF_MIN_PARTICLES", s.mcl.min_particles);
    s.mcl.max_particles = count_from_env("PF_MAX_PARTICLES", s.mcl.max_particles);
    s.mcl.range_noise = s.filter.range_noise;
    s.mcl.velocity_noise = s.filter.velocity_noise;
    s.mcl.yaw_rate_noise = s.filter.yaw_rate_noise;
    s.mcl.dt = s.filter.dt;
    s.seed = count_from_env("PF_SEED", 0, (uint64_t)1 << 53);
    s.device = (int)count_from_env("PF_DEVICE", 0, 63);
    s.resident_idle_us = number_from_env("PF_RESIDENT_IDLE_US", s.resident_idle_us);
    s.log_interval_s = number_from_env("PF_LOG_INTERVAL_S", kLogIntervalSeconds);
    return s;
  }
};

// main.rs:28-33 EkfState
struct PfState {
  std::unique_ptr<rr::ParticleFilterLocalizer> localizer;
  bool initialized = false;
  std::optional<Stamp> last_update_at;
  std::optional<std::chrono::steady_clock::time_point> last_log_at;
  rr::PFMeasurement pending_ranges;  // observations since the last step
  // statistics of the log line
  uint64_t steps = 0, steps_at_last_log = 0;
  double latency_sum_us = 0.0, latency_max_us = 0.0, dt_observed_sum = 0.0;
};

class Node {
 public:
  Node(Settings settings, Transport* transport) : cfg_(std::move(settings)), io_(transport) {
    cfg_.filter.validate();  // InvalidParameter here is fatal, as `?` on the initial localizer is in main.rs:178-180
  }

  const Settings& settings() const { return cfg_; }

  // main.rs:299-301: loop { selector.wait()? } with the subscriber callback inside
  void spin() {
    io_->log(LogLevel::kInfo, "pf localizer started (odom: " + cfg_.topics.input_odom + ", ranges: " + cfg_.topics.input_ranges +
                                  ", pose: " + cfg_.topics.output_pose + ", odom out: " + cfg_.topics.output_odom + ", particles: " +
                                  std::to_string(cfg_.filter.n_particles) + ")");
    Input in;
    while (io_->wait(&in)) {
      if (in.kind == Input::kLandmarkRanges) on_ranges(in.ranges);
      else on_odom(in.odom);
    }
    std::lock_guard<std::mutex> lock(mu_);
    io_->log(LogLevel::kInfo, stats_line("shutting down"));
  }

  // range observations: stored for the next step (the odometry callback consumes them)
  void on_ranges(const LandmarkRanges& m) {
    std::lock_guard<std::mutex> lock(mu_);
    if (m.ranges.size() % 3 != 0) {
      io_->log(LogLevel::kWarn, "landmark ranges must be n x (distance, landmark_x, landmark_y); dropped");
      return;
    }
    st_.pending_ranges.clear();
    for (size_t k = 0; k + 2 < m.ranges.size(); k += 3) st_.pending_ranges.emplace_back(m.ranges[k], m.ranges[k + 1], m.ranges[k + 2]);
  }

  // main.rs:205-297
  void on_odom(const Odometry& msg) {
    const auto now = std::chrono::steady_clock::now();
    std::lock_guard<std::mutex> lock(mu_);
    if (!st_.initialized) {
      const rr::State2D init = initial_state_from_odom(msg);
      try {
        if (cfg_.use_mcl)
          st_.localizer = std::make_unique<rr::MonteCarloLocalizer>(rr::PFState{init.x, init.y, init.yaw, init.v}, cfg_.mcl, cfg_.seed, cfg_.device);
        else
          st_.localizer = std::make_unique<rr::ParticleFilterLocalizer>(
              rr::ParticleFilterLocalizer::with_initial_state_2d(init, cfg_.filter, cfg_.seed, cfg_.device));
        if (cfg_.resident_idle_us > 0.0) st_.localizer->set_resident(cfg_.resident_idle_us);
      } catch (const rr::RoboticsError& e) {
        io_->log(LogLevel::kWarn, std::string("failed to initialize PF state: ") + e.what());
        return;
      }
      st_.initialized = true;
      st_.last_update_at = msg.stamp;
      publish(init, msg);
      if (should_log(now)) io_->log(LogLevel::kInfo, "initialized filtered pose " + pose_text(init));
      return;
    }
    const double dt = sanitize_dt(st_.last_update_at, msg.stamp);
    st_.last_update_at = msg.stamp;
    const rr::ControlInput control{msg.linear[0], msg.angular[2]};
    rr::State2D estimate;
    try {
      estimate = st_.localizer->try_step_state(control, st_.pending_ranges);
    } catch (const rr::RoboticsError& e) {
      io_->log(LogLevel::kWarn, std::string("PF update failed: ") + e.what());
      st_.pending_ranges.clear();
      return;
    }
    st_.pending_ranges.clear();
    publish(estimate, msg);
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - now).count();
    st_.steps += 1;
    st_.latency_sum_us += us;
    st_.latency_max_us = us > st_.latency_max_us ? us : st_.latency_max_us;
    st_.dt_observed_sum += dt;
    if (should_log(now)) io_->log(LogLevel::kInfo, stats_line("filtered pose " + pose_text(estimate)));
  }

 private:
  // main.rs:116-162: both outputs carry the source's stamp and frames; the odometry keeps the source's twist and covariances
  void publish(const rr::State2D& s, const Odometry& source) {
    Output pose;
    pose.kind = Output::kPose;
    pose.topic = cfg_.topics.output_pose;
    pose.pose.stamp = source.stamp;
    pose.pose.frame_id = output_frame_id(source);
    pose.pose.position[0] = s.x;
    pose.pose.position[1] = s.y;
    pose.pose.position[2] = 0.0;
    apply_yaw_to_pose(pose.pose.orientation, s.yaw);
    const double st[4] = {s.x, s.y, s.yaw, s.v};
    std::memcpy(pose.state, st, sizeof st);
    io_->publish(pose);
    Output odom;
    odom.kind = Output::kOdometry;
    odom.topic = cfg_.topics.output_odom;
    odom.odom = source;  // twist, covariances, child frame (main.rs:142-158)
    odom.odom.frame_id = output_frame_id(source);
    odom.odom.position[0] = s.x;
    odom.odom.position[1] = s.y;
    odom.odom.position[2] = 0.0;
    apply_yaw_to_pose(odom.odom.orientation, s.yaw);
    std::memcpy(odom.state, st, sizeof st);
    io_->publish(odom);
  }
  // main.rs:163-171
  bool should_log(std::chrono::steady_clock::time_point now) {
    const bool go = !st_.last_log_at || std::chrono::duration<double>(now - *st_.last_log_at).count() >= cfg_.log_interval_s;
    if (go) st_.last_log_at = now;
    return go;
  }
  static std::string pose_text(const rr::State2D& s) {
    char b[160];
    std::snprintf(b, sizeof b, "x=%.2f y=%.2f yaw=%.2f v=%.2f", s.x, s.y, s.yaw, s.v);
    return b;
  }
  std::string stats_line(const std::string& head) {
    char b[256];
    const double n = (double)(st_.steps ? st_.steps : 1);
    std::snprintf(b, sizeof b, " | steps=%llu (+%llu) step latency mean=%.1f us max=%.1f us, message interval mean=%.4f s (filter dt %.4f s)",
                  (unsigned long long)st_.steps, (unsigned long long)(st_.steps - st_.steps_at_last_log), st_.latency_sum_us / n,
                  st_.latency_max_us, st_.dt_observed_sum / n, cfg_.filter.dt);
    st_.steps_at_last_log = st_.steps;
    return head + b;
  }

  Settings cfg_;
  Transport* io_;
  std::mutex mu_;  // Arc<Mutex<EkfState>> of main.rs:181-186: the callback is the only writer, other threads may read
  PfState st_;
};

}  // namespace pfnode
