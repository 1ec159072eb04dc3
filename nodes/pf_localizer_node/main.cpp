// pf_localizer_node -- a particle-filter localizer node on the MI355X engine, structured as the reference's
// ros2_nodes/ekf_localizer_node/src/main.rs:173-301 (see node_logic.hpp) and its dora twin
// crates/rust_robotics/examples/dora_ekf_node.rs.  The transport is chosen at build time:
//   default                 JSON lines over stdio / a UNIX socket (transport_jsonl.hpp) -- builds with g++ alone
//   -DPFNODE_WITH_RCLCPP    ROS 2 (transport_rclcpp.cpp; CMake: find_package(rclcpp))
//   -DPFNODE_WITH_DORA      dora-rs C++ node API (transport_dora.cpp)
// Environment: PF_INPUT_ODOM_TOPIC, PF_INPUT_RANGES_TOPIC, PF_OUTPUT_ODOM_TOPIC, PF_OUTPUT_POSE_TOPIC (main.rs:174-176),
// PF_PARTICLES, PF_RESAMPLE_THRESHOLD, PF_RANGE_NOISE, PF_VELOCITY_NOISE, PF_YAW_RATE_NOISE, PF_DT, PF_SEED, PF_DEVICE,
// PF_RESIDENT_IDLE_US, PF_LOG_INTERVAL_S, PF_TRANSPORT, PF_PIN_TO_GPU_NUMA; PF_LOCALIZER=mcl (+ PF_MIN_PARTICLES, PF_MAX_PARTICLES): the
// MonteCarloLocalizer with its KLD-adaptive particle count instead of the ParticleFilterLocalizer.
//   pf_localizer_node --self-test   the reference node's own unit tests (main.rs:303-384) on this node's helpers; no GPU
#include <csignal>
#include <sched.h>

#include <cstdio>
#include <fstream>
#include <iostream>
#include <memory>

#include "node_logic.hpp"
#include "transport_jsonl.hpp"

namespace pfnode {
std::unique_ptr<Transport> make_rclcpp_transport(int argc, char** argv, const Topics& topics);  // transport_rclcpp.cpp
std::unique_ptr<Transport> make_dora_transport(const Topics& topics);                            // transport_dora.cpp

// The synchronous step is two trips over the host link (csrc/resident_core.hpp): keep the calling thread on the NUMA node the
// GPU hangs off, or every poll crosses the socket interconnect as well (measured: +2 us per step on a two-socket host).
static void pin_to_gpu_numa_node(int device, Transport* io) {
  char bus[64] = {0};
  if (rr_device_pci_bus_id(device, bus, sizeof bus) != RR_OK) return;
  std::ifstream f(std::string("/sys/bus/pci/devices/") + bus + "/local_cpulist");
  std::string list;
  if (!f || !std::getline(f, list) || list.empty()) return;
  cpu_set_t set;
  CPU_ZERO(&set);
  int n_cpus = 0;
  for (size_t pos = 0; pos < list.size();) {  // "64-127,192-255"
    size_t end = list.find(',', pos);
    if (end == std::string::npos) end = list.size();
    const std::string part = list.substr(pos, end - pos);
    const size_t dash = part.find('-');
    const int a = std::atoi(part.c_str()), b = dash == std::string::npos ? a : std::atoi(part.c_str() + dash + 1);
    for (int c = a; c <= b && c < CPU_SETSIZE; ++c) CPU_SET(c, &set), ++n_cpus;
    pos = end + 1;
  }
  if (n_cpus && sched_setaffinity(0, sizeof set, &set) == 0) io->log(LogLevel::kInfo, "pinned to the GPU's NUMA node (cpus " + list + ")");
}
}  // namespace pfnode

static int self_test();

int main(int argc, char** argv) {
  using namespace pfnode;
  if (argc > 1 && std::string(argv[1]) == "--self-test") return self_test();
  // a peer that hangs up must end this node through the transport's "peer is gone" path (a closing log line, the resident kernel
  // parked), not through SIGPIPE's default action
  std::signal(SIGPIPE, SIG_IGN);
  try {
    Settings settings = Settings::from_env();
    std::unique_ptr<Transport> io;
#if defined(PFNODE_WITH_RCLCPP)
    io = make_rclcpp_transport(argc, argv, settings.topics);
#elif defined(PFNODE_WITH_DORA)
    io = make_dora_transport(settings.topics);
#else
    io = std::make_unique<JsonLinesTransport>(topic_from_env("PF_TRANSPORT", "stdio"), settings.topics);
#endif
    if (number_from_env("PF_PIN_TO_GPU_NUMA", 1.0) != 0.0) pin_to_gpu_numa_node(settings.device, io.get());
    Node node(std::move(settings), io.get());
    node.spin();
    return 0;
  } catch (const std::exception& e) {
    std::fprintf(stderr, "pf_localizer_node: %s\n", e.what());
    return 1;
  }
}

// ---- the reference node's unit tests (ros2_nodes/ekf_localizer_node/src/main.rs:303-384), on this node's helpers,
// plus the wire codec.  No GPU, no localizer.
#define PF_EXPECT(cond)                                                      \
  do {                                                                       \
    if (!(cond)) {                                                           \
      std::fprintf(stderr, "self-test failed at line %d: %s\n", __LINE__, #cond); \
      return 1;                                                              \
    }                                                                        \
  } while (0)

static int self_test() {
  using namespace pfnode;
  const auto close = [](double a, double b) { return std::fabs(a - b) < 1e-9; };
  // sanitize_dt_uses_default_without_history, main.rs:311-318
  PF_EXPECT(close(sanitize_dt(std::nullopt, Stamp{10, 0}), kFallbackDt));
  // sanitize_dt_clamps_bounds, :320-333
  PF_EXPECT(close(sanitize_dt(Stamp{9, 999900000u}, Stamp{10, 0}), kMinDt));
  PF_EXPECT(close(sanitize_dt(Stamp{8, 0}, Stamp{10, 0}), kMaxDt));
  // sanitize_dt_uses_min_dt_for_non_monotonic_stamp, :335-346
  PF_EXPECT(close(sanitize_dt(Stamp{10, 500000000u}, Stamp{10, 400000000u}), kMinDt));
  PF_EXPECT(close(sanitize_dt(Stamp{10, 0}, Stamp{10, 100000000u}), 0.1));
  // yaw_from_quaternion_handles_planar_rotation, :348-353
  const double half = 0.78539816339744830962;
  PF_EXPECT(close(yaw_from_quaternion(0.0, 0.0, std::sin(half), std::cos(half)), 2.0 * half));
  // output_frame_id_falls_back_when_source_is_empty, :355-359
  Odometry m;
  PF_EXPECT(output_frame_id(m) == kDefaultFrameId);
  m.frame_id = "map";
  PF_EXPECT(output_frame_id(m) == "map");
  // copy_stamp_preserves_source_timestamp / stamp_to_nanos_combines_sec_and_nanosec, :361-383
  PF_EXPECT(stamp_to_nanos(Stamp{3, 25}) == (__int128)3000000025LL);
  PF_EXPECT(stamp_to_nanos(Stamp{-1, 0}) == (__int128)-1000000000LL);
  // apply_yaw_to_pose is the inverse of yaw_from_quaternion on planar rotations
  double q[4];
  apply_yaw_to_pose(q, 0.7);
  PF_EXPECT(close(yaw_from_quaternion(q[0], q[1], q[2], q[3]), 0.7));
  // initial_state_from_odom, :78-86
  m.position[0] = 5.0, m.position[1] = 6.0, m.linear[0] = 0.25;
  apply_yaw_to_pose(m.orientation, -0.3);
  const rr::State2D s0 = initial_state_from_odom(m);
  PF_EXPECT(close(s0.x, 5.0) && close(s0.y, 6.0) && close(s0.yaw, -0.3) && close(s0.v, 0.25));
  // topic_from_env: unset and blank values fall back, :41-46
  unsetenv("PF_TEST_TOPIC");
  PF_EXPECT(topic_from_env("PF_TEST_TOPIC", "/odom") == "/odom");
  setenv("PF_TEST_TOPIC", "   ", 1);
  PF_EXPECT(topic_from_env("PF_TEST_TOPIC", "/odom") == "/odom");
  setenv("PF_TEST_TOPIC", "/robot/odom", 1);
  PF_EXPECT(topic_from_env("PF_TEST_TOPIC", "/odom") == "/robot/odom");
  // the wire codec: doubles survive a round trip bit for bit, unknown topics are refused, malformed lines throw
  const Topics topics{"/odom", "/landmark_ranges", "/pf_odom", "/pf_pose"};
  Input in;
  PF_EXPECT(decode_input("{\"topic\": \"/odom\", \"stamp\": [12, 34], \"frame_id\": \"odom\", \"child_frame_id\": \"base\", "
                         "\"pose\": [0.1, 0.2, 0, 0, 0, 0.5, 0.8660254037844386], \"twist\": [1.1, 0, 0, 0, 0, 0.63]}",
                         topics, &in));
  PF_EXPECT(in.kind == Input::kOdometry && in.odom.stamp.sec == 12 && in.odom.stamp.nanosec == 34u && in.odom.child_frame_id == "base");
  PF_EXPECT(in.odom.position[0] == 0.1 && in.odom.orientation[3] == 0.8660254037844386 && in.odom.linear[0] == 1.1 && in.odom.angular[2] == 0.63);
  PF_EXPECT(decode_input("{\"topic\": \"/landmark_ranges\", \"stamp\": [1, 2], \"ranges\": [3.5, 2.0, 2.0, 4.25, 10.0, 2.0]}", topics, &in));
  PF_EXPECT(in.kind == Input::kLandmarkRanges && in.ranges.ranges.size() == 6 && in.ranges.ranges[3] == 4.25);
  PF_EXPECT(!decode_input("{\"topic\": \"/scan\"}", topics, &in));
  bool threw = false;
  try {
    decode_input("{\"topic\": \"/odom\", \"pose\": [1, 2]}", topics, &in);
  } catch (const std::exception&) {
    threw = true;
  }
  PF_EXPECT(threw);
  Output out;
  out.kind = Output::kPose;
  out.topic = "/pf_pose";
  out.pose.frame_id = "odom";
  out.state[0] = 0.1 + 0.2;  // 0.30000000000000004
  const std::string line = encode_output(out);
  const JsonObject o = JsonReader(line).object();
  PF_EXPECT(o.at("state").a[0] == 0.1 + 0.2 && o.at("topic").s == "/pf_pose" && o.at("pose").a.size() == 7);
  // configuration: ParticleFilterConfig::default() (particle_filter.rs:67-78) unless the environment overrides it
  unsetenv("PF_PARTICLES");
  PF_EXPECT(Settings::from_env().filter.n_particles == 100);
  setenv("PF_PARTICLES", "150", 1);
  setenv("PF_RANGE_NOISE", "0.25", 1);
  const Settings st = Settings::from_env();
  PF_EXPECT(st.filter.n_particles == 150 && st.filter.range_noise == 0.25 && st.filter.dt == 0.1 && st.topics.output_pose == kDefaultOutputPoseTopic);
  std::puts("pf_localizer_node self-test ok");
  return 0;
}
