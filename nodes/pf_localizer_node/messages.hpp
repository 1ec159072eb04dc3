// messages.hpp -- the message shapes the PF localizer node moves, named after the ROS 2 types the reference's node uses
// (ros2_nodes/ekf_localizer_node/src/main.rs:3-11: nav_msgs/Odometry in, geometry_msgs/PoseStamped + nav_msgs/Odometry out)
// plus the one input the EKF node does not have: range observations to known landmarks (the PF's measurement,
// particle_filter.rs:310-334: n x (distance, landmark_x, landmark_y)).  Plain structs: every transport converts to and from them.
#pragma once

#include <array>
#include <cstdint>
#include <string>
#include <vector>

namespace pfnode {

struct Stamp {  // builtin_interfaces/Time; main.rs:35-39 MessageStamp
  int32_t sec = 0;
  uint32_t nanosec = 0;
};

struct Odometry {  // nav_msgs/Odometry
  Stamp stamp;
  std::string frame_id, child_frame_id;
  double position[3] = {0, 0, 0};
  double orientation[4] = {0, 0, 0, 1};  // x, y, z, w
  std::array<double, 36> pose_covariance{};
  double linear[3] = {0, 0, 0};
  double angular[3] = {0, 0, 0};
  std::array<double, 36> twist_covariance{};
};

struct PoseStamped {  // geometry_msgs/PoseStamped
  Stamp stamp;
  std::string frame_id;
  double position[3] = {0, 0, 0};
  double orientation[4] = {0, 0, 0, 1};
};

struct LandmarkRanges {  // n x (distance, landmark_x, landmark_y), PFMeasurement of particle_filter.rs:47
  Stamp stamp;
  std::vector<double> ranges;
};

struct Input {
  enum Kind { kOdometry, kLandmarkRanges } kind = kOdometry;
  Odometry odom;
  LandmarkRanges ranges;
};

struct Output {
  enum Kind { kPose, kOdometry } kind = kPose;
  std::string topic;
  PoseStamped pose;
  Odometry odom;
  double state[4] = {0, 0, 0, 0};  // (x, y, yaw, v) as the localizer returned it: carried by transports that can (JSON lines)
};

enum class LogLevel { kInfo, kWarn };

// The seam between the node's logic and whatever moves its messages: the three things the reference's node asks of
// safe_drive -- selector.wait() with the subscriber callback (main.rs:204,299-301), publisher.send() (:130,160) and
// pr_info!/pr_warn! (:193-199,210).
class Transport {
 public:
  virtual ~Transport() = default;
  // blocks until one message of a subscribed topic is there; false: the transport has shut down
  virtual bool wait(Input* out) = 0;
  virtual void publish(const Output& msg) = 0;
  virtual void log(LogLevel level, const std::string& text) = 0;
};

struct Topics {  // main.rs:19-21,174-176 (defaults and the environment variables that override them)
  std::string input_odom, input_ranges, output_odom, output_pose;
};

}  // namespace pfnode
