// transport_rclcpp.cpp -- ROS 2 adapter of the node's three-function transport (messages.hpp): nav_msgs/Odometry in,
// std_msgs/Float64MultiArray in (n x (distance, landmark_x, landmark_y) -- ROS 2 has no stock range-to-landmark message),
// geometry_msgs/PoseStamped + nav_msgs/Odometry out: the topics and types of ros2_nodes/ekf_localizer_node/src/main.rs:188-191
// plus the ranges input.  Built only where rclcpp is (CMakeLists.txt: find_package(rclcpp)); SOURCE ONLY in this repository's
// image (no ROS 2 there), so it has not met a compiler -- the logic it feeds is the one tests/test_gpu_node.py exercises.
//
// The reference's node registers its callback with a selector and loops on selector.wait() (main.rs:204,299-301); here the
// subscriptions queue their messages and wait() spins the executor until one is there, so the node's loop stays the single
// place where the localizer is touched.
#include <deque>
#include <memory>

#include <geometry_msgs/msg/pose_stamped.hpp>
#include <nav_msgs/msg/odometry.hpp>
#include <rclcpp/rclcpp.hpp>
#include <std_msgs/msg/float64_multi_array.hpp>

#include "messages.hpp"

namespace pfnode {

class RclcppTransport : public Transport {
 public:
  RclcppTransport(int argc, char** argv, const Topics& topics) {
    rclcpp::init(argc, argv);
    node_ = std::make_shared<rclcpp::Node>("pf_localizer_node");
    odom_sub_ = node_->create_subscription<nav_msgs::msg::Odometry>(topics.input_odom, 10, [this](nav_msgs::msg::Odometry::ConstSharedPtr m) {
      Input in;
      in.kind = Input::kOdometry;
      Odometry& o = in.odom;
      o.stamp = {m->header.stamp.sec, m->header.stamp.nanosec};
      o.frame_id = m->header.frame_id;
      o.child_frame_id = m->child_frame_id;
      o.position[0] = m->pose.pose.position.x, o.position[1] = m->pose.pose.position.y, o.position[2] = m->pose.pose.position.z;
      o.orientation[0] = m->pose.pose.orientation.x, o.orientation[1] = m->pose.pose.orientation.y;
      o.orientation[2] = m->pose.pose.orientation.z, o.orientation[3] = m->pose.pose.orientation.w;
      std::copy(m->pose.covariance.begin(), m->pose.covariance.end(), o.pose_covariance.begin());
      o.linear[0] = m->twist.twist.linear.x, o.linear[1] = m->twist.twist.linear.y, o.linear[2] = m->twist.twist.linear.z;
      o.angular[0] = m->twist.twist.angular.x, o.angular[1] = m->twist.twist.angular.y, o.angular[2] = m->twist.twist.angular.z;
      std::copy(m->twist.covariance.begin(), m->twist.covariance.end(), o.twist_covariance.begin());
      queue_.push_back(std::move(in));
    });
    ranges_sub_ = node_->create_subscription<std_msgs::msg::Float64MultiArray>(topics.input_ranges, 10, [this](std_msgs::msg::Float64MultiArray::ConstSharedPtr m) {
      Input in;
      in.kind = Input::kLandmarkRanges;
      const auto now = node_->now();
      in.ranges.stamp = {(int32_t)(now.nanoseconds() / 1000000000), (uint32_t)(now.nanoseconds() % 1000000000)};
      in.ranges.ranges.assign(m->data.begin(), m->data.end());
      queue_.push_back(std::move(in));
    });
    pose_pub_ = node_->create_publisher<geometry_msgs::msg::PoseStamped>(topics.output_pose, 10);
    odom_pub_ = node_->create_publisher<nav_msgs::msg::Odometry>(topics.output_odom, 10);
    exec_.add_node(node_);
  }
  ~RclcppTransport() override { rclcpp::shutdown(); }

  bool wait(Input* out) override {
    while (queue_.empty()) {
      if (!rclcpp::ok()) return false;
      exec_.spin_once(std::chrono::milliseconds(100));
    }
    *out = std::move(queue_.front());
    queue_.pop_front();
    return true;
  }
  void publish(const Output& msg) override {
    if (msg.kind == Output::kPose) {  // main.rs:118-131
      geometry_msgs::msg::PoseStamped m;
      m.header.stamp.sec = msg.pose.stamp.sec, m.header.stamp.nanosec = msg.pose.stamp.nanosec;
      m.header.frame_id = msg.pose.frame_id;
      m.pose.position.x = msg.pose.position[0], m.pose.position.y = msg.pose.position[1], m.pose.position.z = msg.pose.position[2];
      m.pose.orientation.x = msg.pose.orientation[0], m.pose.orientation.y = msg.pose.orientation[1];
      m.pose.orientation.z = msg.pose.orientation[2], m.pose.orientation.w = msg.pose.orientation[3];
      pose_pub_->publish(m);
      return;
    }
    nav_msgs::msg::Odometry m;  // main.rs:133-162
    const Odometry& o = msg.odom;
    m.header.stamp.sec = o.stamp.sec, m.header.stamp.nanosec = o.stamp.nanosec;
    m.header.frame_id = o.frame_id;
    m.child_frame_id = o.child_frame_id;
    m.pose.pose.position.x = o.position[0], m.pose.pose.position.y = o.position[1], m.pose.pose.position.z = o.position[2];
    m.pose.pose.orientation.x = o.orientation[0], m.pose.pose.orientation.y = o.orientation[1];
    m.pose.pose.orientation.z = o.orientation[2], m.pose.pose.orientation.w = o.orientation[3];
    std::copy(o.pose_covariance.begin(), o.pose_covariance.end(), m.pose.covariance.begin());
    m.twist.twist.linear.x = o.linear[0], m.twist.twist.linear.y = o.linear[1], m.twist.twist.linear.z = o.linear[2];
    m.twist.twist.angular.x = o.angular[0], m.twist.twist.angular.y = o.angular[1], m.twist.twist.angular.z = o.angular[2];
    std::copy(o.twist_covariance.begin(), o.twist_covariance.end(), m.twist.covariance.begin());
    odom_pub_->publish(m);
  }
  void log(LogLevel level, const std::string& text) override {
    if (level == LogLevel::kWarn) RCLCPP_WARN(node_->get_logger(), "%s", text.c_str());
    else RCLCPP_INFO(node_->get_logger(), "%s", text.c_str());
  }

 private:
  rclcpp::Node::SharedPtr node_;
  rclcpp::executors::SingleThreadedExecutor exec_;
  rclcpp::Subscription<nav_msgs::msg::Odometry>::SharedPtr odom_sub_;
  rclcpp::Subscription<std_msgs::msg::Float64MultiArray>::SharedPtr ranges_sub_;
  rclcpp::Publisher<geometry_msgs::msg::PoseStamped>::SharedPtr pose_pub_;
  rclcpp::Publisher<nav_msgs::msg::Odometry>::SharedPtr odom_pub_;
  std::deque<Input> queue_;
};

std::unique_ptr<Transport> make_rclcpp_transport(int argc, char** argv, const Topics& topics) {
  return std::make_unique<RclcppTransport>(argc, argv, topics);
}

}  // namespace pfnode
