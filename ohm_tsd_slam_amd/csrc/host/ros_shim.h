// ros_shim.h -- the slice of the ROS 2 API that ThreadLocalize / ThreadMapping touch
// (SURVEY Appendix D).  With rclcpp available the real headers are used; on boxes without ROS
// (this image, the GPU box) a plain-C++ stand-in with the same spelling is compiled instead:
// parameters are a string-keyed map, publishers keep the last message and a counter, the tf
// broadcaster keeps the last transform.  Topic names, message fields, parameter names and defaults are
// the reference's.
#pragma once

#if __has_include(<rclcpp/rclcpp.hpp>) && !defined(OHM_TSD_SLAM_NO_ROS)
#define OHM_TSD_SLAM_HAVE_ROS 1
#include <rclcpp/rclcpp.hpp>
#include <sensor_msgs/msg/laser_scan.hpp>
#include <geometry_msgs/msg/pose_stamped.hpp>
#include <geometry_msgs/msg/transform_stamped.hpp>
#include <tf2_ros/transform_broadcaster.h>
#else
#define OHM_TSD_SLAM_HAVE_ROS 0
#include <chrono>
#include <cstdint>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <variant>
#include <vector>

namespace builtin_interfaces { namespace msg {
struct Time { int32_t sec = 0; uint32_t nanosec = 0; };
} }
namespace std_msgs { namespace msg {
struct Header { builtin_interfaces::msg::Time stamp; std::string frame_id; };
} }
namespace sensor_msgs { namespace msg {
// fields consumed by the reference: ranges, angle_min, angle_increment, header.stamp
// (ThreadLocalize.cpp:252-256,321-326,487-499)
struct LaserScan {
  std_msgs::msg::Header header;
  float angle_min = 0.f, angle_max = 0.f, angle_increment = 0.f;
  float time_increment = 0.f, scan_time = 0.f, range_min = 0.f, range_max = 0.f;
  std::vector<float> ranges, intensities;
};
} }
namespace geometry_msgs { namespace msg {
struct Point { double x = 0, y = 0, z = 0; };
struct Quaternion { double x = 0, y = 0, z = 0, w = 1; };
struct Vector3 { double x = 0, y = 0, z = 0; };
struct Pose { Point position; Quaternion orientation; };
struct PoseStamped { std_msgs::msg::Header header; Pose pose; };
struct Transform { Vector3 translation; Quaternion rotation; };
struct TransformStamped { std_msgs::msg::Header header; std::string child_frame_id; Transform transform; };
} }

namespace rclcpp {

struct Time { int64_t ns = 0; operator builtin_interfaces::msg::Time() const { return {(int32_t)(ns / 1000000000LL), (uint32_t)(ns % 1000000000LL)}; } };
struct Clock {
  Time now() const { return Time{std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count()}; }
};

class Parameter {
public:
  using Value = std::variant<bool, int64_t, double, std::string>;
  Parameter() = default;
  explicit Parameter(Value v) : _v(std::move(v)) {}
  bool as_bool() const { return std::get<bool>(_v); }
  int64_t as_int() const { return std::holds_alternative<int64_t>(_v) ? std::get<int64_t>(_v) : (int64_t)std::get<double>(_v); }
  double as_double() const { return std::holds_alternative<double>(_v) ? std::get<double>(_v) : (double)std::get<int64_t>(_v); }
  std::string as_string() const { return std::get<std::string>(_v); }
  const Value& value() const { return _v; }
private:
  Value _v;
};

template <class MsgT>
class Publisher {
public:
  explicit Publisher(std::string topic) : _topic(std::move(topic)) {}
  void publish(const MsgT& m) { std::lock_guard<std::mutex> lk(_mx); _last = m; _count++; }
  MsgT last() const { std::lock_guard<std::mutex> lk(_mx); return _last; }
  uint64_t count() const { std::lock_guard<std::mutex> lk(_mx); return _count; }
  const std::string& topic() const { return _topic; }
private:
  std::string _topic;
  mutable std::mutex _mx;
  MsgT _last{};
  uint64_t _count = 0;
};

class Node : public std::enable_shared_from_this<Node> {
public:
  explicit Node(std::string name) : _name(std::move(name)) {}
  const char* get_name() const { return _name.c_str(); }
  std::shared_ptr<Clock> get_clock() { return _clock; }

  // declare_parameter keeps an earlier value (a launch-file / YAML override) like rclcpp does
  template <class T>
  void declare_parameter(const std::string& name, const T& def) {
    std::lock_guard<std::mutex> lk(_mx);
    if (!_declared.count(name)) _declared[name] = Parameter(to_value(def));     // what the code declares (name, type, default)
    if (_params.count(name)) return;
    _params[name] = Parameter(to_value(def));
  }
  /** every declare_parameter() so far with its DECLARED default (an earlier set_parameter does not show here) */
  std::map<std::string, Parameter> declared_parameters() const { std::lock_guard<std::mutex> lk(_mx); return _declared; }
  bool has_parameter(const std::string& name) const { std::lock_guard<std::mutex> lk(_mx); return _params.count(name) != 0; }
  Parameter get_parameter(const std::string& name) const {
    std::lock_guard<std::mutex> lk(_mx);
    auto it = _params.find(name);
    if (it == _params.end()) throw std::runtime_error("parameter not declared: " + name);
    return it->second;
  }
  template <class T>
  void set_parameter(const std::string& name, const T& v) { std::lock_guard<std::mutex> lk(_mx); _params[name] = Parameter(to_value(v)); }

  template <class MsgT>
  std::shared_ptr<Publisher<MsgT>> create_publisher(const std::string& topic, int /*qos*/ = 1) {
    return std::make_shared<Publisher<MsgT>>(topic);
  }
private:
  static Parameter::Value to_value(bool v) { return v; }
  static Parameter::Value to_value(int v) { return (int64_t)v; }
  static Parameter::Value to_value(int64_t v) { return v; }
  static Parameter::Value to_value(double v) { return v; }
  static Parameter::Value to_value(const char* v) { return std::string(v); }
  static Parameter::Value to_value(const std::string& v) { return v; }
  std::string _name;
  std::shared_ptr<Clock> _clock = std::make_shared<Clock>();
  mutable std::mutex _mx;
  std::map<std::string, Parameter> _params, _declared;
};

}  // namespace rclcpp

namespace tf2_ros {
class TransformBroadcaster {
public:
  explicit TransformBroadcaster(rclcpp::Node&) {}
  void sendTransform(const geometry_msgs::msg::TransformStamped& t) { std::lock_guard<std::mutex> lk(_mx); _last = t; _count++; }
  geometry_msgs::msg::TransformStamped last() const { std::lock_guard<std::mutex> lk(_mx); return _last; }
  uint64_t count() const { std::lock_guard<std::mutex> lk(_mx); return _count; }
private:
  mutable std::mutex _mx;
  geometry_msgs::msg::TransformStamped _last;
  uint64_t _count = 0;
};
}  // namespace tf2_ros
#endif
