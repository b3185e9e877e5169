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
#include <tf2_ros/transform_listener.h>
#include <tf2_ros/buffer.h>
#include <tf2/LinearMath/Transform.h>
#include <tf2_geometry_msgs/tf2_geometry_msgs.hpp>
#else
#define OHM_TSD_SLAM_HAVE_ROS 0
#include <chrono>
#include <cstdint>
#include <map>
#include <cmath>
#include <memory>
#include <mutex>
#include <stdexcept>
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

// tf2's linear algebra and the tf buffer, as far as ThreadLocalize::sendTransform uses them (ThreadLocalize.cpp:603-689):
// Quaternion::setEuler, Transform::mult / setOrigin / setRotation, fromMsg / toMsg, Buffer::canTransform and
// Buffer::lookupTransform throwing tf2::TransformException.  The arithmetic follows tf2's published LinearMath (Quaternion.h setEuler, Matrix3x3.h
// setRotation / getRotation, Transform.h mult) so that the message the stand-in produces is the one the real library would.
namespace tf2 {
class TransformException : public std::runtime_error { public: explicit TransformException(const std::string& m) : std::runtime_error(m) {} };
class LookupException : public TransformException { public: explicit LookupException(const std::string& m) : TransformException(m) {} };
using TimePoint = std::chrono::time_point<std::chrono::system_clock, std::chrono::nanoseconds>;
static const TimePoint TimePointZero = TimePoint(std::chrono::nanoseconds(0));

class Vector3 {
public:
  Vector3() = default;
  Vector3(double x, double y, double z) : _v{x, y, z} {}
  double x() const { return _v[0]; }
  double y() const { return _v[1]; }
  double z() const { return _v[2]; }
  double operator[](int i) const { return _v[i]; }
private:
  double _v[3] = {0, 0, 0};
};

class Quaternion {
public:
  Quaternion() = default;
  Quaternion(double x, double y, double z, double w) : _q{x, y, z, w} {}
  // tf2's argument order: yaw about Y, pitch about X, roll about Z -- sendTransform passes (0, 0, theta)
  void setEuler(double yaw, double pitch, double roll) {
    const double hy = yaw * 0.5, hp = pitch * 0.5, hr = roll * 0.5;
    const double cy = std::cos(hy), sy = std::sin(hy), cp = std::cos(hp), sp = std::sin(hp), cr = std::cos(hr), sr = std::sin(hr);
    _q[0] = cr * sp * cy + sr * cp * sy;
    _q[1] = cr * cp * sy - sr * sp * cy;
    _q[2] = sr * cp * cy - cr * sp * sy;
    _q[3] = cr * cp * cy + sr * sp * sy;
  }
  double x() const { return _q[0]; }
  double y() const { return _q[1]; }
  double z() const { return _q[2]; }
  double w() const { return _q[3]; }
  double length2() const { return _q[0] * _q[0] + _q[1] * _q[1] + _q[2] * _q[2] + _q[3] * _q[3]; }
private:
  double _q[4] = {0, 0, 0, 1};
};

class Transform {
public:
  Transform() { for(int i = 0; i < 3; i++) for(int j = 0; j < 3; j++) _m[i][j] = (i == j) ? 1.0 : 0.0; }
  void setOrigin(const Vector3& o) { _o = o; }
  const Vector3& getOrigin() const { return _o; }
  void setRotation(const Quaternion& q) {           // Matrix3x3::setRotation
    const double d = q.length2(), s = 2.0 / d;
    const double xs = q.x() * s, ys = q.y() * s, zs = q.z() * s;
    const double wx = q.w() * xs, wy = q.w() * ys, wz = q.w() * zs;
    const double xx = q.x() * xs, xy = q.x() * ys, xz = q.x() * zs;
    const double yy = q.y() * ys, yz = q.y() * zs, zz = q.z() * zs;
    _m[0][0] = 1.0 - (yy + zz); _m[0][1] = xy - wz;         _m[0][2] = xz + wy;
    _m[1][0] = xy + wz;         _m[1][1] = 1.0 - (xx + zz); _m[1][2] = yz - wx;
    _m[2][0] = xz - wy;         _m[2][1] = yz + wx;         _m[2][2] = 1.0 - (xx + yy);
  }
  Quaternion getRotation() const {                  // Matrix3x3::getRotation
    const double trace = _m[0][0] + _m[1][1] + _m[2][2];
    double t[4];
    if(trace > 0.0) {
      double s = std::sqrt(trace + 1.0);
      t[3] = s * 0.5; s = 0.5 / s;
      t[0] = (_m[2][1] - _m[1][2]) * s; t[1] = (_m[0][2] - _m[2][0]) * s; t[2] = (_m[1][0] - _m[0][1]) * s;
    } else {
      const int i = _m[0][0] < _m[1][1] ? (_m[1][1] < _m[2][2] ? 2 : 1) : (_m[0][0] < _m[2][2] ? 2 : 0);
      const int j = (i + 1) % 3, k = (i + 2) % 3;
      double s = std::sqrt(_m[i][i] - _m[j][j] - _m[k][k] + 1.0);
      t[i] = s * 0.5; s = 0.5 / s;
      t[3] = (_m[k][j] - _m[j][k]) * s; t[j] = (_m[j][i] + _m[i][j]) * s; t[k] = (_m[k][i] + _m[i][k]) * s;
    }
    return Quaternion(t[0], t[1], t[2], t[3]);
  }
  // this = t1 * t2: basis = b1 * b2, origin = b1 * o2 + o1
  void mult(const Transform& t1, const Transform& t2) {
    double m[3][3], o[3];
    for(int i = 0; i < 3; i++) {
      for(int j = 0; j < 3; j++) m[i][j] = t1._m[i][0] * t2._m[0][j] + t1._m[i][1] * t2._m[1][j] + t1._m[i][2] * t2._m[2][j];
      o[i] = t1._m[i][0] * t2._o[0] + t1._m[i][1] * t2._o[1] + t1._m[i][2] * t2._o[2] + t1._o[i];
    }
    for(int i = 0; i < 3; i++) for(int j = 0; j < 3; j++) _m[i][j] = m[i][j];
    _o = Vector3(o[0], o[1], o[2]);
  }
  Transform inverse() const {
    Transform r;
    for(int i = 0; i < 3; i++) for(int j = 0; j < 3; j++) r._m[i][j] = _m[j][i];
    double o[3];
    for(int i = 0; i < 3; i++) o[i] = -(r._m[i][0] * _o[0] + r._m[i][1] * _o[1] + r._m[i][2] * _o[2]);
    r._o = Vector3(o[0], o[1], o[2]);
    return r;
  }
private:
  double _m[3][3];
  Vector3 _o;
};

inline void fromMsg(const geometry_msgs::msg::Transform& in, Transform& out) {
  out.setOrigin(Vector3(in.translation.x, in.translation.y, in.translation.z));
  out.setRotation(Quaternion(in.rotation.x, in.rotation.y, in.rotation.z, in.rotation.w));
}
inline geometry_msgs::msg::Transform toMsg(const Transform& in) {
  geometry_msgs::msg::Transform out;
  out.translation.x = in.getOrigin().x(); out.translation.y = in.getOrigin().y(); out.translation.z = in.getOrigin().z();
  const Quaternion q = in.getRotation();
  out.rotation.x = q.x(); out.rotation.y = q.y(); out.rotation.z = q.z(); out.rotation.w = q.w();
  return out;
}
}  // namespace tf2

namespace tf2_ros {
// the tf tree as a set of edges: setTransform stores "child in parent" (header.frame_id = parent), lookupTransform(target,
// source) answers from a stored edge, its inverse or -- for target == source -- the identity, and throws tf2::LookupException
// otherwise, which is what the real buffer does while no tree has been heard (the case the reference catches)
class Buffer {
public:
  explicit Buffer(std::shared_ptr<rclcpp::Clock> = nullptr) {}
  bool setTransform(const geometry_msgs::msg::TransformStamped& t, const std::string& /*authority*/, bool /*is_static*/ = false) {
    if(t.header.frame_id.empty() || t.child_frame_id.empty() || t.header.frame_id == t.child_frame_id) return false;
    std::lock_guard<std::mutex> lk(_mx);
    _edges[std::make_pair(t.header.frame_id, t.child_frame_id)] = t;
    return true;
  }
  void clear() { std::lock_guard<std::mutex> lk(_mx); _edges.clear(); }
  // tf2::BufferCore::canTransform: the same question without an exception (false + the reason in *error_msg)
  bool canTransform(const std::string& target, const std::string& source, const tf2::TimePoint&, std::string* error_msg = nullptr) const {
    std::lock_guard<std::mutex> lk(_mx);
    if(_edges.count(std::make_pair(target, source)) || _edges.count(std::make_pair(source, target))) return true;
    if(target == source)
      for(const auto& e : _edges)
        if(e.first.first == target || e.first.second == target) return true;
    if(error_msg) *error_msg = notConnected(target, source);
    return false;
  }
  geometry_msgs::msg::TransformStamped lookupTransform(const std::string& target, const std::string& source, const tf2::TimePoint&) const {
    std::lock_guard<std::mutex> lk(_mx);
    auto it = _edges.find(std::make_pair(target, source));
    if(it != _edges.end()) return it->second;
    it = _edges.find(std::make_pair(source, target));
    if(it != _edges.end()) {
      tf2::Transform t;
      tf2::fromMsg(it->second.transform, t);
      geometry_msgs::msg::TransformStamped r;
      r.header.stamp = it->second.header.stamp; r.header.frame_id = target; r.child_frame_id = source;
      r.transform = tf2::toMsg(t.inverse());
      return r;
    }
    if(target == source)
      for(const auto& e : _edges)
        if(e.first.first == target || e.first.second == target) {
          geometry_msgs::msg::TransformStamped r;
          r.header.frame_id = target; r.child_frame_id = source;
          return r;
        }
    throw tf2::LookupException(notConnected(target, source));
  }
private:
  static std::string notConnected(const std::string& target, const std::string& source) {
    return "\"" + target + "\" passed to lookupTransform argument target_frame does not exist or is not connected to \"" + source + "\"";
  }
  mutable std::mutex _mx;
  std::map<std::pair<std::string, std::string>, geometry_msgs::msg::TransformStamped> _edges;
};
class TransformListener {
public:
  explicit TransformListener(Buffer&) {}
};
}  // namespace tf2_ros
#endif
