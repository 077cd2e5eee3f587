// TEST INFRASTRUCTURE ONLY -- private stand-in (oracle/stubs_node/) so that the UNMODIFIED node source
// /root/reference/hector_mapping/src/HectorMappingRos.cpp compiles without ROS / tf / boost (none is in this image).
// Nothing here is part of the product; see oracle/node_shim.cpp.
#pragma once
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <map>
#include <string>
#include <vector>

#include "boost/thread.hpp"
namespace ros {
struct Time {
  uint32_t sec = 0, nsec = 0;
  Time() {}
  Time(uint32_t s, uint32_t n) : sec(s), nsec(n) {}
  static Time now() { return Time(); }
};
struct Duration {
  double d;
  explicit Duration(double s = 0.0) : d(s) {}
};
struct WallDuration {
  double toSec() const { return 0.0; }
};
struct WallTime {
  static WallTime now() { return WallTime(); }
  WallDuration operator-(const WallTime&) const { return WallDuration(); }
};
struct Rate {
  explicit Rate(double) {}
  void sleep() {}
};
inline bool ok() { return false; }  // no node is running: publishMapLoop (never started) would return at once
struct Publisher {
  template <class M> void publish(const M&) const {}
  int getNumSubscribers() const { return 0; }
};
struct Subscriber {};
struct ServiceServer {};
// the parameter server: oracle/node_shim.cpp fills it before it constructs the node
inline std::map<std::string, double>& param_numbers() {
  static std::map<std::string, double> m;
  return m;
}
inline std::map<std::string, std::string>& param_strings() {
  static std::map<std::string, std::string> m;
  return m;
}
class NodeHandle {
 public:
  NodeHandle() {}
  explicit NodeHandle(const std::string&) {}
  template <class T> bool param(const std::string& name, T& v, const T& def) const {
    auto it = param_numbers().find(name);
    if (it == param_numbers().end()) {
      v = def;
      return false;
    }
    v = static_cast<T>(it->second);
    return true;
  }
  bool param(const std::string& name, std::string& v, const std::string& def) const {
    auto it = param_strings().find(name);
    v = it == param_strings().end() ? def : it->second;
    return it != param_strings().end();
  }
  template <class M> Publisher advertise(const std::string&, uint32_t, bool = false) { return Publisher(); }
  template <class C, class Req, class Res> ServiceServer advertiseService(const std::string&, bool (C::*)(Req&, Res&), C*) {
    return ServiceServer();
  }
  template <class C, class M> Subscriber subscribe(const std::string&, uint32_t, void (C::*)(const M&), C*) { return Subscriber(); }
};
}  // namespace ros
#define ROS_INFO(...) do { } while (0)
#define ROS_ERROR(...) do { } while (0)
#define ROS_WARN(...) do { } while (0)
#define ROS_ASSERT(x) do { (void)(x); } while (0)
