// TEST INFRASTRUCTURE ONLY -- private stand-in (oracle/stubs_node/) so that the UNMODIFIED node source
// /root/reference/hector_mapping/src/HectorMappingRos.cpp compiles without ROS / tf / boost (none is in this image).
// Nothing here is part of the product; see oracle/node_shim.cpp.
#pragma once
// the subset of boost the node's translation unit touches: thread (never started here), mutex, bind (never called),
// lexical_cast (to std::string only), array, shared_ptr
#include <stddef.h>

#include <memory>
#include <mutex>
#include <sstream>
#include <string>
namespace boost {
class mutex {
 public:
  void lock() { m_.lock(); }
  void unlock() { m_.unlock(); }
 private:
  std::mutex m_;
};
// HectorMappingRos starts publishMapLoop on a boost::thread; the shim has no publisher loop to run (ros::ok() is false)
class thread {
 public:
  template <class F> explicit thread(F) {}
  void join() {}
};
struct bound_call {};
template <class... A> inline bound_call bind(A&&...) { return bound_call(); }
template <class T, class S> inline T lexical_cast(const S& s) {
  std::ostringstream o;
  o << s;
  return o.str();
}
template <class T, size_t N> struct array {
  T elems[N];
  T& operator[](size_t i) { return elems[i]; }
  const T& operator[](size_t i) const { return elems[i]; }
  static size_t size() { return N; }
  void assign(const T& v) { for (size_t i = 0; i < N; ++i) elems[i] = v; }
};
template <class T> using shared_ptr = std::shared_ptr<T>;
}  // namespace boost
namespace { struct hsm_stub_placeholder {} _1, _2; }  // boost::bind's global placeholders
