// TEST INFRASTRUCTURE ONLY -- private stand-in (oracle/stubs_node/) so that the UNMODIFIED node source
// /root/reference/hector_mapping/src/HectorMappingRos.cpp compiles without ROS / tf / boost (none is in this image).
// Nothing here is part of the product; see oracle/node_shim.cpp.
#pragma once
#include "tf/transform_listener.h"
namespace tf {
template <class M> class MessageFilter {
 public:
  template <class S> MessageFilter(S&, TransformListener&, const std::string&, uint32_t) {}
  template <class F> void registerCallback(const F&) {}
};
}  // namespace tf
