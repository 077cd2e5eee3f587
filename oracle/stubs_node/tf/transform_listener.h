// TEST INFRASTRUCTURE ONLY -- private stand-in (oracle/stubs_node/) so that the UNMODIFIED node source
// /root/reference/hector_mapping/src/HectorMappingRos.cpp compiles without ROS / tf / boost (none is in this image).
// Nothing here is part of the product; see oracle/node_shim.cpp.
#pragma once
#include "tf/transform_datatypes.h"
namespace tf {
// the tf tree of the stand-in: at most ONE static transform, base_frame <- scan frame (oracle/node_shim.cpp sets it for the
// node's default use_tf_scan_transformation path); every other lookup fails like an empty tree
struct StaticTree {
  bool have = false;
  std::string target, source;
  Transform t;
};
inline StaticTree& static_tree() {
  static StaticTree s;
  return s;
}
class TransformListener {
 public:
  bool waitForTransform(const std::string& target, const std::string& source, const ros::Time&, const ros::Duration&) const {
    const StaticTree& s = static_tree();
    return s.have && s.target == target && s.source == source;
  }
  void lookupTransform(const std::string& target, const std::string& source, const ros::Time& stamp, StampedTransform& out) const {
    const StaticTree& s = static_tree();
    if (!(s.have && s.target == target && s.source == source)) throw TransformException("oracle/stubs_node: no such transform");
    out = StampedTransform(s.t, stamp, target, source);
  }
};
}  // namespace tf
