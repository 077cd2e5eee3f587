// TEST INFRASTRUCTURE ONLY -- private stand-in (oracle/stubs_node/) so that the UNMODIFIED node source
// /root/reference/hector_mapping/src/HectorMappingRos.cpp compiles without ROS / tf / boost (none is in this image).
// Nothing here is part of the product; see oracle/node_shim.cpp.
#pragma once
#include "tf/transform_datatypes.h"
namespace tf {
class TransformListener {
 public:
  bool waitForTransform(const std::string&, const std::string&, const ros::Time&, const ros::Duration&) const { return false; }
  void lookupTransform(const std::string&, const std::string&, const ros::Time&, StampedTransform&) const {
    throw TransformException("oracle/stubs_node: no tf tree");
  }
};
}  // namespace tf
