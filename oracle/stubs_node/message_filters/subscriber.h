// TEST INFRASTRUCTURE ONLY -- private stand-in (oracle/stubs_node/) so that the UNMODIFIED node source
// /root/reference/hector_mapping/src/HectorMappingRos.cpp compiles without ROS / tf / boost (none is in this image).
// Nothing here is part of the product; see oracle/node_shim.cpp.
#pragma once
#include "ros/ros.h"
namespace message_filters {
template <class M> class Subscriber {
 public:
  Subscriber(ros::NodeHandle&, const std::string&, uint32_t) {}
};
}  // namespace message_filters
