// TEST INFRASTRUCTURE ONLY -- private stand-in (oracle/stubs_node/) so that the UNMODIFIED node source
// /root/reference/hector_mapping/src/HectorMappingRos.cpp compiles without ROS / tf / boost (none is in this image).
// Nothing here is part of the product; see oracle/node_shim.cpp.
#pragma once
#include <string>
#include "ros/ros.h"
namespace std_msgs {
struct Header {
  uint32_t seq = 0;
  ros::Time stamp;
  std::string frame_id;
};
}  // namespace std_msgs
