// TEST INFRASTRUCTURE ONLY -- private stand-in (oracle/stubs_node/) so that the UNMODIFIED node source
// /root/reference/hector_mapping/src/HectorMappingRos.cpp compiles without ROS / tf / boost (none is in this image).
// Nothing here is part of the product; see oracle/node_shim.cpp.
#pragma once
#include <vector>
#include "geometry_msgs/Point32.h"
#include "std_msgs/Header.h"
namespace sensor_msgs {
struct ChannelFloat32 {
  std::string name;
  std::vector<float> values;
};
struct PointCloud {
  std_msgs::Header header;
  std::vector<geometry_msgs::Point32> points;
  std::vector<ChannelFloat32> channels;
};
}  // namespace sensor_msgs
