// TEST INFRASTRUCTURE ONLY -- private stand-in (oracle/stubs_node/) so that the UNMODIFIED node source
// /root/reference/hector_mapping/src/HectorMappingRos.cpp compiles without ROS / tf / boost (none is in this image).
// Nothing here is part of the product; see oracle/node_shim.cpp.
#pragma once
#include <vector>
#include "std_msgs/Header.h"
namespace sensor_msgs {
struct LaserScan {  // float32 angle_min angle_max angle_increment time_increment scan_time range_min range_max, float32[] ranges intensities
  std_msgs::Header header;
  float angle_min = 0.0f, angle_max = 0.0f, angle_increment = 0.0f, time_increment = 0.0f, scan_time = 0.0f, range_min = 0.0f,
        range_max = 0.0f;
  std::vector<float> ranges, intensities;
};
}  // namespace sensor_msgs
