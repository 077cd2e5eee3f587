// TEST INFRASTRUCTURE ONLY -- private stand-in (oracle/stubs_node/) so that the UNMODIFIED node source
// /root/reference/hector_mapping/src/HectorMappingRos.cpp compiles without ROS / tf / boost (none is in this image).
// Nothing here is part of the product; see oracle/node_shim.cpp.
#pragma once
#include "geometry_msgs/Pose.h"
#include "ros/ros.h"
namespace nav_msgs {
struct MapMetaData {  // time map_load_time, float32 resolution, uint32 width height, geometry_msgs/Pose origin
  ros::Time map_load_time;
  float resolution = 0.0f;
  uint32_t width = 0, height = 0;
  geometry_msgs::Pose origin;
};
}  // namespace nav_msgs
