// TEST INFRASTRUCTURE ONLY -- private stand-in (oracle/stubs_node/) so that the UNMODIFIED node source
// /root/reference/hector_mapping/src/HectorMappingRos.cpp compiles without ROS / tf / boost (none is in this image).
// Nothing here is part of the product; see oracle/node_shim.cpp.
#pragma once
#include <string>
#include "geometry_msgs/Pose.h"
#include "geometry_msgs/Vector3.h"
#include "std_msgs/Header.h"
namespace visualization_msgs {
struct Marker {
  enum { ARROW = 0, CUBE = 1, SPHERE = 2, CYLINDER = 3, ADD = 0 };
  std_msgs::Header header;
  std::string ns;
  int32_t id = 0, type = 0, action = 0;
  geometry_msgs::Pose pose;
  geometry_msgs::Vector3 scale;
  struct ColorRGBA {
    float r = 0.0f, g = 0.0f, b = 0.0f, a = 0.0f;
  } color;
};
}  // namespace visualization_msgs
