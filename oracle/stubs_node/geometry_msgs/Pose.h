// TEST INFRASTRUCTURE ONLY -- private stand-in (oracle/stubs_node/) so that the UNMODIFIED node source
// /root/reference/hector_mapping/src/HectorMappingRos.cpp compiles without ROS / tf / boost (none is in this image).
// Nothing here is part of the product; see oracle/node_shim.cpp.
#pragma once
#include "geometry_msgs/Point.h"
#include "geometry_msgs/Quaternion.h"
namespace geometry_msgs {
struct Pose {
  Point position;
  Quaternion orientation;
};
}  // namespace geometry_msgs
