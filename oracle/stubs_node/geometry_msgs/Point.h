// TEST INFRASTRUCTURE ONLY -- private stand-in (oracle/stubs_node/) so that the UNMODIFIED node source
// /root/reference/hector_mapping/src/HectorMappingRos.cpp compiles without ROS / tf / boost (none is in this image).
// Nothing here is part of the product; see oracle/node_shim.cpp.
#pragma once
namespace geometry_msgs {
struct Point {
  double x = 0.0, y = 0.0, z = 0.0;
};
}  // namespace geometry_msgs
