// TEST INFRASTRUCTURE ONLY -- private stand-in (oracle/stubs_node/) so that the UNMODIFIED node source
// /root/reference/hector_mapping/src/HectorMappingRos.cpp compiles without ROS / tf / boost (none is in this image).
// Nothing here is part of the product; see oracle/node_shim.cpp.
#pragma once
namespace geometry_msgs {
struct Point32 {  // float32 x y z
  float x = 0.0f, y = 0.0f, z = 0.0f;
};
}  // namespace geometry_msgs
