// TEST INFRASTRUCTURE ONLY -- private stand-in (oracle/stubs_node/) so that the UNMODIFIED node source
// /root/reference/hector_mapping/src/HectorMappingRos.cpp compiles without ROS / tf / boost (none is in this image).
// Nothing here is part of the product; see oracle/node_shim.cpp.
#pragma once
#include <vector>
#include "visualization_msgs/Marker.h"
namespace visualization_msgs {
struct MarkerArray {
  std::vector<Marker> markers;
};
}  // namespace visualization_msgs
