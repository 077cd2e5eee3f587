// TEST INFRASTRUCTURE ONLY -- private stand-in (oracle/stubs_node/) so that the UNMODIFIED node source
// /root/reference/hector_mapping/src/HectorMappingRos.cpp compiles without ROS / tf / boost (none is in this image).
// Nothing here is part of the product; see oracle/node_shim.cpp.
#pragma once
#include "nav_msgs/OccupancyGrid.h"
namespace nav_msgs {
struct GetMap {
  struct Request {};
  struct Response {
    OccupancyGrid map;
  };
};
}  // namespace nav_msgs
