// TEST INFRASTRUCTURE ONLY -- private stand-in (oracle/stubs_node/) so that the UNMODIFIED node source
// /root/reference/hector_mapping/src/HectorMappingRos.cpp compiles without ROS / tf / boost (none is in this image).
// Nothing here is part of the product; see oracle/node_shim.cpp.
#pragma once
// (hector_slam_lib/util/UtilFunctions.h:33 includes this one; in the node's translation unit the full set is wanted)
#include "tf/transform_datatypes.h"
