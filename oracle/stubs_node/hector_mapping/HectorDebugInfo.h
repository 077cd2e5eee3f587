// TEST INFRASTRUCTURE ONLY -- private stand-in (oracle/stubs_node/) so that the UNMODIFIED node source
// /root/reference/hector_mapping/src/HectorMappingRos.cpp compiles without ROS / tf / boost (none is in this image).
// Nothing here is part of the product; see oracle/node_shim.cpp.
#pragma once
#include <vector>
namespace hector_mapping {
struct HectorIterData {  // float64[9] hessian, float64 conditionNum determinant conditionNum2d determinant2d
  boost::array<double, 9> hessian;
  double conditionNum = 0.0, determinant = 0.0, conditionNum2d = 0.0, determinant2d = 0.0;
};
struct HectorDebugInfo {
  std::vector<HectorIterData> iterData;
};
}  // namespace hector_mapping
