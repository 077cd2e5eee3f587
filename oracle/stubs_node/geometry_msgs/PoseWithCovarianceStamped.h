// TEST INFRASTRUCTURE ONLY -- private stand-in (oracle/stubs_node/) so that the UNMODIFIED node source
// /root/reference/hector_mapping/src/HectorMappingRos.cpp compiles without ROS / tf / boost (none is in this image).
// Nothing here is part of the product; see oracle/node_shim.cpp.
#pragma once
#include "geometry_msgs/Pose.h"
#include "std_msgs/Header.h"
namespace geometry_msgs {
struct PoseWithCovariance {
  Pose pose;
  boost::array<double, 36> covariance;
  PoseWithCovariance() { covariance.assign(0.0); }
};
struct PoseWithCovarianceStamped {
  std_msgs::Header header;
  PoseWithCovariance pose;
};
typedef boost::shared_ptr<PoseWithCovarianceStamped const> PoseWithCovarianceStampedConstPtr;
}  // namespace geometry_msgs
