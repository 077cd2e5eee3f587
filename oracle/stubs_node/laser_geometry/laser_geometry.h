// TEST INFRASTRUCTURE ONLY -- private stand-in (oracle/stubs_node/) so that the UNMODIFIED node source
// /root/reference/hector_mapping/src/HectorMappingRos.cpp compiles without ROS / tf / boost (none is in this image).
// Nothing here is part of the product; see oracle/node_shim.cpp.
#pragma once
// laser_geometry 1.6.x LaserProjection::projectLaser (third party, not in /root/reference): the restatement the oracle
// already carries (oracle/hector_oracle.cpp ho_project_laser) -- double unit vectors cos/sin(angle_min + i * increment),
// double range * unit vector narrowed to float32, kept iff range < range_cutoff && range >= range_min
#include "sensor_msgs/LaserScan.h"
#include "sensor_msgs/PointCloud.h"
namespace laser_geometry {
class LaserProjection {
 public:
  void projectLaser(const sensor_msgs::LaserScan& scan, sensor_msgs::PointCloud& cloud, double range_cutoff = -1.0, int = 0) {
    if (range_cutoff < 0) range_cutoff = scan.range_max;
    cloud.header = scan.header;
    cloud.points.clear();
    const double a0 = scan.angle_min, inc = scan.angle_increment;
    for (size_t i = 0; i < scan.ranges.size(); ++i) {
      const double r = (double)scan.ranges[i];
      const double ox = r * cos(a0 + (double)i * inc), oy = r * sin(a0 + (double)i * inc);
      const float range = (float)r;
      if ((range < range_cutoff) && (range >= scan.range_min)) {
        geometry_msgs::Point32 p;
        p.x = (float)ox;
        p.y = (float)oy;
        p.z = 0.0f;
        cloud.points.push_back(p);
      }
    }
  }
};
}  // namespace laser_geometry
