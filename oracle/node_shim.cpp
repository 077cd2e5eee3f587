// TEST INFRASTRUCTURE ONLY -- the reference's ROS node source, UNMODIFIED, as a CPU checker for SURVEY.md rows f1 / f2.
//
// hector_mapping/src/HectorMappingRos.cpp (and PoseInfoContainer.cpp) are compiled where they lie under /root/reference
// (found through -I, nothing is copied) against private stand-ins for roscpp / tf / the message headers / boost /
// laser_geometry (oracle/stubs_node/) and the Eigen stand-in of the library checker (oracle/stubs/).  Exposed: the node's own
//   HectorMappingRos::rosLaserScanToDataContainer    hector_mapping/src/HectorMappingRos.cpp:483-507   (row f1)
//   HectorMappingRos::rosPointCloudToDataContainer   :509-542                                          (row f1; the node's default)
//   scanCallback's projector_.projectLaser + the above, :273-282  (projectLaser itself is laser_geometry: third party, restated
//                                                        in stubs_node/laser_geometry/laser_geometry.h)
//   HectorMappingRos::publishMap / setServiceGetMapData  :435-481, :544-560                            (row f2 and the map metadata)
// through a plain C API.  What is "the reference" here and what is a stand-in: the control flow, the gates, the float / double
// types and every arithmetic expression of those functions are the reference's; `tf::Transform * tf::Vector3` and the message
// structs are stand-ins that restate tf 1.12's LinearMath and the .msg definitions (absent from /root/reference, like Eigen).
// The node is constructed once per handle with the parameters the functions read (laser_min/max_dist, laser_z_min/max_value;
// a small map so that the constructor's HectorSlamProcessor is cheap); no ROS master, no threads, no tf tree.
// Output: oracle/_ref/libhector_node_ref.so (oracle/Makefile `node`).  Only tests/ load it (oracle/pyoracle.py NodeRef).
#include <math.h>
#include <stdlib.h>

#include <iostream>

#include "HectorMappingRos.cpp"   // the node, as shipped
#include "PoseInfoContainer.cpp"  // its pose bookkeeping (scanCallback links against it)

namespace {
// the node keeps its processor and its map publisher protected; a derived class may look
struct NodeAccess : public HectorMappingRos {
  hectorslam::HectorSlamProcessor* proc() { return slamProcessor; }
  MapPublisherContainer& pub0() { return mapPubContainer[0]; }
};
struct Node {
  NodeAccess* n;
};
}  // namespace

extern "C" {

static tf::StampedTransform transform_of(const double T[12]);

// the one transform of the stand-in tf tree: base_link <- laser (rows [R | t]); NULL = empty tree.  Set BEFORE hn_create: a node
// created with a transform runs scanCallback's default path (:260-327: lookupTransform, projectLaser, rosPointCloudToDataContainer,
// update from the last pose), a node created without runs the LaserScan path (:253-259)
void hn_set_laser_transform(const double* T) {
  tf::StaticTree& s = tf::static_tree();
  s.have = T != nullptr;
  if (T) {
    s.target = "base_link";  // p_base_frame_'s default (:82)
    s.source = "laser";      // the frame_id hn_scan_callback stamps on its messages
    s.t = transform_of(T);
  }
}

// The node with the parameters its constructor reads from the parameter server (HectorMappingRos.cpp:56-107).  map_size <= 0:
// a 64-cell single-level map (handles that only convert containers / publish given cells).  The scan path is the one without
// the tf tree (use_tf_scan_transformation = false: rosLaserScanToDataContainer + update from the last pose, :253-259).
void* hn_create(double laser_min_dist, double laser_max_dist, double laser_z_min, double laser_z_max, int map_size, int levels,
                double resolution, double update_dist_thresh, double update_angle_thresh, double factor_free, double factor_occ) {
  std::map<std::string, double>& p = ros::param_numbers();
  p.clear();
  ros::param_strings().clear();
  p["map_size"] = map_size > 0 ? map_size : 64;
  p["map_multi_res_levels"] = map_size > 0 ? levels : 1;
  p["map_resolution"] = map_size > 0 ? resolution : 0.05;
  p["map_update_distance_thresh"] = update_dist_thresh;
  p["map_update_angle_thresh"] = update_angle_thresh;
  p["update_factor_free"] = factor_free;
  p["update_factor_occupied"] = factor_occ;
  p["use_tf_scan_transformation"] = tf::static_tree().have ? 1 : 0;  // hn_set_laser_transform() before hn_create() selects the tf path
  p["laser_min_dist"] = laser_min_dist;  // -> p_sqr_laser_min_dist_ = (float)(d * d), :95-100
  p["laser_max_dist"] = laser_max_dist;
  p["laser_z_min_value"] = laser_z_min;
  p["laser_z_max_value"] = laser_z_max;
  Node* h = new Node;
  std::streambuf* out = std::cout.rdbuf(nullptr);  // the map representation prints a banner per level (MapRepMultiMap.h:60)
  h->n = new NodeAccess();
  std::cout.rdbuf(out);
  return h;
}
void hn_destroy(void* h) {
  if (!h) return;
  delete ((Node*)h)->n;
  delete (Node*)h;
}

static int take(const hectorslam::DataContainer& c, float* out_xy, float origo[2]) {
  const int m = c.getSize();
  for (int i = 0; i < m; ++i) {
    out_xy[2 * i] = c.getVecEntry(i)[0];
    out_xy[2 * i + 1] = c.getVecEntry(i)[1];
  }
  if (origo) {
    const Eigen::Vector2f o = c.getOrigo();
    origo[0] = o[0];
    origo[1] = o[1];
  }
  return m;
}

// rosLaserScanToDataContainer (:483-507)
int hn_laser_scan_to_container(void* h, const float* ranges, int n, float angle_min, float angle_increment, float range_min,
                               float range_max, float scale_to_map, float* out_xy, float origo[2]) {
  sensor_msgs::LaserScan scan;
  scan.angle_min = angle_min;
  scan.angle_increment = angle_increment;
  scan.range_min = range_min;
  scan.range_max = range_max;
  scan.ranges.assign(ranges, ranges + n);
  hectorslam::DataContainer c;
  ((Node*)h)->n->rosLaserScanToDataContainer(scan, c, scale_to_map);
  return take(c, out_xy, origo);
}

tf::StampedTransform transform_of(const double T[12]) {  // rows [R | t] of the laser -> base transform
  tf::Matrix3x3 b;
  b.setValue(T[0], T[1], T[2], T[4], T[5], T[6], T[8], T[9], T[10]);
  return tf::StampedTransform(tf::Transform(b, tf::Vector3(T[3], T[7], T[11])), ros::Time(), "base_link", "laser");
}

// rosPointCloudToDataContainer (:509-542)
int hn_point_cloud_to_container(void* h, const float* pts_xyz, int n, const double T[12], float scale_to_map, float* out_xy,
                                float origo[2]) {
  sensor_msgs::PointCloud cloud;
  cloud.points.resize(n);
  for (int i = 0; i < n; ++i) {
    cloud.points[i].x = pts_xyz[3 * i];
    cloud.points[i].y = pts_xyz[3 * i + 1];
    cloud.points[i].z = pts_xyz[3 * i + 2];
  }
  hectorslam::DataContainer c;
  ((Node*)h)->n->rosPointCloudToDataContainer(cloud, transform_of(T), c, scale_to_map);
  return take(c, out_xy, origo);
}

// scanCallback's default ingestion (:273-282): projector_.projectLaser(scan, cloud, 30.0) -> rosPointCloudToDataContainer
int hn_project_and_convert(void* h, const float* ranges, int n, float angle_min, float angle_increment, float range_min,
                           float range_max, double range_cutoff, const double T[12], float scale_to_map, float* out_xy,
                           float origo[2], float* out_cloud_xyz, int* out_cloud_n) {
  sensor_msgs::LaserScan scan;
  scan.angle_min = angle_min;
  scan.angle_increment = angle_increment;
  scan.range_min = range_min;
  scan.range_max = range_max;
  scan.ranges.assign(ranges, ranges + n);
  laser_geometry::LaserProjection projector;
  sensor_msgs::PointCloud cloud;
  projector.projectLaser(scan, cloud, range_cutoff);
  if (out_cloud_xyz)
    for (size_t i = 0; i < cloud.points.size(); ++i) {
      out_cloud_xyz[3 * i] = cloud.points[i].x;
      out_cloud_xyz[3 * i + 1] = cloud.points[i].y;
      out_cloud_xyz[3 * i + 2] = cloud.points[i].z;
    }
  if (out_cloud_n) *out_cloud_n = (int)cloud.points.size();
  hectorslam::DataContainer c;
  ((Node*)h)->n->rosPointCloudToDataContainer(cloud, transform_of(T), c, scale_to_map);
  return take(c, out_xy, origo);
}

// HectorMappingRos::scanCallback (:232-370) on one LaserScan message: conversion, HectorSlamProcessor::update from the last
// pose, pose bookkeeping, the (stand-in) publishers.  -> the processor's pose and covariance afterwards
void hn_scan_callback(void* h, const float* ranges, int n, float angle_min, float angle_increment, float range_min, float range_max,
                      float out_pose[3], float out_cov[9]) {
  sensor_msgs::LaserScan scan;
  scan.header.frame_id = "laser";
  scan.angle_min = angle_min;
  scan.angle_increment = angle_increment;
  scan.range_min = range_min;
  scan.range_max = range_max;
  scan.ranges.assign(ranges, ranges + n);
  NodeAccess* node = ((Node*)h)->n;
  node->scanCallback(scan);
  const Eigen::Vector3f& pose = node->proc()->getLastScanMatchPose();
  const Eigen::Matrix3f& cov = node->proc()->getLastScanMatchCovariance();
  for (int i = 0; i < 3; ++i) out_pose[i] = pose[i];
  for (int i = 0; i < 9; ++i) out_cov[i] = cov.data()[i];
}

// publishMap (:435-476) of the node's OWN level-0 map, as publishMapLoop calls it: -> cells [size * size], update index
int hn_node_map(void* h, signed char* out_cells, float* out_logodds) {
  NodeAccess* node = ((Node*)h)->n;
  const hectorslam::GridMap& grid = node->proc()->getGridMap(0);
  node->publishMap(node->pub0(), grid, ros::Time(), node->proc()->getMapMutex(0));
  const int size = grid.getSizeX() * grid.getSizeY();
  for (int i = 0; i < size; ++i) {
    out_cells[i] = node->pub0().map_.map.data[i];
    if (out_logodds) out_logodds[i] = grid.getCell(i).getValue();
  }
  return grid.getUpdateIndex();
}

// publishMap's cell loop (:435-476) and setServiceGetMapData (:544-560) on a grid with the given cells:
// out_cells [sx * sy] int8 (-1 / 0 / 100), out_meta = {origin x, origin y, resolution, width, height}
void hn_publish_map(void* h, float resolution, int sx, int sy, float start_x, float start_y, const float* logodds, signed char* out_cells,
                    double out_meta[5]) {
  // the grid as MapRepMultiMap's constructor makes level 0 (MapRepMultiMap.h:48-72)
  const Eigen::Vector2i size(sx, sy);
  const float totalX = resolution * static_cast<float>(sx), totalY = resolution * static_cast<float>(sy);
  const Eigen::Vector2f offset(totalX * start_x, totalY * start_y);  // mid_offset_x / _y
  hectorslam::GridMap grid(resolution, size, offset);
  for (int i = 0; i < sx * sy; ++i) grid.getCell(i).logOddsVal = logodds[i];
  grid.setUpdated();
  MapPublisherContainer pub;
  HectorMappingRos* n = ((Node*)h)->n;
  n->setServiceGetMapData(pub.map_, grid);
  n->publishMap(pub, grid, ros::Time(), 0);
  for (int i = 0; i < sx * sy; ++i) out_cells[i] = pub.map_.map.data[i];
  out_meta[0] = pub.map_.map.info.origin.position.x;
  out_meta[1] = pub.map_.map.info.origin.position.y;
  out_meta[2] = pub.map_.map.info.resolution;
  out_meta[3] = pub.map_.map.info.width;
  out_meta[4] = pub.map_.map.info.height;
}

}  // extern "C"
