/* TEST INFRASTRUCTURE ONLY -- C API shared by the two CPU checkers:
 *
 *   ho_*  oracle/hector_oracle.cpp   plain-C++ restatement of the reference path
 *   hr_*  oracle/ref_shim.cpp        the UNMODIFIED reference headers from
 *                                    /root/reference compiled through the private
 *                                    Eigen/tf stand-in (oracle/stubs/) -> oracle/_ref/
 *
 * Both export the same entry points (prefix differs) so tests can drive either.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * these libraries; the product (libhector_mi355.so) never links or calls them.
 *
 * Conventions (all follow the reference):
 *   poses    float[3] = x, y, theta      world: metres/rad, map: cells/rad
 *   pts      float[2*n] AoS endpoints, robot frame, LEVEL-0 cell units
 *            (DataPointContainer.h:92-96); *_level entry points take points
 *            already scaled for that level (what DataContainer::setFrom produced)
 *   cov/H    float[9] column-major 3x3 (Eigen default)
 *   planes   row-major, index = y*sizeX + x (GridMapBase.h:141-144)
 */
#ifndef HECTOR_ORACLE_API_H
#define HECTOR_ORACLE_API_H

#ifndef ORACLE_PREFIX
#error "define ORACLE_PREFIX (ho_ or hr_) before including oracle_api.h"
#endif
#define OR_CAT2(a, b) a##b
#define OR_CAT(a, b) OR_CAT2(a, b)
#define ORF(name) OR_CAT(ORACLE_PREFIX, name)

#ifdef __cplusplus
extern "C" {
#endif

/* HectorSlamProcessor ctor (HectorSlamProcessor.h:54-64) -> MapRepMultiMap ctor
 * (MapRepMultiMap.h:48-72). */
void* ORF(create)(float map_resolution, int size_x, int size_y, unsigned levels,
                  float start_x, float start_y);
void ORF(destroy)(void* h);
void ORF(reset)(void* h);                                   /* HectorSlamProcessor::reset :115-124 */
int ORF(levels)(void* h);                                   /* getMapLevels */
float ORF(scale_to_map)(void* h);                           /* getScaleToMap (level 0) */
void ORF(set_update_factor_free)(void* h, float f);         /* MapRepMultiMap.h:149-157 */
void ORF(set_update_factor_occupied)(void* h, float f);     /* MapRepMultiMap.h:159-167 */
void ORF(level_info)(void* h, int level, int* sx, int* sy, float* cell_length, float* scale_to_map);
void ORF(download_level)(void* h, int level, float* logodds, int* update_index);
/* overwrite cells, then invalidate the probability cache (like onMapUpdated) */
void ORF(upload_level)(void* h, int level, const float* logodds, const int* update_index);

void ORF(map_coords_pose)(void* h, int level, const float world[3], float map[3]);   /* GridMapBase.h:235-239 */
void ORF(world_coords_pose)(void* h, int level, const float map[3], float world[3]); /* GridMapBase.h:226-230 */

/* a1: OccGridMapUtil::interpMapValueWithDerivatives (OccGridMapUtil.h:287-347), n coords */
void ORF(interp)(void* h, int level, const float* coords_xy, int n, float* out_mgxgy);
/* a2: OccGridMapUtil::getCompleteHessianDerivs (OccGridMapUtil.h:64-104) */
void ORF(hessian_derivs)(void* h, int level, const float pose_map[3], const float* pts_level,
                         int n, float H[9], float dTr[3]);
/* a5: ScanMatcher::matchData (ScanMatcher.h:54-190) on one level */
void ORF(match_level)(void* h, int level, const float begin_world[3], const float* pts_level,
                      int n, int max_iterations, float out_pose_world[3], float cov[9]);
/* a7: MapRepMultiMap::matchData (MapRepMultiMap.h:116-132); a DataContainer is (pts, n, origo);
 * cov is in/out (untouched for n==0).  Retains the per-level scaled copies like the reference. */
void ORF(match)(void* h, const float begin_world[3], const float* pts, int n, const float origo[2],
                float out_pose_world[3], float cov[9]);
/* `batch` consecutive match() calls in one C loop (CSR offsets in points); timing helper so the
 * CPU baseline is not charged for Python call overhead.  out_pose [batch*3]. */
void ORF(match_many)(void* h, int batch, const float* begin_world, const float* pts,
                     const int* offsets, float* out_pose_world);
/* a11: MapRepMultiMap::updateByScan (MapRepMultiMap.h:134-147): level 0 from pts,
 * coarse levels from the containers retained by the last match() call */
void ORF(update_by_scan)(void* h, const float pose_world[3], const float* pts, int n,
                         const float origo[2]);
/* one level only, explicit points (OccGridMapBase.h:121-168) */
void ORF(update_by_scan_level)(void* h, int level, const float pose_world[3],
                               const float* pts_level, int n, const float origo_level[2]);
void ORF(on_map_updated)(void* h);                          /* MapRepMultiMap.h:107-114 */
/* Test-harness guard.  "ho": how many map reads so far carried a NaN coordinate -- the reference's bounds test lets NaN through
 * and it then indexes the grid with (int)NaN (OccGridMapUtil.h:295,302): undefined behaviour, a segmentation fault in practice;
 * the restatement returns zeros for such a read and counts it, so property tests can discard inputs the reference has no defined
 * result for.  "hr" (the reference itself): -1, it cannot tell. */
long ORF(undefined_reads)(void* h);

/* a12: HectorSlamProcessor::update (HectorSlamProcessor.h:71-113) */
void ORF(proc_set_thresholds)(void* h, float min_dist, float min_angle);
void ORF(proc_update)(void* h, const float* pts, int n, const float origo[2],
                      const float pose_hint_world[3], int map_without_matching);
void ORF(proc_last_pose)(void* h, float pose[3], float cov[9]);

/* a8 helpers (UtilFunctions.h:37-92) */
float ORF(normalize_angle)(float a);
int ORF(pose_difference_larger_than)(const float p1[3], const float p2[3], float dist, float ang);

/* the host libm calls of the path, as this process links them (glibc sincosf / expf through their ifunc
 * variants), for n arguments each -- the device's csrc/libm_exact.h is checked against these bit for bit.
 * out_prob = getGridProbability (GridMapLogOdds.h:163-166): odds = expf(x); odds / (odds + 1.0f). */
void ORF(libm_sincosf)(int n, const float* x, float* out_sin, float* out_cos);
void ORF(libm_expf)(int n, const float* x, float* out_exp, float* out_prob);

/* ---- rows next to the path, restated from the ROS node hector_mapping/src/HectorMappingRos.cpp */
/* f2: publishMap's cell loop (:449-468): -1 unknown, 0 if isFree (logOdds < 0), 100 if isOccupied (> 0);
 * GridMapLogOdds.h:76-84 */
void ORF(occupancy_grid)(void* h, int level, signed char* out);
/* f3: OccGridMapUtil::getLikelihoodForState (OccGridMapUtil.h:184-214, interpMapValue :233-285) for
 * `batch` map-frame states against one level-scaled scan: out_lh[b] = 1 - residual/size */
void ORF(likelihood_states)(void* h, int level, int batch, const float* states_map, const float* pts_level,
                            int n, float* out_lh);
/* f3: getResidualForState (:205-221) */
void ORF(residual_states)(void* h, int level, int batch, const float* states_map, const float* pts_level, int n,
                          float* out_residual);
/* f3: getCovarianceForPose (:106-160) then getCovMatrixWorldCoords (:162-188) per map-frame pose; 9 floats
 * column major each, 7 likelihoods in sigma-point order */
void ORF(covariance_for_poses)(void* h, int level, int batch, const float* poses_map, const float* pts_level, int n,
                               float* out_cov_map, float* out_cov_world, float* out_lh7);
/* f4: hectormaptools::DistanceMeasurementProvider::getDist (hector_map_tools/include/hector_map_tools/
 * HectorMapTools.h:133-234) on an int8 occupancy grid with OccupancyGrid metadata (origin, resolution):
 * out_dist[i] = resolution * cells to the first occupied (== 100) cell on the Bresenham line, or
 * resolution * -1 when none within min(5000, |major|) steps / an end point is outside; out_hit = world
 * coordinates of the hit cell (written only when there is a hit). */
void ORF(ray_distances)(const signed char* grid, int sx, int sy, float origin_x, float origin_y, float resolution,
                        int n, const float* begin_world, const float* end_world, float* out_dist, float* out_hit);
/* f1: rosLaserScanToDataContainer (:483-507): fp32 running angle, range gate (range_min, range_max - 0.1f),
 * float cos/sin; returns the number of endpoints written to out_pts (capacity n). */
int ORF(laser_scan_to_container)(const float* ranges, int n, float angle_min, float angle_increment,
                                 float range_min, float range_max, float scale_to_map, float* out_pts);

/* f1b: rosPointCloudToDataContainer (:509-542): Point32 cloud (n x {x,y,z} floats) + the laser->base
 * tf::Transform given as 12 doubles, row major [R | t] (tfScalar = double; tf::Transform::operator()
 * = row.dot(v) + origin, dot = x*x' + y*y' + z*z' left to right).  Gates: float dist_sqr in
 * (sqr_min, sqr_max); x < 0 && dist_sqr < 0.5 skipped; float(z_base - t_z) in (z_min, z_max).
 * out_origo = Vector2f(t_x, t_y) * scale_to_map.  Returns the number of endpoints. */
int ORF(point_cloud_to_container)(const float* pts_xyz, int n, const double tf_rows[12], float sqr_min, float sqr_max,
                                  float z_min, float z_max, float scale_to_map, float* out_pts, float out_origo[2]);
/* step before f1b in the node's default configuration (use_tf_scan_transformation = true, :273):
 * laser_geometry::LaserProjection::projectLaser(scan, cloud, range_cutoff) -- THIRD PARTY, absent from
 * the reference tree (package.xml: laser_geometry, unpinned; Noetic ships 1.6.7).  Restated from its
 * published algorithm: double unit vectors cos/sin(angle_min + (double)i * angle_increment), point =
 * float((double)range * unit), kept when range < range_cutoff (double compare; cutoff < 0 => range_max)
 * and range >= range_min, z = 0.  Parity for THIS step is pinned only to the restatement. */
int ORF(project_laser)(const float* ranges, int n, float angle_min, float angle_increment, float range_min,
                       float range_max, double range_cutoff, float* out_xyz);

#ifdef __cplusplus
}
#endif
#endif
