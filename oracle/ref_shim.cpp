// TEST INFRASTRUCTURE ONLY -- the reference itself behind the oracle C API.
//
// This translation unit #includes the UNMODIFIED hector_slam_lib headers
// straight from /root/reference (via -I, nothing is copied into the repo) and
// instantiates the reference's own classes: HectorSlamProcessor ->
// MapRepMultiMap -> MapProcContainer{GridMap, OccGridMapUtilConfig, ScanMatcher}.
// Eigen3 and tf (third-party, absent from /root/reference and from this image)
// are provided by the private stand-ins in oracle/stubs/.  Built by
// oracle/Makefile into oracle/_ref/libhector_ref.so (git-ignored, travels to the
// GPU box with the snapshot).  Used to pin oracle/hector_oracle.cpp bit-for-bit
// and as the "reference" CPU baseline in bench.py.
#include "slam_main/HectorSlamProcessor.h"
#include "hector_map_tools/HectorMapTools.h"  // f4: -I $(REFROOT)/hector_map_tools/include, nav_msgs from oracle/stubs

#include <mutex>
#include <sstream>
#include <vector>

#define ORACLE_PREFIX hr_
#include "oracle_api.h"

namespace {


typedef hectorslam::GridMap RefGridMap;
using hectorslam::HectorSlamProcessor;
using hectorslam::MapProcContainer;
using hectorslam::MapRepMultiMap;
using hectorslam::MapRepresentationInterface;

// reach protected members without touching the reference sources
struct ProcAccess : HectorSlamProcessor {
  static MapRepresentationInterface* get(HectorSlamProcessor& p) { return p.*(&ProcAccess::mapRep); }
};
struct MapAccess : MapRepMultiMap {
  static std::vector<MapProcContainer>& get(MapRepMultiMap& m) { return m.*(&MapAccess::mapContainer); }
};

struct Ref {
  HectorSlamProcessor* proc;
  MapRepMultiMap* map;
  Ref() : proc(0), map(0) {}
  MapProcContainer& level(int l) { return MapAccess::get(*map)[l]; }
};

// the reference prints a banner (MapRepMultiMap.h:60) and a clamp message
// (ScanMatcher.h:211,214) on std::cout; keep test output clean
// Process-wide and reference counted: the bench's all-cores leg calls in from many threads, and
// per-call save/restore of std::cout's buffer would leave it pointing at another thread's dead sink.
struct CoutMute {
  struct NullBuf : std::streambuf {
    int overflow(int c) override { return c; }
  };
  static std::mutex& mu() { static std::mutex m; return m; }
  static int& depth() { static int d = 0; return d; }
  static std::streambuf*& saved() { static std::streambuf* s = nullptr; return s; }
  CoutMute() {
    std::lock_guard<std::mutex> lk(mu());
    if (depth()++ == 0) {
      static NullBuf* nb = new NullBuf;  // never destroyed: safe at any point of process exit
      saved() = std::cout.rdbuf(nb);
    }
  }
  ~CoutMute() {
    std::lock_guard<std::mutex> lk(mu());
    if (--depth() == 0) std::cout.rdbuf(saved());
  }
};

static hectorslam::DataContainer make_container(const float* pts, int n, const float origo[2]) {
  hectorslam::DataContainer dc(n > 0 ? n : 1);
  for (int i = 0; i < n; ++i) dc.add(Eigen::Vector2f(pts[2 * i], pts[2 * i + 1]));
  dc.setOrigo(origo ? Eigen::Vector2f(origo[0], origo[1]) : Eigen::Vector2f(0.0f, 0.0f));
  return dc;
}
static inline Eigen::Vector3f v3(const float p[3]) { return Eigen::Vector3f(p[0], p[1], p[2]); }
static inline void store3(const Eigen::Vector3f& v, float out[3]) {
  out[0] = v[0];
  out[1] = v[1];
  out[2] = v[2];
}
static inline Eigen::Matrix3f m3(const float c[9]) {
  Eigen::Matrix3f m;
  for (int i = 0; i < 9; ++i) m.data()[i] = c[i];
  return m;
}
static inline void store9(const Eigen::Matrix3f& m, float out[9]) {
  for (int i = 0; i < 9; ++i) out[i] = m.data()[i];
}

}  // namespace

extern "C" {

void* hr_create(float res, int sx, int sy, unsigned levels, float start_x, float start_y) {
  CoutMute mute;
  Ref* r = new Ref();
  r->proc = new HectorSlamProcessor(res, sx, sy, Eigen::Vector2f(start_x, start_y), (int)levels, 0, 0);
  r->map = static_cast<MapRepMultiMap*>(ProcAccess::get(*r->proc));
  return r;
}
void hr_destroy(void* h) {
  Ref* r = (Ref*)h;
  delete r->proc;
  delete r;
}
void hr_reset(void* h) { ((Ref*)h)->proc->reset(); }
int hr_levels(void* h) { return ((Ref*)h)->proc->getMapLevels(); }
float hr_scale_to_map(void* h) { return ((Ref*)h)->proc->getScaleToMap(); }
void hr_set_update_factor_free(void* h, float f) { ((Ref*)h)->proc->setUpdateFactorFree(f); }
void hr_set_update_factor_occupied(void* h, float f) { ((Ref*)h)->proc->setUpdateFactorOccupied(f); }
void hr_level_info(void* h, int level, int* sx, int* sy, float* cell, float* scale) {
  const RefGridMap& g = ((Ref*)h)->proc->getGridMap(level);
  *sx = g.getSizeX();
  *sy = g.getSizeY();
  *cell = g.getCellLength();
  *scale = g.getScaleToMap();
}
void hr_download_level(void* h, int level, float* lo, int* ui) {
  const RefGridMap& g = ((Ref*)h)->proc->getGridMap(level);
  const int n = g.getSizeX() * g.getSizeY();
  for (int i = 0; i < n; ++i) {
    if (lo) lo[i] = g.getCell(i).logOddsVal;
    if (ui) ui[i] = g.getCell(i).updateIndex;
  }
}
void hr_upload_level(void* h, int level, const float* lo, const int* ui) {
  Ref* r = (Ref*)h;
  RefGridMap& g = r->level(level).getGridMap();
  const int n = g.getSizeX() * g.getSizeY();
  for (int i = 0; i < n; ++i) {
    if (lo) g.getCell(i).logOddsVal = lo[i];
    if (ui) g.getCell(i).updateIndex = ui[i];
  }
  r->level(level).resetCachedData();
}
void hr_map_coords_pose(void* h, int level, const float w[3], float m[3]) {
  store3(((Ref*)h)->proc->getGridMap(level).getMapCoordsPose(v3(w)), m);
}
void hr_world_coords_pose(void* h, int level, const float m[3], float w[3]) {
  store3(((Ref*)h)->proc->getGridMap(level).getWorldCoordsPose(v3(m)), w);
}
void hr_interp(void* h, int level, const float* xy, int n, float* out) {
  Ref* r = (Ref*)h;
  for (int i = 0; i < n; ++i) {
    Eigen::Vector3f v = r->level(level).gridMapUtil->interpMapValueWithDerivatives(
        Eigen::Vector2f(xy[2 * i], xy[2 * i + 1]));
    store3(v, out + 3 * i);
  }
}
void hr_hessian_derivs(void* h, int level, const float pose[3], const float* pts, int n, float H[9],
                       float dTr[3]) {
  Ref* r = (Ref*)h;
  hectorslam::DataContainer dc = make_container(pts, n, 0);
  Eigen::Matrix3f Hm;
  Eigen::Vector3f d;
  r->level(level).gridMapUtil->getCompleteHessianDerivs(v3(pose), dc, Hm, d);
  store9(Hm, H);
  store3(d, dTr);
}
void hr_match_level(void* h, int level, const float begin[3], const float* pts, int n, int maxIter,
                    float out[3], float cov[9]) {
  CoutMute mute;
  Ref* r = (Ref*)h;
  hectorslam::DataContainer dc = make_container(pts, n, 0);
  Eigen::Matrix3f c = m3(cov);
  store3(r->level(level).matchData(v3(begin), dc, c, maxIter), out);
  store9(c, cov);
}
void hr_match(void* h, const float begin[3], const float* pts, int n, const float origo[2],
              float out[3], float cov[9]) {
  CoutMute mute;
  Ref* r = (Ref*)h;
  hectorslam::DataContainer dc = make_container(pts, n, origo);
  Eigen::Matrix3f c = m3(cov);
  store3(r->map->matchData(v3(begin), dc, c), out);
  store9(c, cov);
}
void hr_match_many(void* h, int batch, const float* begin, const float* pts, const int* offs, float* out) {
  CoutMute mute;
  Ref* r = (Ref*)h;
  const float zero[2] = {0.0f, 0.0f};
  Eigen::Matrix3f c = Eigen::Matrix3f::Zero();
  for (int b = 0; b < batch; ++b) {
    // container construction is part of what the ROS node does per scan too (HectorMappingRos.cpp:483-507)
    hectorslam::DataContainer dc = make_container(pts + 2 * (size_t)offs[b], offs[b + 1] - offs[b], zero);
    store3(r->map->matchData(v3(begin + 3 * b), dc, c), out + 3 * b);
  }
}
void hr_update_by_scan(void* h, const float pose[3], const float* pts, int n, const float origo[2]) {
  Ref* r = (Ref*)h;
  hectorslam::DataContainer dc = make_container(pts, n, origo);
  r->map->updateByScan(dc, v3(pose));
}
void hr_update_by_scan_level(void* h, int level, const float pose[3], const float* pts, int n,
                             const float origo[2]) {
  Ref* r = (Ref*)h;
  hectorslam::DataContainer dc = make_container(pts, n, origo);
  r->level(level).updateByScan(dc, v3(pose));
}
void hr_on_map_updated(void* h) { ((Ref*)h)->map->onMapUpdated(); }
long hr_undefined_reads(void*) { return -1; }  // (the reference cannot tell: it crashes on such a read, oracle_api.h)

void hr_proc_set_thresholds(void* h, float d, float a) {
  ((Ref*)h)->proc->setMapUpdateMinDistDiff(d);
  ((Ref*)h)->proc->setMapUpdateMinAngleDiff(a);
}
void hr_proc_update(void* h, const float* pts, int n, const float origo[2], const float hint[3],
                    int mapWithoutMatching) {
  CoutMute mute;
  Ref* r = (Ref*)h;
  hectorslam::DataContainer dc = make_container(pts, n, origo);
  r->proc->update(dc, v3(hint), mapWithoutMatching != 0);
}
void hr_proc_last_pose(void* h, float pose[3], float cov[9]) {
  Ref* r = (Ref*)h;
  store3(r->proc->getLastScanMatchPose(), pose);
  store9(r->proc->getLastScanMatchCovariance(), cov);
}
float hr_normalize_angle(float a) { return util::normalize_angle(a); }
int hr_pose_difference_larger_than(const float p1[3], const float p2[3], float d, float a) {
  return util::poseDifferenceLargerThan(v3(p1), v3(p2), d, a) ? 1 : 0;
}

// f3 through the reference's own OccGridMapUtil::getLikelihoodForState
void hr_likelihood_states(void* h, int level, int batch, const float* states, const float* pts, int n, float* out) {
  Ref* r = (Ref*)h;
  hectorslam::DataContainer dc = make_container(pts, n, 0);
  for (int b = 0; b < batch; ++b) out[b] = r->level(level).gridMapUtil->getLikelihoodForState(v3(states + 3 * b), dc);
}
void hr_residual_states(void* h, int level, int batch, const float* states, const float* pts, int n, float* out) {
  Ref* r = (Ref*)h;
  hectorslam::DataContainer dc = make_container(pts, n, 0);
  for (int b = 0; b < batch; ++b) out[b] = r->level(level).gridMapUtil->getResidualForState(v3(states + 3 * b), dc);
}
// the reference's own getCovarianceForPose / getCovMatrixWorldCoords; its per-call print (:136) goes to a sink
void hr_covariance_for_poses(void* h, int level, int batch, const float* poses, const float* pts, int n, float* out_map,
                             float* out_world, float* out_lh7) {
  Ref* r = (Ref*)h;
  hectorslam::DataContainer dc = make_container(pts, n, 0);
  CoutMute mute;
  for (int b = 0; b < batch; ++b) {
    auto* util = r->level(level).gridMapUtil;
    const Eigen::Vector3f p = v3(poses + 3 * b);
    const Eigen::Matrix3f cm = util->getCovarianceForPose(p, dc);
    const Eigen::Matrix3f cw = util->getCovMatrixWorldCoords(cm);
    if (out_map) memcpy(out_map + 9 * b, cm.data(), 9 * sizeof(float));
    if (out_world) memcpy(out_world + 9 * b, cw.data(), 9 * sizeof(float));
    if (out_lh7) {
      const float x = p[0], y = p[1], a = p[2];
      const float sp[7][3] = {{x + 1.5f, y, a}, {x - 1.5f, y, a}, {x, y + 1.5f, a}, {x, y - 1.5f, a},
                              {x, y, a + 0.05f}, {x, y, a - 0.05f}, {x, y, a}};
      for (int i = 0; i < 7; ++i) out_lh7[7 * b + i] = util->getLikelihoodForState(v3(sp[i]), dc);
    }
  }
}
// f4 through the reference's own code: hectormaptools' DistanceMeasurementProvider::getDist, the UNMODIFIED
// hector_map_tools/include/hector_map_tools/HectorMapTools.h:132-234 compiled against a stand-in for the two
// generated message headers it includes (oracle/stubs/nav_msgs/).  One provider per call; the grid is copied into
// the message's data vector exactly as hector_map_server receives it.
void hr_ray_distances(const signed char* grid, int sx, int sy, float ox, float oy, float res, int n, const float* bw,
                      const float* ew, float* out_dist, float* out_hit) {
  std::shared_ptr<nav_msgs::OccupancyGrid> map(new nav_msgs::OccupancyGrid());
  map->info.resolution = res;
  map->info.width = (uint32_t)sx;
  map->info.height = (uint32_t)sy;
  map->info.origin.position.x = ox;
  map->info.origin.position.y = oy;
  map->data.assign(grid, grid + (size_t)sx * sy);
  HectorMapTools::DistanceMeasurementProvider dmp;
  dmp.setMap(map);
  for (int r = 0; r < n; ++r) {
    // getDist writes hitCoords from an UNINITIALISED end_point_map when there is no hit (HectorMapTools.h:146-156):
    // only the distance says whether the hit is meaningful
    Eigen::Vector2f hit(0.0f, 0.0f);
    const float d = dmp.getDist(Eigen::Vector2f(bw[2 * r], bw[2 * r + 1]), Eigen::Vector2f(ew[2 * r], ew[2 * r + 1]), &hit);
    out_dist[r] = d;
    if (d >= 0.0f) {
      out_hit[2 * r] = hit[0];
      out_hit[2 * r + 1] = hit[1];
    }
  }
}
// f2 through the reference's own GridMap::isFree / isOccupied (GridMapLogOdds.h:76-84)
void hr_occupancy_grid(void* h, int level, signed char* out) {
  const RefGridMap& g = ((Ref*)h)->proc->getGridMap(level);
  const int size = g.getSizeX() * g.getSizeY();
  for (int i = 0; i < size; ++i) out[i] = g.isFree(i) ? 0 : (g.isOccupied(i) ? 100 : -1);
}
// f1 lives in the ROS node (needs sensor_msgs); only the restatement exists -- forwarded so both
// libraries export the same symbols
int hr_laser_scan_to_container(const float* r, int n, float a0, float inc, float rmin, float rmax, float s, float* out) {
  float angle = a0;
  int m = 0;
  const float maxRangeForContainer = rmax - 0.1f;
  for (int i = 0; i < n; ++i) {
    float dist = r[i];
    if ((dist > rmin) && (dist < maxRangeForContainer)) {
      dist *= s;
      out[2 * m] = cos(angle) * dist;  // unqualified, float argument: resolves as in the node's TU
      out[2 * m + 1] = sin(angle) * dist;
      ++m;
    }
    angle += inc;
  }
  return m;
}

// f1b and the laser_geometry projection: node / third-party code, restatement only (same symbols in both libraries)
int hr_point_cloud_to_container(const float* pts, int n, const double T[12], float sqr_min, float sqr_max, float z_min,
                                float z_max, float scaleToMap, float* out_pts, float out_origo[2]) {
  // HectorMappingRos.cpp:509-542.  laserPos = laserTransform.getOrigin() (doubles)
  const double lx = T[3], ly = T[7], lz = T[11];
  out_origo[0] = (float)lx * scaleToMap;  // Eigen::Vector2f(laserPos.x(), laserPos.y()) * scaleToMap  (:517)
  out_origo[1] = (float)ly * scaleToMap;
  int m = 0;
  for (int i = 0; i < n; ++i) {
    const float px = pts[3 * i], py = pts[3 * i + 1], pz = pts[3 * i + 2];
    const float dist_sqr = px * px + py * py;  // :524
    if ((dist_sqr > sqr_min) && (dist_sqr < sqr_max)) {
      if ((px < 0.0f) && (dist_sqr < 0.50f)) continue;  // :528
      // tf::Transform * tf::Vector3: m_basis[r].dot(v) + m_origin[r], all double
      const double vx = px, vy = py, vz = pz;
      const double bx = (T[0] * vx + T[1] * vy + T[2] * vz) + lx;
      const double by = (T[4] * vx + T[5] * vy + T[6] * vz) + ly;
      const double bz = (T[8] * vx + T[9] * vy + T[10] * vz) + lz;
      const float pointPosLaserFrameZ = (float)(bz - lz);  // :534
      if (pointPosLaserFrameZ > z_min && pointPosLaserFrameZ < z_max) {
        out_pts[2 * m] = (float)bx * scaleToMap;  // Eigen::Vector2f(x, y) * scaleToMap  (:538)
        out_pts[2 * m + 1] = (float)by * scaleToMap;
        ++m;
      }
    }
  }
  return m;
}

int hr_project_laser(const float* ranges, int n, float angle_min, float angle_increment, float range_min,
                     float range_max, double range_cutoff, float* out_xyz) {
  // laser_geometry 1.6.x LaserProjection::projectLaser_ / getUnitVectors_ (third party, restated)
  if (range_cutoff < 0) range_cutoff = range_max;
  const double a0 = angle_min, inc = angle_increment;
  int count = 0;
  for (int i = 0; i < n; ++i) {
    const double r = (double)ranges[i];
    const double ox = r * cos(a0 + (double)i * inc);
    const double oy = r * sin(a0 + (double)i * inc);
    const float range = (float)r;
    if ((range < range_cutoff) && (range >= range_min)) {
      out_xyz[3 * count] = (float)ox;
      out_xyz[3 * count + 1] = (float)oy;
      out_xyz[3 * count + 2] = 0.0f;
      ++count;
    }
  }
  return count;
}

void hr_libm_sincosf(int n, const float* x, float* out_sin, float* out_cos) {
  for (int i = 0; i < n; ++i) {
    sincosf(x[i], &out_sin[i], &out_cos[i]);  // what GCC makes of the reference's sin(pose[2]), cos(pose[2]) pair
  }
}
void hr_libm_expf(int n, const float* x, float* out_exp, float* out_prob) {
  for (int i = 0; i < n; ++i) {
    const float odds = expf(x[i]);
    out_exp[i] = odds;
    out_prob[i] = odds / (odds + 1.0f);
  }
}


}  // extern "C"
