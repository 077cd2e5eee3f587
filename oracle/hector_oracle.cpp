// TEST INFRASTRUCTURE ONLY -- CPU oracle for the hector_mapping scan-match path.
//
// A plain-C++ (no Eigen, no ROS) restatement of the reference algorithm rows
// a1-a12 of SURVEY.md section 8, single threaded, IEEE fp32 with the reference's
// evaluation order.  Build: g++ -O2 -ffp-contract=off (oracle/Makefile).
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
// it.  The product library never links, includes or calls anything in oracle/.
//
// PINNING: the reference ships no tests, golden vectors or fixtures for this
// path, so this file is pinned against the reference's own
// code instead: oracle/ref_shim.cpp compiles the UNMODIFIED headers from
// /root/reference through a private Eigen/tf stand-in (oracle/stubs/) into
// oracle/_ref/libhector_ref.so, and tests/test_oracle_vs_reference.py asserts
// bit-for-bit equality of every entry point below on seeded scenes.  The hector
// code is therefore pinned; the arithmetic that lives inside Eigen3 (a
// third-party dependency absent from /root/reference, version not pinned by the
// reference; 3.3.x semantics restated in oracle/stubs/Eigen/mini_eigen.h and
// again here) is restated from its published sources, not executed.
//
// File:line citations are relative to
//   /root/reference/hector_mapping/include/hector_slam_lib/   (abbrev. HSL/)
#include <math.h>   // first: the ROS build sees tf's <math.h>, so the reference's
#include <stdlib.h> // unqualified sin/cos/exp/log/abs on floats are the float overloads
#include <float.h>
#include <limits.h>
#include <string.h>
#include <vector>

#define ORACLE_PREFIX ho_
#include "oracle_api.h"

namespace {

// Eigen Affine2f = 2x2 linear (column-major) + translation.
struct Affine2 {
  float l00, l10, l01, l11;
  float t0, t1;
};

// Eigen Transform<Affine> * Vector2f: res = t; res += linear * v, where the 2-term
// coefficient product is (l(i,0)*v0 + l(i,1)*v1)   (mini_eigen.h header, Transform.h)
static inline void affine_apply(const Affine2& a, float vx, float vy, float& ox, float& oy) {
  ox = a.t0 + (a.l00 * vx + a.l01 * vy);
  oy = a.t1 + (a.l10 * vx + a.l11 * vy);
}

// Translation2f(x,y) * Rotation2Df(theta)  (HSL/map/OccGridMapUtil.h:349-352,
// HSL/map/OccGridMapBase.h:130-131): linear = [c -s; s c] from float sin/cos, t = (x,y).
static inline Affine2 pose_transform(float x, float y, float theta) {
  const float sinA = sinf(theta);
  const float cosA = cosf(theta);
  Affine2 a;
  a.l00 = cosA;
  a.l01 = -sinA;
  a.l10 = sinA;
  a.l11 = cosA;
  a.t0 = 0.0f;
  a.t1 = 0.0f;
  a.t0 += x;
  a.t1 += y;
  return a;
}

struct Level {
  int sx, sy;
  float cellLength;
  float scaleToMap;
  float limx, limy;  // MapDimensionProperties::mapLimitsf = dims - 2.0f  (HSL/map/MapDimensionProperties.h:70-74)
  Affine2 mapTworld, worldTmap;
  std::vector<float> logOdds;    // LogOddsCell::logOddsVal   (HSL/map/GridMapLogOdds.h:99)
  std::vector<int> updateIndex;  // LogOddsCell::updateIndex  (HSL/map/GridMapLogOdds.h:100)
  // GridMapCacheArray (HSL/map/GridMapCacheArray.h:34-39,150-155)
  std::vector<float> cacheVal;
  std::vector<int> cacheIdx;
  int currCacheIndex;
  // GridMapLogOddsFunctions (HSL/map/GridMapLogOdds.h:200-203)
  float logOddsOccupied, logOddsFree;
  // OccGridMapBase counters (HSL/map/OccGridMapBase.h:264-266), GridMapBase::lastUpdateIndex (:390)
  int currUpdateIndex, currMarkOccIndex, currMarkFreeIndex, lastUpdateIndex;
  // ScanMatcher members (HSL/matcher/ScanMatcher.h:242-243); H column-major
  float H[9], dTr[3];
  // TEST-HARNESS GUARD, not reference behaviour: reads with a NaN coordinate.  The reference's bounds test lets NaN through
  // (every comparison is false), then indexes the grid with (int)NaN (OccGridMapUtil.h:295,302) -- undefined behaviour, a
  // segmentation fault in practice, reached whenever its own Gauss-Newton step divides by a zero determinant.  The restatement
  // returns zeros for such a read and counts it, so that property tests can discard inputs on which the reference has no
  // defined result instead of dying with it (ho_undefined_reads).
  long undefinedReads = 0;
};

static inline float prob_to_log_odds(float prob) {  // GridMapLogOdds.h:196-200
  float odds = prob / (1.0f - prob);
  return logf(odds);
}

// GridMapBase::setMapTransformation (HSL/map/GridMapBase.h:265-280)
static void set_map_transformation(Level& L, float offx, float offy, float cellLength) {
  L.cellLength = cellLength;
  L.scaleToMap = 1.0f / cellLength;
  const float s = L.scaleToMap;
  // AlignedScaling2f(s,s) * Translation2f(off): linear = diag, t = diag * off
  Affine2 m;
  m.l00 = s;
  m.l10 = 0.0f;
  m.l01 = 0.0f;
  m.l11 = s;
  m.t0 = s * offx;
  m.t1 = s * offy;
  L.mapTworld = m;
  // Transform::inverse(): 2x2 inverse by cofactors * invdet, then t' = (-inv) * t
  const float det = m.l00 * m.l11 - m.l10 * m.l01;
  const float invdet = 1.0f / det;
  Affine2 w;
  w.l00 = m.l11 * invdet;
  w.l10 = -m.l10 * invdet;
  w.l01 = -m.l01 * invdet;
  w.l11 = m.l00 * invdet;
  w.t0 = (-w.l00) * m.t0 + (-w.l01) * m.t1;
  w.t1 = (-w.l10) * m.t0 + (-w.l11) * m.t1;
  L.worldTmap = w;
}

static void level_clear(Level& L) {  // GridMapBase::clear (GridMapBase.h:77-88) + LogOddsCell::resetGridCell (:89-93)
  const size_t n = (size_t)L.sx * L.sy;
  for (size_t i = 0; i < n; ++i) {
    L.logOdds[i] = 0.0f;
    L.updateIndex[i] = -1;
  }
}

static void level_init(Level& L, float cellLength, int sx, int sy, float offx, float offy) {
  L.sx = sx;
  L.sy = sy;
  const size_t n = (size_t)sx * sy;
  L.logOdds.assign(n, 0.0f);
  L.updateIndex.assign(n, -1);
  L.limx = (float)sx - 2.0f;
  L.limy = (float)sy - 2.0f;
  set_map_transformation(L, offx, offy, cellLength);
  L.logOddsFree = prob_to_log_odds(0.4f);      // GridMapLogOdds.h:117
  L.logOddsOccupied = prob_to_log_odds(0.6f);  // GridMapLogOdds.h:118
  L.currUpdateIndex = 0;
  L.currMarkOccIndex = -1;
  L.currMarkFreeIndex = -1;
  L.lastUpdateIndex = -1;
  L.cacheVal.assign(n, 0.0f);
  L.cacheIdx.assign(n, -1);  // GridMapCacheArray.h:128-132
  L.currCacheIndex = 0;
  for (int i = 0; i < 9; ++i) L.H[i] = 0.0f;
  for (int i = 0; i < 3; ++i) L.dTr[i] = 0.0f;
}

// GridMapLogOddsFunctions::getGridProbability (GridMapLogOdds.h:163-166)
static inline float grid_probability(float logOddsVal) {
  float odds = expf(logOddsVal);
  return odds / (odds + 1.0f);
}

// cache lookup-or-fill, OccGridMapUtil.h:306-309 + GridMapCacheArray.h:80-102
static inline float cached_prob(Level& L, int index) {
  if (L.cacheIdx[index] == L.currCacheIndex) {
    return L.cacheVal[index];
  }
  const float v = grid_probability(L.logOdds[index]);
  L.cacheIdx[index] = L.currCacheIndex;
  L.cacheVal[index] = v;
  return v;
}

// a1: OccGridMapUtil::interpMapValueWithDerivatives (OccGridMapUtil.h:287-347)
static inline void interp_with_derivs(Level& L, float cx, float cy, float& M, float& gx, float& gy) {
  if (cx != cx || cy != cy) {  // (test-harness guard, see Level::undefinedReads)
    ++L.undefinedReads;
    M = gx = gy = 0.0f;
    return;
  }
  // MapDimensionProperties::pointOutOfMapBounds (MapDimensionProperties.h:65-68)
  if ((cx < 0.0f) || (cx > L.limx) || (cy < 0.0f) || (cy > L.limy)) {
    M = 0.0f;
    gx = 0.0f;
    gy = 0.0f;
    return;
  }
  const int ix = (int)cx;  // cast<int>() truncation (:295)
  const int iy = (int)cy;
  const float fx = cx - (float)ix;  // :298
  const float fy = cy - (float)iy;
  const int sizeX = L.sx;
  int index = iy * sizeX + ix;  // :302
  const float i0 = cached_prob(L, index);
  ++index;
  const float i1 = cached_prob(L, index);
  index += sizeX - 1;
  const float i2 = cached_prob(L, index);
  ++index;
  const float i3 = cached_prob(L, index);
  const float dx1 = i0 - i1;  // :332-336
  const float dx2 = i2 - i3;
  const float dy1 = i0 - i2;
  const float dy2 = i1 - i3;
  const float xFacInv = (1.0f - fx);  // :338-339
  const float yFacInv = (1.0f - fy);
  // :341-346 -- the source blends dx with the X fractions and dy with the Y
  // fractions (not the analytic bilinear derivative); reproduced on purpose.
  M = ((i0 * xFacInv + i1 * fx) * (yFacInv)) + ((i2 * xFacInv + i3 * fx) * (fy));
  gx = -((dx1 * xFacInv) + (dx2 * fx));
  gy = -((dy1 * yFacInv) + (dy2 * fy));
}

// a2: OccGridMapUtil::getCompleteHessianDerivs (OccGridMapUtil.h:64-104)
static void complete_hessian_derivs(Level& L, const float pose[3], const float* pts, int n,
                                    float H[9], float dTr[3]) {
  const Affine2 transform = pose_transform(pose[0], pose[1], pose[2]);  // :68
  const float sinRot = sinf(pose[2]);  // :70-71
  const float cosRot = cosf(pose[2]);
  float h00 = 0.0f, h11 = 0.0f, h22 = 0.0f, h01 = 0.0f, h02 = 0.0f, h12 = 0.0f;
  float d0 = 0.0f, d1 = 0.0f, d2 = 0.0f;
  for (int i = 0; i < n; ++i) {  // :76-98, beam order
    const float px = pts[2 * i], py = pts[2 * i + 1];
    float tx, ty;
    affine_apply(transform, px, py, tx, ty);
    float M, gx, gy;
    interp_with_derivs(L, tx, ty, M, gx, gy);
    const float funVal = 1.0f - M;
    d0 += gx * funVal;
    d1 += gy * funVal;
    const float rotDeriv = ((-sinRot * px - cosRot * py) * gx + (cosRot * px - sinRot * py) * gy);  // :87
    d2 += rotDeriv * funVal;
    h00 += gx * gx;
    h11 += gy * gy;
    h22 += rotDeriv * rotDeriv;
    h01 += gx * gy;
    h02 += gx * rotDeriv;
    h12 += gy * rotDeriv;
  }
  // column-major 3x3, mirrored (:100-102)
  H[0] = h00; H[4] = h11; H[8] = h22;
  H[3] = h01; H[1] = h01;
  H[6] = h02; H[2] = h02;
  H[7] = h12; H[5] = h12;
  dTr[0] = d0; dTr[1] = d1; dTr[2] = d2;
}

#define HM(r, c) H[(c) * 3 + (r)]
// Eigen Matrix3f::inverse() * Vector3f  (ScanMatcher.h:205): cofactor inverse
// (LU/InverseImpl.h) then coefficient-based product; 3-term sums are x0 + (x1 + x2).
static inline float cof3(const float* H, int i, int j) {
  const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
  return HM(i1, j1) * HM(i2, j2) - HM(i1, j2) * HM(i2, j1);
}
static void inverse3_times(const float* H, const float v[3], float out[3]) {
  const float c00 = cof3(H, 0, 0), c10 = cof3(H, 1, 0), c20 = cof3(H, 2, 0);
  const float det = c00 * HM(0, 0) + (c10 * HM(1, 0) + c20 * HM(2, 0));
  const float invdet = 1.0f / det;
  float inv[9];
#define INV(r, c) inv[(c) * 3 + (r)]
  INV(0, 0) = c00 * invdet;
  INV(0, 1) = c10 * invdet;
  INV(0, 2) = c20 * invdet;
  INV(1, 0) = cof3(H, 0, 1) * invdet;
  INV(1, 1) = cof3(H, 1, 1) * invdet;
  INV(1, 2) = cof3(H, 2, 1) * invdet;
  INV(2, 0) = cof3(H, 0, 2) * invdet;
  INV(2, 1) = cof3(H, 1, 2) * invdet;
  INV(2, 2) = cof3(H, 2, 2) * invdet;
  for (int i = 0; i < 3; ++i) out[i] = INV(i, 0) * v[0] + (INV(i, 1) * v[1] + INV(i, 2) * v[2]);
#undef INV
}

// a4: ScanMatcher::estimateTransformationLogLh (ScanMatcher.h:194-221)
static bool estimate_transformation_log_lh(Level& L, float estimate[3], const float* pts, int n) {
  float* H = L.H;
  complete_hessian_derivs(L, estimate, pts, n, H, L.dTr);
  if ((HM(0, 0) != 0.0f) && (HM(1, 1) != 0.0f)) {  // :201
    float searchDir[3];
    inverse3_times(H, L.dTr, searchDir);  // :205
    if (searchDir[2] > 0.2f) {            // :209-215 (the std::cout message is not restated)
      searchDir[2] = 0.2f;
    } else if (searchDir[2] < -0.2f) {
      searchDir[2] = -0.2f;
    }
    estimate[0] += searchDir[0];  // :217, :223-226
    estimate[1] += searchDir[1];
    estimate[2] += searchDir[2];
    return true;
  }
  return false;
}
#undef HM

// a8: util::normalize_angle (HSL/util/UtilFunctions.h:37-49) -- double fmod, float result
static inline float normalize_angle_pos(float angle) {
  return (float)fmod(fmod((double)angle, 2.0f * M_PI) + 2.0f * M_PI, 2.0f * M_PI);
}
static inline float normalize_angle(float angle) {
  float a = normalize_angle_pos(angle);
  if (a > M_PI) {
    a -= 2.0f * M_PI;
  }
  return a;
}
static inline int util_sign(int x) { return x > 0 ? 1 : -1; }  // UtilFunctions.h:56-59

static inline void map_coords_pose(const Level& L, const float w[3], float m[3]) {  // GridMapBase.h:235-239
  affine_apply(L.mapTworld, w[0], w[1], m[0], m[1]);
  m[2] = w[2];
}
static inline void world_coords_pose(const Level& L, const float m[3], float w[3]) {  // GridMapBase.h:226-230
  affine_apply(L.worldTmap, m[0], m[1], w[0], w[1]);
  w[2] = m[2];
}

// a5: ScanMatcher::matchData (ScanMatcher.h:54-190), draw/debug hooks null
static void scan_matcher_match_data(Level& L, const float beginWorld[3], const float* pts, int n,
                                    int maxIterations, float outWorld[3], float cov[9]) {
  if (n != 0) {  // :68
    float estimate[3];
    map_coords_pose(L, beginWorld, estimate);            // :70-72
    estimate_transformation_log_lh(L, estimate, pts, n);  // :74
    for (int i = 0; i < maxIterations; ++i) {             // :91-110
      estimate_transformation_log_lh(L, estimate, pts, n);
    }
    estimate[2] = normalize_angle(estimate[2]);  // :170
    for (int i = 0; i < 9; ++i) cov[i] = L.H[i];  // :184
    world_coords_pose(L, estimate, outWorld);     // :186
    return;
  }
  outWorld[0] = beginWorld[0];  // :189
  outWorld[1] = beginWorld[1];
  outWorld[2] = beginWorld[2];
}

// a11: OccGridMapBase::bresenhamCellFree / bresenhamCellOcc (OccGridMapBase.h:216-241)
static inline void cell_free(Level& L, unsigned int offset) {
  if (L.updateIndex[offset] < L.currMarkFreeIndex) {
    L.logOdds[offset] += L.logOddsFree;  // GridMapLogOdds.h:146-151
    L.updateIndex[offset] = L.currMarkFreeIndex;
  }
}
static inline void cell_occ(Level& L, unsigned int offset) {
  if (L.updateIndex[offset] < L.currMarkOccIndex) {
    if (L.updateIndex[offset] == L.currMarkFreeIndex) {
      L.logOdds[offset] -= L.logOddsFree;  // updateUnsetFree, GridMapLogOdds.h:153-156
    }
    if (L.logOdds[offset] < 50.0f) {  // updateSetOccupied, GridMapLogOdds.h:135-140
      L.logOdds[offset] += L.logOddsOccupied;
    }
    L.updateIndex[offset] = L.currMarkOccIndex;
  }
}

// OccGridMapBase::bresenham2D (OccGridMapBase.h:243-260)
static void bresenham2d(Level& L, unsigned int abs_da, unsigned int abs_db, int error_b,
                        int offset_a, int offset_b, unsigned int offset) {
  cell_free(L, offset);
  const unsigned int end = abs_da - 1;
  for (unsigned int i = 0; i < end; ++i) {
    offset += offset_a;
    error_b += abs_db;
    if ((unsigned int)error_b >= abs_da) {
      offset += offset_b;
      error_b -= abs_da;
    }
    cell_free(L, offset);
  }
}

// OccGridMapBase::updateLineBresenhami (OccGridMapBase.h:170-214)
static void update_line_bresenhami(Level& L, int x0, int y0, int x1, int y1) {
  if ((x0 < 0) || (x0 >= L.sx) || (y0 < 0) || (y0 >= L.sy)) return;
  if ((x1 < 0) || (x1 >= L.sx) || (y1 < 0) || (y1 >= L.sy)) return;
  const int dx = x1 - x0;
  const int dy = y1 - y0;
  const unsigned int abs_dx = abs(dx);
  const unsigned int abs_dy = abs(dy);
  const int offset_dx = util_sign(dx);
  const int offset_dy = util_sign(dy) * L.sx;
  const unsigned int startOffset = y0 * L.sx + x0;
  if (abs_dx >= abs_dy) {
    const int error_y = abs_dx / 2;
    bresenham2d(L, abs_dx, abs_dy, error_y, offset_dx, offset_dy, startOffset);
  } else {
    const int error_x = abs_dy / 2;
    bresenham2d(L, abs_dy, abs_dx, error_x, offset_dy, offset_dx, startOffset);
  }
  const unsigned int endOffset = y1 * L.sx + x1;
  cell_occ(L, endOffset);
}

// OccGridMapBase::updateByScan (OccGridMapBase.h:121-168)
static void level_update_by_scan(Level& L, const float* pts, int n, const float origo[2],
                                 const float poseWorld[3]) {
  L.currMarkFreeIndex = L.currUpdateIndex + 1;
  L.currMarkOccIndex = L.currUpdateIndex + 2;
  float mapPose[3];
  map_coords_pose(L, poseWorld, mapPose);
  const Affine2 poseTransform = pose_transform(mapPose[0], mapPose[1], mapPose[2]);
  float bx, by;
  affine_apply(poseTransform, origo[0], origo[1], bx, by);
  const int bxi = (int)(bx + 0.5f);  // Vector2i(float, float): truncation
  const int byi = (int)(by + 0.5f);
  for (int i = 0; i < n; ++i) {
    float ex, ey;
    affine_apply(poseTransform, pts[2 * i], pts[2 * i + 1], ex, ey);
    ex += 0.5f;
    ey += 0.5f;
    const int exi = (int)ex;
    const int eyi = (int)ey;
    if (bxi != exi || byi != eyi) {
      update_line_bresenhami(L, bxi, byi, exi, eyi);
    }
  }
  L.lastUpdateIndex++;      // setUpdated(), GridMapBase.h:343
  L.currUpdateIndex += 3;   // :167
}

struct Ctx {
  std::vector<Level> levels;
  // MapRepMultiMap::dataContainers (MapRepMultiMap.h:171): scaled copies retained by matchData
  std::vector<std::vector<float> > coarsePts;
  std::vector<float> coarseOrigo;  // 2 per coarse level
  // HectorSlamProcessor state (HectorSlamProcessor.h:145-152)
  float lastMapUpdatePose[3], lastScanMatchPose[3], lastScanMatchCov[9];
  float paramMinDist, paramMinAngle;
};

// DataPointContainer::setFrom (HSL/scan/DataPointContainer.h:46-58)
static void set_from(std::vector<float>& dst, float dstOrigo[2], const float* pts, int n,
                     const float origo[2], float factor) {
  dstOrigo[0] = origo[0] * factor;
  dstOrigo[1] = origo[1] * factor;
  dst.assign(pts, pts + 2 * (size_t)n);
  for (size_t i = 0; i < dst.size(); ++i) dst[i] *= factor;
}

// a7: MapRepMultiMap::matchData (MapRepMultiMap.h:116-132)
static void multimap_match(Ctx& c, const float beginWorld[3], const float* pts, int n,
                           const float origo[2], float outWorld[3], float cov[9]) {
  float tmp[3] = {beginWorld[0], beginWorld[1], beginWorld[2]};
  const int size = (int)c.levels.size();
  for (int index = size - 1; index >= 0; --index) {
    float next[3];
    if (index == 0) {
      scan_matcher_match_data(c.levels[0], tmp, pts, n, 5, next, cov);
    } else {
      const float factor = (float)(1.0 / pow(2.0, (double)index));
      set_from(c.coarsePts[index - 1], &c.coarseOrigo[2 * (index - 1)], pts, n, origo, factor);
      scan_matcher_match_data(c.levels[index], tmp, c.coarsePts[index - 1].data(), n, 3, next, cov);
    }
    tmp[0] = next[0];
    tmp[1] = next[1];
    tmp[2] = next[2];
  }
  outWorld[0] = tmp[0];
  outWorld[1] = tmp[1];
  outWorld[2] = tmp[2];
}

// MapRepMultiMap::updateByScan (MapRepMultiMap.h:134-147)
static void multimap_update(Ctx& c, const float* pts, int n, const float origo[2],
                            const float poseWorld[3]) {
  for (size_t i = 0; i < c.levels.size(); ++i) {
    if (i == 0) {
      level_update_by_scan(c.levels[0], pts, n, origo, poseWorld);
    } else {
      const std::vector<float>& p = c.coarsePts[i - 1];
      level_update_by_scan(c.levels[i], p.data(), (int)(p.size() / 2), &c.coarseOrigo[2 * (i - 1)],
                           poseWorld);
    }
  }
}

static void multimap_on_map_updated(Ctx& c) {  // MapRepMultiMap.h:107-114 -> GridMapCacheArray.h:69-72
  for (size_t i = 0; i < c.levels.size(); ++i) c.levels[i].currCacheIndex++;
}

// a8: util::poseDifferenceLargerThan (UtilFunctions.h:73-92); abs() is the float overload (row a8)
static bool pose_difference_larger_than(const float p1[3], const float p2[3], float distThresh,
                                        float angThresh) {
  const float dx = p1[0] - p2[0];
  const float dy = p1[1] - p2[1];
  if (sqrtf(dx * dx + dy * dy) > distThresh) return true;
  float angleDiff = (p1[2] - p2[2]);
  if (angleDiff > M_PI) {
    angleDiff -= M_PI * 2.0f;
  } else if (angleDiff < -M_PI) {
    angleDiff += M_PI * 2.0f;
  }
  if (fabsf(angleDiff) > angThresh) return true;
  return false;
}

static void proc_reset(Ctx& c) {  // HectorSlamProcessor::reset (:115-124) -> MapProcContainer::reset (:67-71)
  for (int i = 0; i < 3; ++i) {
    c.lastMapUpdatePose[i] = FLT_MAX;
    c.lastScanMatchPose[i] = 0.0f;
  }
  for (size_t i = 0; i < c.levels.size(); ++i) {
    level_clear(c.levels[i]);
    c.levels[i].currCacheIndex++;
  }
}

}  // namespace

extern "C" {

void* ho_create(float mapResolution, int mapSizeX, int mapSizeY, unsigned levels, float startX,
                float startY) {
  Ctx* c = new Ctx();
  // MapRepMultiMap ctor (MapRepMultiMap.h:48-72)
  int rx = mapSizeX, ry = mapSizeY;
  const float totalMapSizeX = mapResolution * (float)mapSizeX;
  const float mid_offset_x = totalMapSizeX * startX;
  const float totalMapSizeY = mapResolution * (float)mapSizeY;
  const float mid_offset_y = totalMapSizeY * startY;
  c->levels.resize(levels);
  for (unsigned i = 0; i < levels; ++i) {
    level_init(c->levels[i], mapResolution, rx, ry, mid_offset_x, mid_offset_y);
    rx /= 2;
    ry /= 2;
    mapResolution *= 2.0f;
  }
  c->coarsePts.resize(levels > 0 ? levels - 1 : 0);
  c->coarseOrigo.assign(levels > 0 ? 2 * (levels - 1) : 0, 0.0f);
  for (int i = 0; i < 9; ++i) c->lastScanMatchCov[i] = 0.0f;
  proc_reset(*c);                // HectorSlamProcessor.h:60
  c->paramMinDist = 0.4f * 1.0f;   // :62
  c->paramMinAngle = 0.13f * 1.0f; // :63
  return c;
}
void ho_destroy(void* h) { delete (Ctx*)h; }
void ho_reset(void* h) { proc_reset(*(Ctx*)h); }
int ho_levels(void* h) { return (int)((Ctx*)h)->levels.size(); }
float ho_scale_to_map(void* h) { return ((Ctx*)h)->levels[0].scaleToMap; }
void ho_set_update_factor_free(void* h, float f) {
  Ctx* c = (Ctx*)h;
  for (size_t i = 0; i < c->levels.size(); ++i) c->levels[i].logOddsFree = prob_to_log_odds(f);
}
void ho_set_update_factor_occupied(void* h, float f) {
  Ctx* c = (Ctx*)h;
  for (size_t i = 0; i < c->levels.size(); ++i) c->levels[i].logOddsOccupied = prob_to_log_odds(f);
}
void ho_level_info(void* h, int level, int* sx, int* sy, float* cell, float* scale) {
  const Level& L = ((Ctx*)h)->levels[level];
  *sx = L.sx;
  *sy = L.sy;
  *cell = L.cellLength;
  *scale = L.scaleToMap;
}
void ho_download_level(void* h, int level, float* lo, int* ui) {
  const Level& L = ((Ctx*)h)->levels[level];
  const size_t n = (size_t)L.sx * L.sy;
  if (lo) memcpy(lo, L.logOdds.data(), n * sizeof(float));
  if (ui) memcpy(ui, L.updateIndex.data(), n * sizeof(int));
}
void ho_upload_level(void* h, int level, const float* lo, const int* ui) {
  Level& L = ((Ctx*)h)->levels[level];
  const size_t n = (size_t)L.sx * L.sy;
  if (lo) memcpy(L.logOdds.data(), lo, n * sizeof(float));
  if (ui) memcpy(L.updateIndex.data(), ui, n * sizeof(int));
  L.currCacheIndex++;
}
void ho_map_coords_pose(void* h, int level, const float w[3], float m[3]) {
  map_coords_pose(((Ctx*)h)->levels[level], w, m);
}
void ho_world_coords_pose(void* h, int level, const float m[3], float w[3]) {
  world_coords_pose(((Ctx*)h)->levels[level], m, w);
}
void ho_interp(void* h, int level, const float* xy, int n, float* out) {
  Level& L = ((Ctx*)h)->levels[level];
  for (int i = 0; i < n; ++i)
    interp_with_derivs(L, xy[2 * i], xy[2 * i + 1], out[3 * i], out[3 * i + 1], out[3 * i + 2]);
}
void ho_hessian_derivs(void* h, int level, const float pose[3], const float* pts, int n, float H[9],
                       float dTr[3]) {
  complete_hessian_derivs(((Ctx*)h)->levels[level], pose, pts, n, H, dTr);
}
void ho_match_level(void* h, int level, const float begin[3], const float* pts, int n, int maxIter,
                    float out[3], float cov[9]) {
  scan_matcher_match_data(((Ctx*)h)->levels[level], begin, pts, n, maxIter, out, cov);
}
void ho_match(void* h, const float begin[3], const float* pts, int n, const float origo[2],
              float out[3], float cov[9]) {
  multimap_match(*(Ctx*)h, begin, pts, n, origo, out, cov);
}
void ho_match_many(void* h, int batch, const float* begin, const float* pts, const int* offs, float* out) {
  static const float zero[2] = {0.0f, 0.0f};
  float cov[9];
  for (int b = 0; b < batch; ++b)
    multimap_match(*(Ctx*)h, begin + 3 * b, pts + 2 * (size_t)offs[b], offs[b + 1] - offs[b], zero,
                   out + 3 * b, cov);
}
void ho_update_by_scan(void* h, const float pose[3], const float* pts, int n, const float origo[2]) {
  multimap_update(*(Ctx*)h, pts, n, origo, pose);
}
void ho_update_by_scan_level(void* h, int level, const float pose[3], const float* pts, int n,
                             const float origo[2]) {
  level_update_by_scan(((Ctx*)h)->levels[level], pts, n, origo, pose);
}
void ho_on_map_updated(void* h) { multimap_on_map_updated(*(Ctx*)h); }
long ho_undefined_reads(void* h) {  // (test-harness guard, see Level::undefinedReads)
  long n = 0;
  for (const Level& L : ((Ctx*)h)->levels) n += L.undefinedReads;
  return n;
}

void ho_proc_set_thresholds(void* h, float d, float a) {
  ((Ctx*)h)->paramMinDist = d;
  ((Ctx*)h)->paramMinAngle = a;
}
// a12: HectorSlamProcessor::update (HectorSlamProcessor.h:71-113)
void ho_proc_update(void* h, const float* pts, int n, const float origo[2], const float hint[3],
                    int mapWithoutMatching) {
  Ctx& c = *(Ctx*)h;
  float newPose[3];
  if (!mapWithoutMatching) {
    multimap_match(c, hint, pts, n, origo, newPose, c.lastScanMatchCov);
  } else {
    newPose[0] = hint[0];
    newPose[1] = hint[1];
    newPose[2] = hint[2];
  }
  for (int i = 0; i < 3; ++i) c.lastScanMatchPose[i] = newPose[i];
  if (pose_difference_larger_than(newPose, c.lastMapUpdatePose, c.paramMinDist, c.paramMinAngle) ||
      mapWithoutMatching) {
    multimap_update(c, pts, n, origo, newPose);
    multimap_on_map_updated(c);
    for (int i = 0; i < 3; ++i) c.lastMapUpdatePose[i] = newPose[i];
  }
}
void ho_proc_last_pose(void* h, float pose[3], float cov[9]) {
  Ctx& c = *(Ctx*)h;
  for (int i = 0; i < 3; ++i) pose[i] = c.lastScanMatchPose[i];
  for (int i = 0; i < 9; ++i) cov[i] = c.lastScanMatchCov[i];
}
// f3: OccGridMapUtil::interpMapValue (OccGridMapUtil.h:233-285), getResidualForState (:198-214),
// getLikelihoodForResidual (:191-197), getLikelihoodForState (:184-189)
static inline float interp_map_value(Level& L, float cx, float cy) {
  if (cx != cx || cy != cy) {  // (test-harness guard, see Level::undefinedReads)
    ++L.undefinedReads;
    return 0.0f;
  }
  if ((cx < 0.0f) || (cx > L.limx) || (cy < 0.0f) || (cy > L.limy)) return 0.0f;
  const int ix = (int)cx, iy = (int)cy;
  const float fx = cx - (float)ix, fy = cy - (float)iy;
  int index = iy * L.sx + ix;
  const float i0 = cached_prob(L, index);
  ++index;
  const float i1 = cached_prob(L, index);
  index += L.sx - 1;
  const float i2 = cached_prob(L, index);
  ++index;
  const float i3 = cached_prob(L, index);
  const float xFacInv = (1.0f - fx);
  const float yFacInv = (1.0f - fy);
  return ((i0 * xFacInv + i1 * fx) * (yFacInv)) + ((i2 * xFacInv + i3 * fx) * (fy));
}
void ho_likelihood_states(void* h, int level, int batch, const float* states, const float* pts, int n,
                          float* out) {
  Level& L = ((Ctx*)h)->levels[level];
  for (int b = 0; b < batch; ++b) {
    const Affine2 T = pose_transform(states[3 * b], states[3 * b + 1], states[3 * b + 2]);
    float residual = 0.0f;
    for (int i = 0; i < n; ++i) {
      float tx, ty;
      affine_apply(T, pts[2 * i], pts[2 * i + 1], tx, ty);
      const float funval = 1.0f - interp_map_value(L, tx, ty);
      residual += funval;
    }
    const float sizef = (float)n;
    out[b] = 1 - (residual / sizef);
  }
}

// getResidualForState (OccGridMapUtil.h:205-221)
static float residual_for_state(Level& L, const float st[3], const float* pts, int n) {
  const Affine2 T = pose_transform(st[0], st[1], st[2]);
  float residual = 0.0f;
  for (int i = 0; i < n; ++i) {
    float tx, ty;
    affine_apply(T, pts[2 * i], pts[2 * i + 1], tx, ty);
    const float funval = 1.0f - interp_map_value(L, tx, ty);
    residual += funval;
  }
  return residual;
}

void ho_residual_states(void* h, int level, int batch, const float* states, const float* pts, int n, float* out) {
  Level& L = ((Ctx*)h)->levels[level];
  for (int b = 0; b < batch; ++b) out[b] = residual_for_state(L, states + 3 * b, pts, n);
}

// getCovarianceForPose (OccGridMapUtil.h:106-160), getCovMatrixWorldCoords (:162-188)
void ho_covariance_for_poses(void* h, int level, int batch, const float* poses, const float* pts, int n,
                             float* out_map, float* out_world, float* out_lh7) {
  Level& L = ((Ctx*)h)->levels[level];
  for (int b = 0; b < batch; ++b) {
    const float deltaTransX = 1.5f, deltaTransY = 1.5f, deltaAng = 0.05f;
    const float x = poses[3 * b], y = poses[3 * b + 1], ang = poses[3 * b + 2];
    const float sp[7][3] = {{x + deltaTransX, y, ang}, {x - deltaTransX, y, ang}, {x, y + deltaTransY, ang},
                            {x, y - deltaTransY, ang}, {x, y, ang + deltaAng},    {x, y, ang - deltaAng}, {x, y, ang}};
    float lh[7];
    for (int i = 0; i < 7; ++i) {
      const float resid = residual_for_state(L, sp[i], pts, n);
      const float sizef = (float)n;
      lh[i] = 1 - (resid / sizef);
    }
    // likelihoods.sum() of a fixed 7-vector: no packet fits, completely unrolled halves (0..2) + (3..6)
    const float sum = ((lh[0] + (lh[1] + lh[2])) + ((lh[3] + lh[4]) + (lh[5] + lh[6])));
    const float invLhNormalizer = 1 / sum;
    float mean[3] = {0.0f, 0.0f, 0.0f};
    for (int i = 0; i < 7; ++i)
      for (int r = 0; r < 3; ++r) mean[r] += sp[i][r] * lh[i];
    for (int r = 0; r < 3; ++r) mean[r] *= invLhNormalizer;
    float cov[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 7; ++i) {
      const float d[3] = {sp[i][0] - mean[0], sp[i][1] - mean[1], sp[i][2] - mean[2]};
      const float w = lh[i] * invLhNormalizer;
      for (int c = 0; c < 3; ++c)
        for (int r = 0; r < 3; ++r) cov[c * 3 + r] += w * (d[r] * d[c]);
    }
    if (out_lh7) memcpy(out_lh7 + 7 * b, lh, sizeof lh);
    if (out_map) memcpy(out_map + 9 * b, cov, sizeof cov);
    if (out_world) {
      const float scaleTrans = L.cellLength, scaleTransSq = scaleTrans * scaleTrans;
      float* W = out_world + 9 * b;
      W[0] = cov[0] * scaleTransSq;
      W[4] = cov[4] * scaleTransSq;
      W[1] = cov[1] * scaleTransSq;
      W[3] = W[1];
      W[2] = cov[2] * scaleTrans;
      W[6] = W[2];
      W[5] = cov[5] * scaleTrans;
      W[7] = W[5];
      W[8] = cov[8];
    }
  }
}

// f4: DistanceMeasurementProvider::getDist / checkOccupancyBresenhami / bresenham2D
// (hector_map_tools/include/hector_map_tools/HectorMapTools.h:133-234), CoordinateTransformer :58-98
static int hmt_bresenham2d(const signed char* data, unsigned int abs_da, unsigned int abs_db, int error_b, int offset_a,
                           int offset_b, unsigned int offset, unsigned int max_length) {
  const unsigned int end = max_length < abs_da ? max_length : abs_da;
  for (unsigned int i = 0; i < end; ++i) {
    if (data[offset] == 100) return (int)offset;
    offset += offset_a;
    error_b += abs_db;
    if ((unsigned int)error_b >= abs_da) {
      offset += offset_b;
      error_b -= abs_da;
    }
  }
  return -1;
}
void ho_ray_distances(const signed char* grid, int sizeX, int sizeY, float origin_x, float origin_y, float resolution,
                        int n, const float* bw, const float* ew, float* out_dist, float* out_hit) {
  const float scale_ = resolution;
  const float inv_scale_ = 1.0f / resolution;
  for (int r = 0; r < n; ++r) {
    // getC2Coords: ((worldCoords - origo_) * inv_scale_).cast<int>()
    const int x0 = (int)((bw[2 * r] - origin_x) * inv_scale_), y0 = (int)((bw[2 * r + 1] - origin_y) * inv_scale_);
    const int x1 = (int)((ew[2 * r] - origin_x) * inv_scale_), y1 = (int)((ew[2 * r + 1] - origin_y) * inv_scale_);
    float dist = -1.0f;
    int end_offset = -1;
    if (!((x0 < 0) || (x0 >= sizeX) || (y0 < 0) || (y0 >= sizeY)) &&
        !((x1 < 0) || (x1 >= sizeX) || (y1 < 0) || (y1 >= sizeY))) {
      const int dx = x1 - x0, dy = y1 - y0;
      const unsigned int abs_dx = abs(dx), abs_dy = abs(dy);
      const int offset_dx = dx > 0 ? 1 : -1;
      const int offset_dy = (dy > 0 ? 1 : -1) * sizeX;
      const unsigned int startOffset = y0 * sizeX + x0;
      if (abs_dx >= abs_dy) {
        end_offset = hmt_bresenham2d(grid, abs_dx, abs_dy, abs_dx / 2, offset_dx, offset_dy, startOffset, 5000);
      } else {
        end_offset = hmt_bresenham2d(grid, abs_dy, abs_dx, abs_dy / 2, offset_dy, offset_dx, startOffset, 5000);
      }
      if (end_offset != -1) {
        const int ex = end_offset % sizeX, ey = end_offset / sizeX;
        const float fx = (float)(x0 - ex), fy = (float)(y0 - ey);
        const int distMap = (int)sqrtf(fx * fx + fy * fy);  // int distMap = (...).cast<float>().norm()
        dist = (float)distMap;
        out_hit[2 * r] = origin_x + ((float)ex * scale_);  // getC1Coords
        out_hit[2 * r + 1] = origin_y + ((float)ey * scale_);
      }
    }
    out_dist[r] = scale_ * dist;  // getC1Scale
  }
}

// f2: HectorMappingRos::publishMap cell loop (HM/src/HectorMappingRos.cpp:449-468) with
// LogOddsCell::isFree / isOccupied (GridMapLogOdds.h:76-84)
void ho_occupancy_grid(void* h, int level, signed char* out) {
  const Level& L = ((Ctx*)h)->levels[level];
  const int size = L.sx * L.sy;
  memset(out, -1, (size_t)size);
  for (int i = 0; i < size; ++i) {
    if (L.logOdds[i] < 0.0f) {
      out[i] = 0;
    } else if (L.logOdds[i] > 0.0f) {
      out[i] = 100;
    }
  }
}

// f1: HectorMappingRos::rosLaserScanToDataContainer (HM/src/HectorMappingRos.cpp:483-507).  `angle` is a
// float accumulated by repeated += (sequential rounding); cos/sin bind to the float overloads in the
// node's translation unit.
int ho_laser_scan_to_container(const float* ranges, int n, float angle_min, float angle_increment,
                               float range_min, float range_max, float scaleToMap, float* out_pts) {
  float angle = angle_min;
  int m = 0;
  const float maxRangeForContainer = range_max - 0.1f;
  for (int i = 0; i < n; ++i) {
    float dist = ranges[i];
    if ((dist > range_min) && (dist < maxRangeForContainer)) {
      dist *= scaleToMap;
      out_pts[2 * m] = cosf(angle) * dist;
      out_pts[2 * m + 1] = sinf(angle) * dist;
      ++m;
    }
    angle += angle_increment;
  }
  return m;
}

int ho_point_cloud_to_container(const float* pts, int n, const double T[12], float sqr_min, float sqr_max, float z_min,
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

int ho_project_laser(const float* ranges, int n, float angle_min, float angle_increment, float range_min,
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

float ho_normalize_angle(float a) { return normalize_angle(a); }
int ho_pose_difference_larger_than(const float p1[3], const float p2[3], float d, float a) {
  return pose_difference_larger_than(p1, p2, d, a) ? 1 : 0;
}

void ho_libm_sincosf(int n, const float* x, float* out_sin, float* out_cos) {
  for (int i = 0; i < n; ++i) {
    sincosf(x[i], &out_sin[i], &out_cos[i]);  // what GCC makes of the reference's sin(pose[2]), cos(pose[2]) pair
  }
}
void ho_libm_expf(int n, const float* x, float* out_exp, float* out_prob) {
  for (int i = 0; i < n; ++i) {
    const float odds = expf(x[i]);
    out_exp[i] = odds;
    out_prob[i] = odds / (odds + 1.0f);
  }
}


}  // extern "C"
