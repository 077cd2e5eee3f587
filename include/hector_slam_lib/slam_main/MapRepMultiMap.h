// MapRepMultiMap.h -- MI355X drop-in for hector_mapping's multi-resolution map representation.
//
// Replaces: hector_mapping/include/hector_slam_lib/slam_main/MapRepMultiMap.h of the reference
// (class hectorslam::MapRepMultiMap, :45-173).  Same class name, same constructor, same
// MapRepresentationInterface virtuals (MapRepresentationInterface.h:38-62), so
// HectorSlamProcessor.h:58 (`new MapRepMultiMap(...)`) and HectorMappingRos.cpp compile against it
// unchanged -- see INTEGRATION.md for the two-line CMake change.
//
// Where the work happens: every virtual forwards to the C ABI of libhector_mi355.so
// (include/hector_mi355/capi.h); the pyramid, the probability texels, the Gauss-Newton matcher
// and the log-odds update are HIP kernels on the GPU.  What stays on the host:
//   * one hectorslam::GridMap per level as a MIRROR, because getGridMap() must hand out a real
//     `const GridMap&` that the map publisher reads cell by cell from another thread
//     (HectorMappingRos.cpp:435-481).  The mirror is refreshed LAZILY: the library accumulates the
//     union of the touched cell boxes, and getGridMap() downloads that box when
//     somebody actually asks for the grid -- the 0.5 Hz publisher, not the 40 Hz scan callback,
//     whose updateByScan() therefore costs no device-to-host traffic at all.
//   * the DrawInterface / HectorDebugInfoInterface hooks (ScanMatcher.h:56-66,100-115): when
//     either is non-null the match records a per-step trace on the device and the hooks are
//     replayed from it in the reference's order.
//
// This header contains no compute path of its own: if the library cannot create a device
// context the constructor throws std::runtime_error (the reference has no error channel; a
// SLAM node without its matcher must not keep running on silently wrong poses).
#ifndef _hectormaprepmultimap_h__
#define _hectormaprepmultimap_h__

#include <cmath>
#include <cstddef>
#include <iostream>
#include <mutex>
#include <cstdlib>
#include <stdexcept>
#include <string>
#include <vector>

#include "MapRepresentationInterface.h"

#include "../map/GridMap.h"
#include "../scan/DataPointContainer.h"
#include "../util/DrawInterface.h"
#include "../util/HectorDebugInfoInterface.h"
#include "../util/MapLockerInterface.h"

#include "hector_mi355/capi.h"

namespace hectorslam {

class MapRepMultiMap : public MapRepresentationInterface
{
public:
  // MapRepMultiMap.h:48-72 of the reference: level i has (mapSize >> i) cells of mapResolution * 2^i
  MapRepMultiMap(float mapResolution, int mapSizeX, int mapSizeY, unsigned int numDepth,
                 const Eigen::Vector2f& startCoords, DrawInterface* drawInterfaceIn,
                 HectorDebugInfoInterface* debugInterfaceIn)
    : ctx(0)
    , drawInterface(drawInterfaceIn)
    , debugInterface(debugInterfaceIn)
  {
    static_assert(sizeof(LogOddsCell) == 8, "mirror cells are {float logOddsVal; int updateIndex}");
    // the library's default layout (float4 texels: one gather per beam; the apply pass of the update writes them, so
    // there is no separate texel pass any more): 41.0 us per HectorSlamProcessor::update against 43.3 us with the
    // plane layout (1081-beam scans, 3 levels).  HSM_LAYOUT=plane in the environment selects the plane layout
    // (16 B/cell less memory; the faster one for dense 16 k-beam scans, whose cost is the update).
    hsm_opts opts;
    opts.device = -1;
    opts.layout = HSM_LAYOUT_AUTO;
    opts.waves_per_scan = 0;
    if (hsm_create(mapResolution, mapSizeX, mapSizeY, numDepth, startCoords.x(), startCoords.y(), &opts, &ctx) != HSM_OK) {
      throw std::runtime_error(std::string("hector_mi355: ") + hsm_last_error());
    }

    Eigen::Vector2i resolution(mapSizeX, mapSizeY);
    const float mid_offset_x = mapResolution * static_cast<float>(mapSizeX) * startCoords.x();
    const float mid_offset_y = mapResolution * static_cast<float>(mapSizeY) * startCoords.y();

    for (unsigned int i = 0; i < numDepth; ++i) {
      std::cout << "HectorSM map lvl " << i << ": cellLength: " << mapResolution << " res x:" << resolution.x()
                << " res y: " << resolution.y() << " (MI355X resident)\n";
      mirrors.push_back(new GridMap(mapResolution, resolution, Eigen::Vector2f(mid_offset_x, mid_offset_y)));
      mutexes.push_back(0);
      forceRefresh.push_back(false);
      resolution /= 2;
      mapResolution *= 2.0f;
    }
    traceBuf.resize(12 * static_cast<size_t>(hsm_gn_iterations_per_match(ctx)));
  }

  virtual ~MapRepMultiMap()
  {
    for (size_t i = 0; i < mirrors.size(); ++i) {
      delete mirrors[i];
      if (mutexes[i]) {
        delete mutexes[i];  // the reference owns and deletes the lockers (MapProcContainer.h:56-65)
      }
    }
    hsm_destroy(ctx);
  }

  virtual void reset()
  {
    std::lock_guard<std::mutex> lk(mirrorMutex);
    hsm_reset(ctx);
    for (size_t i = 0; i < mirrors.size(); ++i) {
      mirrors[i]->reset();
      forceRefresh[i] = true;  // hsm_reset marked the whole level dirty on the device
    }
  }

  virtual float getScaleToMap() const { return mirrors[0]->getScaleToMap(); }

  virtual int getMapLevels() const { return static_cast<int>(mirrors.size()); }
  virtual const GridMap& getGridMap(int mapLevel) const
  {
    refreshMirror(mapLevel);
    return *mirrors[mapLevel];
  }

  virtual void addMapMutex(int i, MapLockerInterface* mapMutex)
  {
    if (mutexes[i]) {
      delete mutexes[i];
    }
    mutexes[i] = mapMutex;
  }

  MapLockerInterface* getMapMutex(int i) { return mutexes[i]; }

  // the probability texels are refreshed by the update kernels themselves; nothing is cached stale
  virtual void onMapUpdated() { hsm_on_map_updated(ctx); }

  virtual Eigen::Vector3f matchData(const Eigen::Vector3f& beginEstimateWorld, const DataContainer& dataContainer,
                                    Eigen::Matrix3f& covMatrix)
  {
    const int n = dataContainer.getSize();
    const float begin[3] = {beginEstimateWorld[0], beginEstimateWorld[1], beginEstimateWorld[2]};
    const Eigen::Vector2f o(dataContainer.getOrigo());
    const float origo[2] = {o[0], o[1]};
    float pose[3] = {begin[0], begin[1], begin[2]};
    float cov[9];
    for (int c = 0; c < 3; ++c) {
      for (int r = 0; r < 3; ++r) {
        cov[3 * c + r] = covMatrix(r, c);
      }
    }
    // std::vector<Eigen::Vector2f> is n contiguous {x, y} float pairs (DataPointContainer.h:92)
    const float* pts = n > 0 ? &dataContainer.getVecEntry(0)[0] : 0;
    const bool hooks = (drawInterface != 0) || (debugInterface != 0);
    int steps = 0;
    if (hooks) {
      hsm_match_trace(ctx, begin, pts, n, origo, pose, cov, &traceBuf[0], static_cast<int>(traceBuf.size() / 12), &steps);
      replayHooks(beginEstimateWorld, dataContainer, steps);
    } else {
      hsm_match(ctx, begin, pts, n, origo, pose, cov);
    }
    if (n != 0) {  // empty scan: covMatrix untouched (ScanMatcher.h:68,189)
      for (int c = 0; c < 3; ++c) {
        for (int r = 0; r < 3; ++r) {
          covMatrix(r, c) = cov[3 * c + r];
        }
      }
    }
    return Eigen::Vector3f(pose[0], pose[1], pose[2]);
  }

  virtual void updateByScan(const DataContainer& dataContainer, const Eigen::Vector3f& robotPoseWorld)
  {
    const int n = dataContainer.getSize();
    const float pose[3] = {robotPoseWorld[0], robotPoseWorld[1], robotPoseWorld[2]};
    const Eigen::Vector2f o(dataContainer.getOrigo());
    const float origo[2] = {o[0], o[1]};
    const float* pts = n > 0 ? &dataContainer.getVecEntry(0)[0] : 0;

    // The reference takes each level's locker around its update (MapProcContainer.h:103-116) because the update
    // writes the cells the publisher thread reads.  Here the update writes DEVICE planes that nobody else reads;
    // the cells the publisher reads are the host mirrors, and those are only written by refreshMirror() -- under
    // the level's locker.  So a reader still sees either the state before or after an update, never one in between.
    hsm_update_by_scan(ctx, pose, pts, n, origo);
  }

  virtual void setUpdateFactorFree(float free_factor)
  {
    hsm_set_update_factor_free(ctx, free_factor);
    for (size_t i = 0; i < mirrors.size(); ++i) {
      mirrors[i]->setUpdateFreeFactor(free_factor);
    }
  }

  virtual void setUpdateFactorOccupied(float occupied_factor)
  {
    hsm_set_update_factor_occupied(ctx, occupied_factor);
    for (size_t i = 0; i < mirrors.size(); ++i) {
      mirrors[i]->setUpdateOccupiedFactor(occupied_factor);
    }
  }

  // ---- extension (not in the reference): B independent (pose hypothesis, scan) pairs in one launch.
  // scans[i] may all be the same container (particle-filter style hypotheses of one scan).
  void matchDataBatch(const std::vector<Eigen::Vector3f>& beginEstimatesWorld,
                      const std::vector<const DataContainer*>& scans, std::vector<Eigen::Vector3f>& posesOut,
                      std::vector<Eigen::Matrix3f>* covOut = 0)
  {
    const int B = static_cast<int>(beginEstimatesWorld.size());
    posesOut.resize(B);
    if (B == 0) {
      return;
    }
    std::vector<float> begin(3 * static_cast<size_t>(B)), pose(3 * static_cast<size_t>(B)), cov;
    bool shared = true;  // every hypothesis looks at the same container: the particle-filter case
    for (int i = 0; i < B; ++i) {
      shared = shared && scans[i] == scans[0];
      for (int k = 0; k < 3; ++k) {
        begin[3 * i + k] = beginEstimatesWorld[i][k];
      }
    }
    if (covOut) {
      cov.assign(9 * static_cast<size_t>(B), 0.0f);
    }
    if (shared) {
      // ONE scan travels (hsm_match_batch with scan_offsets == NULL: the start poses and the results cross PCIe once each, the
      // scan is copied once) -- not B copies of it packed into a CSR array
      const int n = scans[0]->getSize();
      std::vector<float> pts(2 * static_cast<size_t>(n));
      for (int j = 0; j < n; ++j) {
        const Eigen::Vector2f& p = scans[0]->getVecEntry(j);
        pts[2 * static_cast<size_t>(j)] = p[0];
        pts[2 * static_cast<size_t>(j) + 1] = p[1];
      }
      hsm_match_batch(ctx, B, &begin[0], pts.empty() ? 0 : &pts[0], 0, n, &pose[0], covOut ? &cov[0] : 0);
    } else {
      std::vector<int> offsets(static_cast<size_t>(B) + 1, 0);
      for (int i = 0; i < B; ++i) {
        offsets[i + 1] = offsets[i] + scans[i]->getSize();
      }
      std::vector<float> pts(2 * static_cast<size_t>(offsets[B]));
      for (int i = 0; i < B; ++i) {
        const int n = scans[i]->getSize();
        for (int j = 0; j < n; ++j) {
          const Eigen::Vector2f& p = scans[i]->getVecEntry(j);
          pts[2 * (static_cast<size_t>(offsets[i]) + j)] = p[0];
          pts[2 * (static_cast<size_t>(offsets[i]) + j) + 1] = p[1];
        }
      }
      hsm_match_batch(ctx, B, &begin[0], pts.empty() ? 0 : &pts[0], &offsets[0], 0, &pose[0], covOut ? &cov[0] : 0);
    }
    for (int i = 0; i < B; ++i) {
      posesOut[i] = Eigen::Vector3f(pose[3 * i], pose[3 * i + 1], pose[3 * i + 2]);
    }
    if (covOut) {
      covOut->resize(B);
      for (int i = 0; i < B; ++i) {
        for (int c = 0; c < 3; ++c) {
          for (int r = 0; r < 3; ++r) {
            (*covOut)[i](r, c) = cov[9 * i + 3 * c + r];
          }
        }
      }
    }
  }

  hsm_ctx* getDeviceContext() { return ctx; }

protected:
  // bring the host mirror of `level` up to date: fetch-and-clear the union of the cell boxes the updates
  // touched since the last refresh, download it as the reference's AoS
  // cells, and bump the update counter like OccGridMapBase::updateByScan does (OccGridMapBase.h:164)
  void refreshMirror(int level) const
  {
    std::lock_guard<std::mutex> lk(mirrorMutex);
    GridMap& m = *mirrors[level];
    // the update counter is read ONCE, before the dirty box is taken: an update the scan thread queues after this
    // point may or may not be inside the box fetched below, so the mirror only claims to be as new as `target`
    // and the next getGridMap() looks again
    const int target = hsm_update_index(ctx, level);
    if (m.getUpdateIndex() == target && !forceRefresh[level]) {
      return;  // nothing happened on the device since the last refresh
    }
    forceRefresh[level] = false;
    // the mirror is written under the level's locker: a publisher thread reading cells under that lock
    // (HectorMappingRos.cpp:453-476) never observes a refresh in progress.  getGridMap() itself is called
    // before the publisher takes the lock (it is an argument of publishMap), so this cannot self-deadlock.
    if (mutexes[level]) {
      mutexes[level]->lockMap();
    }
    int bb[4];
    if (hsm_take_dirty_bbox(ctx, level, bb) == HSM_OK && bb[2] >= bb[0] && bb[3] >= bb[1]) {
      LogOddsCell* first = &m.getCell(bb[0], bb[1]);
      hsm_download_cells(ctx, level, bb[0], bb[1], bb[2], bb[3], first, m.getSizeX());
    }
    while (m.getUpdateIndex() < target) {
      m.setUpdated();
    }
    if (mutexes[level]) {
      mutexes[level]->unlockMap();
    }
  }

  // ScanMatcher::matchData's draw/debug calls (ScanMatcher.h:56-66,100-115,228-239), replayed from
  // the device trace: per GN step the map-frame estimate after the step and the Hessian used.
  void replayHooks(const Eigen::Vector3f& beginEstimateWorld, const DataContainer& dataContainer, int steps)
  {
    const int levels = static_cast<int>(mirrors.size());
    Eigen::Vector3f levelBeginWorld(beginEstimateWorld);
    int s = 0;
    DataContainer scaled;
    for (int level = levels - 1; level >= 0; --level) {
      const GridMap& m = *mirrors[level];
      const DataContainer* dc = &dataContainer;
      if (level > 0) {
        scaled.setFrom(dataContainer, static_cast<float>(1.0 / pow(2.0, static_cast<double>(level))));
        dc = &scaled;
      }
      if (drawInterface) {
        drawInterface->setScale(0.05f);
        drawInterface->setColor(0.0f, 1.0f, 0.0f);
        drawInterface->drawArrow(levelBeginWorld);
        drawScan(m.getMapCoordsPose(levelBeginWorld), m, *dc);
        drawInterface->setColor(1.0, 0.0, 0.0);
      }
      const int numIter = (level == 0) ? 5 : 3;
      Eigen::Vector3f estimate(m.getMapCoordsPose(levelBeginWorld));
      if (dc->getSize() != 0) {
        for (int i = -1; i < numIter && s < steps; ++i, ++s) {
          const float* t = &traceBuf[12 * static_cast<size_t>(s)];
          estimate = Eigen::Vector3f(t[0], t[1], t[2]);
          if (i < 0) {
            continue;  // the unconditional first step has no hooks (ScanMatcher.h:74)
          }
          if (drawInterface) {
            const float invNumIterf = 1.0f / static_cast<float>(numIter);
            drawInterface->setColor(static_cast<float>(i) * invNumIterf, 0.0f, 0.0f);
            drawInterface->drawArrow(m.getWorldCoordsPose(estimate));
          }
          if (debugInterface) {
            Eigen::Matrix3f H;
            for (int c = 0; c < 3; ++c) {
              for (int r = 0; r < 3; ++r) {
                H(r, c) = t[3 + 3 * c + r];
              }
            }
            debugInterface->addHessianMatrix(H);
          }
        }
        if (drawInterface) {
          drawInterface->setColor(0.0, 0.0, 1.0);
          drawScan(estimate, m, *dc);
        }
        estimate[2] = util::normalize_angle(estimate[2]);
        levelBeginWorld = m.getWorldCoordsPose(estimate);
      }
    }
  }

  void drawScan(const Eigen::Vector3f& poseMap, const GridMap& m, const DataContainer& dc)
  {
    drawInterface->setScale(0.02);
    const Eigen::Affine2f transform(Eigen::Translation2f(poseMap[0], poseMap[1]) * Eigen::Rotation2Df(poseMap[2]));
    const int size = dc.getSize();
    for (int i = 0; i < size; ++i) {
      drawInterface->drawPoint(m.getWorldCoords(transform * dc.getVecEntry(i)));
    }
  }

  hsm_ctx* ctx;
  std::vector<GridMap*> mirrors;
  std::vector<MapLockerInterface*> mutexes;
  mutable std::vector<bool> forceRefresh;
  mutable std::mutex mirrorMutex;
  std::vector<float> traceBuf;
  DrawInterface* drawInterface;
  HectorDebugInfoInterface* debugInterface;

private:
  MapRepMultiMap(const MapRepMultiMap&);
  MapRepMultiMap& operator=(const MapRepMultiMap&);
};

}

#endif
