// slam_driver.cpp -- a stand-in for hector_mapping's ROS node (HectorMappingRos.cpp) that drives the
// reference's UNCHANGED HectorSlamProcessor the way scanCallback / publishMap do, from a scenario
// file.  TEST INFRASTRUCTURE.  The same source is compiled twice (oracle/Makefile):
//   _ref/slam_driver_ref    against the reference's own include tree            (CPU reference)
//   _ref/slam_driver_mi355  against an overlay of that tree in which only
//                           slam_main/MapRepMultiMap.h is replaced by ours       (GPU drop-in)
// and tests/test_facade_dropin.py compares what the two print.  Nothing here knows which one it is.
//
// scenario file (little endian): float res; int size, levels; float free, occ, minDist, minAng;
//   int hooks (bit 0: draw/debug hooks, bit 1: concurrent publisher thread), n_steps; then per step: float hint[3]; int use_last_pose, map_without_matching;
//   float origo[2]; int n; float pts[2n]
// output file: per step float pose[3], cov[9]; then hook log; then per level the mirror grid.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <thread>
#include <vector>

#include "slam_main/HectorSlamProcessor.h"

namespace {

// like hector_mapping/src/HectorMapMutex.h (a boost::mutex there): a real lock, plus call counters
struct Locker : public MapLockerInterface {
  std::mutex m;
  std::atomic<int> locks{0}, unlocks{0};
  virtual void lockMap() { m.lock(); ++locks; }
  virtual void unlockMap() { ++unlocks; m.unlock(); }
};

std::vector<float> g_log;  // flat record of every hook call: tag, argc, args...

struct Draw : public DrawInterface {
  void rec(float tag, std::initializer_list<double> a) {
    g_log.push_back(tag);
    g_log.push_back((float)a.size());
    for (double v : a) g_log.push_back((float)v);
  }
  virtual void drawPoint(const Eigen::Vector2f& p) { rec(1, {p[0], p[1]}); }
  virtual void drawArrow(const Eigen::Vector3f& p) { rec(2, {p[0], p[1], p[2]}); }
  virtual void drawCovariance(const Eigen::Vector2f& m, const Eigen::Matrix2f& c) { rec(3, {m[0], m[1], c(0, 0), c(1, 1)}); }
  virtual void setScale(double s) { rec(4, {s}); }
  virtual void setColor(double r, double g, double b, double a = 1.0) { rec(5, {r, g, b, a}); }
  virtual void sendAndResetData() { rec(6, {}); }
};

struct Debug : public HectorDebugInfoInterface {
  virtual void sendAndResetData() { g_log.push_back(7); g_log.push_back(0); }
  virtual void addHessianMatrix(const Eigen::Matrix3f& H) {
    g_log.push_back(8);
    g_log.push_back(9);
    for (int c = 0; c < 3; ++c) for (int r = 0; r < 3; ++r) g_log.push_back(H(r, c));
  }
  virtual void addPoseLikelihood(float lh) { g_log.push_back(9); g_log.push_back(1); g_log.push_back(lh); }
};

// the processor keeps its map representation protected; the batch phase below needs it
struct ProcAccess : public hectorslam::HectorSlamProcessor {
  ProcAccess(float res, int sx, int sy, const Eigen::Vector2f& start, int levels, DrawInterface* d,
             HectorDebugInfoInterface* dbg)
      : hectorslam::HectorSlamProcessor(res, sx, sy, start, levels, d, dbg) {}
  hectorslam::MapRepresentationInterface* rep() { return mapRep; }
};

template <typename T> T rd(FILE* f) {
  T v;
  if (fread(&v, sizeof v, 1, f) != 1) { fprintf(stderr, "slam_driver: short scenario file\n"); exit(2); }
  return v;
}
template <typename T> void wr(FILE* f, const T& v) { fwrite(&v, sizeof v, 1, f); }

}  // namespace

int main(int argc, char** argv) {
  if (argc != 3) { fprintf(stderr, "usage: %s scenario.bin out.bin\n", argv[0]); return 2; }
  FILE* in = fopen(argv[1], "rb");
  FILE* out = fopen(argv[2], "wb");
  if (!in || !out) { perror("slam_driver"); return 2; }
  const float res = rd<float>(in);
  const int size = rd<int>(in), levels = rd<int>(in);
  const float ffree = rd<float>(in), focc = rd<float>(in), minDist = rd<float>(in), minAng = rd<float>(in);
  const int hooks = rd<int>(in), steps = rd<int>(in);

  Draw draw;
  Debug debug;
  // HectorMappingRos.cpp:127-134
  ProcAccess* slam = new ProcAccess(res, size, size, Eigen::Vector2f(0.5f, 0.5f), levels, (hooks & 1) ? &draw : 0,
                                    (hooks & 1) ? &debug : 0);
  slam->setUpdateFactorFree(ffree);
  slam->setUpdateFactorOccupied(focc);
  slam->setMapUpdateMinDistDiff(minDist);
  slam->setMapUpdateMinAngleDiff(minAng);
  Locker* locker = new Locker();  // owned by the map representation from here on
  slam->addMapMutex(0, locker);

  // optional map-publisher thread like HectorMappingRos::publishMapLoop (:577-595): fetch the grid, then read
  // every cell under the map mutex, while the main thread keeps matching and updating
  std::atomic<bool> stop(false);
  std::atomic<long> published(0), occupiedSeen(0);
  std::thread publisher;
  if (hooks & 2) {
    publisher = std::thread([&]() {
      while (!stop.load()) {
        const hectorslam::GridMap& g = slam->getGridMap(0);
        MapLockerInterface* mtx = slam->getMapMutex(0);
        mtx->lockMap();
        long occ = 0;
        const int cells = g.getSizeX() * g.getSizeY();
        for (int i = 0; i < cells; ++i) occ += g.isOccupied(i) ? 1 : 0;
        mtx->unlockMap();
        occupiedSeen = occ;
        ++published;
        std::this_thread::sleep_for(std::chrono::milliseconds(2));
      }
    });
  }

  std::vector<double> update_us;  // wall time of every HectorSlamProcessor::update() call (SLAM_DRIVER_TIMING)
  hectorslam::DataContainer scan;
  std::vector<hectorslam::DataContainer> kept;  // the last scans, for the batch phase
  std::vector<Eigen::Vector3f> keptPose;
  for (int t = 0; t < steps; ++t) {
    float hint[3];
    for (int k = 0; k < 3; ++k) hint[k] = rd<float>(in);
    const int use_last = rd<int>(in), mwm = rd<int>(in);
    const float ox = rd<float>(in), oy = rd<float>(in);
    const int n = rd<int>(in);
    scan.clear();
    scan.setOrigo(Eigen::Vector2f(ox, oy));
    for (int i = 0; i < n; ++i) {
      const float x = rd<float>(in), y = rd<float>(in);
      scan.add(Eigen::Vector2f(x, y));
    }
    // scanCallback: start estimate = last pose (+ the scenario's odometry delta) or the given hint
    Eigen::Vector3f start(hint[0], hint[1], hint[2]);
    if (use_last) start += slam->getLastScanMatchPose();
    const std::chrono::steady_clock::time_point t_begin = std::chrono::steady_clock::now();
    slam->update(scan, start, mwm != 0);
    update_us.push_back(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_begin).count());
    const Eigen::Vector3f& p = slam->getLastScanMatchPose();
    const Eigen::Matrix3f& c = slam->getLastScanMatchCovariance();
    for (int k = 0; k < 3; ++k) wr(out, p[k]);
    for (int k = 0; k < 9; ++k) wr(out, c(k % 3, k / 3));
    if (t >= steps - 8) {
      kept.push_back(scan);
      keptPose.push_back(p);
    }
  }
  if (const char* tp = getenv("SLAM_DRIVER_TIMING")) {
    // latency of the node's per-scan call: one JSON line {steps, median_us, p90_us, min_us} (first 5 steps dropped)
    std::vector<double> v(update_us.begin() + (update_us.size() > 10 ? 5 : 0), update_us.end());
    std::sort(v.begin(), v.end());
    if (FILE* tf = fopen(tp, "w")) {
      if (!v.empty())
        fprintf(tf, "{\"steps\": %zu, \"median_us\": %.2f, \"p90_us\": %.2f, \"min_us\": %.2f}\n", v.size(),
                v[v.size() / 2], v[(v.size() * 9) / 10], v[0]);
      fclose(tf);
    }
  }
  if (publisher.joinable()) {
    while (published.load() < 3) std::this_thread::sleep_for(std::chrono::milliseconds(1));  // at least 3 full reads
    stop = true;
    publisher.join();
  }
  // batch phase: the kept scans re-matched from slightly displaced starts.  Reference build: one
  // matchData call per scan (the only form it has); MI355X build: ONE matchDataBatch launch.
  {
    const size_t logMark = g_log.size();  // the batch phase is not part of the hook-stream comparison
    std::vector<Eigen::Vector3f> hints, poses(kept.size());
    for (size_t k = 0; k < kept.size(); ++k)
      hints.push_back(keptPose[k] + Eigen::Vector3f(0.03f, -0.02f, 0.01f));
#ifdef HECTOR_MI355_CAPI_H
    std::vector<const hectorslam::DataContainer*> ptrs;
    for (size_t k = 0; k < kept.size(); ++k) ptrs.push_back(&kept[k]);
    static_cast<hectorslam::MapRepMultiMap*>(slam->rep())->matchDataBatch(hints, ptrs, poses);
#else
    Eigen::Matrix3f cov;
    for (size_t k = 0; k < kept.size(); ++k) poses[k] = slam->rep()->matchData(hints[k], kept[k], cov);
#endif
    g_log.resize(logMark);
    wr(out, (int)poses.size());
    for (size_t k = 0; k < poses.size(); ++k)
      for (int j = 0; j < 3; ++j) wr(out, poses[k][j]);
  }
  // hypotheses phase (SLAM_DRIVER_HYPOTHESES=N, e.g. 4096: BASELINE configs[2]'s particle-filter-style use): N start estimates
  // around the last pose, all looking at the LAST scan.  Reference build: N matchData calls; MI355X build: ONE matchDataBatch whose
  // N container pointers are the same container.  The poses go to a file of their own (SLAM_DRIVER_HYP_OUT), the time of the call
  // (median of 7 repetitions after 2 warm-up calls on the MI355X build; one pass on the reference) to SLAM_DRIVER_HYP_TIMING.
  if (const char* hn = getenv("SLAM_DRIVER_HYPOTHESES")) {
    const int N = atoi(hn);
    if (N > 0 && !kept.empty()) {
      const size_t logMark = g_log.size();
      const hectorslam::DataContainer& last = kept.back();
      std::vector<Eigen::Vector3f> hints((size_t)N), poses((size_t)N);
      unsigned lcg = 12345u;
      auto uni = [&]() { lcg = lcg * 1664525u + 1013904223u; return (float)((lcg >> 8) & 0xffffu) / 65535.0f * 2.0f - 1.0f; };
      for (int k = 0; k < N; ++k) {
        const float dx = uni(), dy = uni(), dth = uni();
        hints[(size_t)k] = keptPose.back() + Eigen::Vector3f(0.30f * dx, 0.30f * dy, 0.10f * dth);
      }
      std::vector<double> call_us;
#ifdef HECTOR_MI355_CAPI_H
      std::vector<const hectorslam::DataContainer*> ptrs((size_t)N, &last);
      for (int rep = 0; rep < 9; ++rep) {
        const std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
        static_cast<hectorslam::MapRepMultiMap*>(slam->rep())->matchDataBatch(hints, ptrs, poses);
        if (rep >= 2) call_us.push_back(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
      }
#else
      Eigen::Matrix3f cov;
      const std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
      for (int k = 0; k < N; ++k) poses[(size_t)k] = slam->rep()->matchData(hints[(size_t)k], last, cov);
      call_us.push_back(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
#endif
      g_log.resize(logMark);
      if (const char* ho = getenv("SLAM_DRIVER_HYP_OUT"))
        if (FILE* hf = fopen(ho, "wb")) {
          for (int k = 0; k < N; ++k)
            for (int j = 0; j < 3; ++j) wr(hf, poses[(size_t)k][j]);
          fclose(hf);
        }
      if (const char* ht = getenv("SLAM_DRIVER_HYP_TIMING"))
        if (FILE* tf = fopen(ht, "w")) {
          std::sort(call_us.begin(), call_us.end());
          fprintf(tf, "{\"hypotheses\": %d, \"beams\": %d, \"median_us_all_hypotheses\": %.2f, \"min_us\": %.2f, \"repetitions\": %zu}\n", N,
                  last.getSize(), call_us[call_us.size() / 2], call_us[0], call_us.size());
          fclose(tf);
        }
    }
  }
  wr(out, (int)g_log.size());
  if (!g_log.empty()) fwrite(&g_log[0], sizeof(float), g_log.size(), out);
  wr(out, (int)locker->locks.load());
  wr(out, (int)locker->unlocks.load());
  wr(out, slam->getScaleToMap());
  wr(out, slam->getMapLevels());
  // publishMap (HectorMappingRos.cpp:435-481): read every cell of every level through getGridMap()
  for (int l = 0; l < slam->getMapLevels(); ++l) {
    const hectorslam::GridMap& g = slam->getGridMap(l);
    wr(out, g.getSizeX());
    wr(out, g.getSizeY());
    wr(out, g.getCellLength());
    wr(out, g.getUpdateIndex());
    const int cells = g.getSizeX() * g.getSizeY();
    for (int i = 0; i < cells; ++i) {
      const signed char occ = g.isOccupied(i) ? 100 : (g.isFree(i) ? 0 : -1);
      wr(out, occ);
    }
    for (int i = 0; i < cells; ++i) wr(out, g.getCell(i).getValue());
  }
  delete slam;
  fclose(in);
  fclose(out);
  return 0;
}
