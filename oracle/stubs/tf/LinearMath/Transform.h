// TEST INFRASTRUCTURE ONLY -- private stand-in for the one tf header the
// reference library includes (hector_slam_lib/util/UtilFunctions.h:33) so the
// unmodified reference headers compile without ROS.  Only getYawFromQuat
// (UtilFunctions.h:94-97) uses tf, and nothing on the hot path calls it.
//
// IMPORTANT side effect reproduced on purpose: the real tf Scalar.h includes
// <math.h> and <stdlib.h> before any hector header is parsed.  With libstdc++
// that pulls `using std::sin; using std::exp; using std::abs; ...` into the
// global namespace, so the reference's unqualified sin/cos/exp/log/abs calls on
// floats resolve to the FLOAT overloads (sinf/cosf/expf/logf, float abs)
// exactly as in a ROS build.
#ifndef ORACLE_TF_TRANSFORM_STUB_H
#define ORACLE_TF_TRANSFORM_STUB_H
#include <math.h>
#include <stdlib.h>

namespace geometry_msgs {
struct Quaternion {
  double x, y, z, w;
};
}  // namespace geometry_msgs

namespace tf {
class Quaternion {
 public:
  Quaternion(double x, double y, double z, double w) : x_(x), y_(y), z_(z), w_(w) {}
  double x_, y_, z_, w_;
};
static inline double getYaw(const Quaternion& q) {
  return atan2(2.0 * (q.w_ * q.z_ + q.x_ * q.y_), 1.0 - 2.0 * (q.y_ * q.y_ + q.z_ * q.z_));
}
}  // namespace tf
#endif
