// TEST INFRASTRUCTURE ONLY -- private stand-in (oracle/stubs_node/) so that the UNMODIFIED node source
// /root/reference/hector_mapping/src/HectorMappingRos.cpp compiles without ROS / tf / boost (none is in this image).
// Nothing here is part of the product; see oracle/node_shim.cpp.
#pragma once
// tf's LinearMath value types with the arithmetic of tf 1.12/1.13 (tfScalar = double):
//   Vector3::dot           x*v.x + y*v.y + z*v.z                      (LinearMath/Vector3.h)
//   Transform * Vector3    (basis[0].dot(v) + origin.x, ...)          (LinearMath/Transform.h operator())
//   Matrix3x3::setRotation / getRotation / Transform::inverse / operator*  as in LinearMath/Matrix3x3.h, Transform.h
// Third-party code absent from /root/reference: restated, like the Eigen stand-in (oracle/stubs/Eigen).
#include <math.h>
#include <stdlib.h>

#include <stdexcept>
#include <string>

#include "geometry_msgs/Pose.h"
#include "ros/ros.h"
#define TF_SCALAR_H
typedef double tfScalar;
namespace tf {
class Vector3 {
 public:
  Vector3() : v_{0.0, 0.0, 0.0} {}
  Vector3(tfScalar x, tfScalar y, tfScalar z) : v_{x, y, z} {}
  tfScalar x() const { return v_[0]; }
  tfScalar y() const { return v_[1]; }
  tfScalar z() const { return v_[2]; }
  tfScalar getX() const { return v_[0]; }
  tfScalar getY() const { return v_[1]; }
  tfScalar getZ() const { return v_[2]; }
  tfScalar dot(const Vector3& o) const { return v_[0] * o.v_[0] + v_[1] * o.v_[1] + v_[2] * o.v_[2]; }
  Vector3 operator-() const { return Vector3(-v_[0], -v_[1], -v_[2]); }
  Vector3 operator+(const Vector3& o) const { return Vector3(v_[0] + o.v_[0], v_[1] + o.v_[1], v_[2] + o.v_[2]); }
  tfScalar v_[3];
};
class Quaternion {
 public:
  Quaternion() : x_(0.0), y_(0.0), z_(0.0), w_(1.0) {}
  Quaternion(tfScalar x, tfScalar y, tfScalar z, tfScalar w) : x_(x), y_(y), z_(z), w_(w) {}
  tfScalar x() const { return x_; }
  tfScalar y() const { return y_; }
  tfScalar z() const { return z_; }
  tfScalar w() const { return w_; }
  tfScalar length2() const { return x_ * x_ + y_ * y_ + z_ * z_ + w_ * w_; }
  tfScalar x_, y_, z_, w_;
};
class Matrix3x3 {
 public:
  Matrix3x3() { setIdentity(); }
  void setIdentity() { r_[0] = Vector3(1, 0, 0), r_[1] = Vector3(0, 1, 0), r_[2] = Vector3(0, 0, 1); }
  void setValue(tfScalar xx, tfScalar xy, tfScalar xz, tfScalar yx, tfScalar yy, tfScalar yz, tfScalar zx, tfScalar zy, tfScalar zz) {
    r_[0] = Vector3(xx, xy, xz), r_[1] = Vector3(yx, yy, yz), r_[2] = Vector3(zx, zy, zz);
  }
  void setRotation(const Quaternion& q) {  // Matrix3x3.h setRotation
    const tfScalar d = q.length2(), s = tfScalar(2.0) / d;
    const tfScalar xs = q.x() * s, ys = q.y() * s, zs = q.z() * s;
    const tfScalar wx = q.w() * xs, wy = q.w() * ys, wz = q.w() * zs;
    const tfScalar xx = q.x() * xs, xy = q.x() * ys, xz = q.x() * zs;
    const tfScalar yy = q.y() * ys, yz = q.y() * zs, zz = q.z() * zs;
    setValue(tfScalar(1.0) - (yy + zz), xy - wz, xz + wy, xy + wz, tfScalar(1.0) - (xx + zz), yz - wx, xz - wy, yz + wx,
             tfScalar(1.0) - (xx + yy));
  }
  const Vector3& operator[](int i) const { return r_[i]; }
  Vector3 column(int c) const { return Vector3(r_[0].v_[c], r_[1].v_[c], r_[2].v_[c]); }
  Matrix3x3 transpose() const {
    Matrix3x3 m;
    m.setValue(r_[0].x(), r_[1].x(), r_[2].x(), r_[0].y(), r_[1].y(), r_[2].y(), r_[0].z(), r_[1].z(), r_[2].z());
    return m;
  }
  Matrix3x3 operator*(const Matrix3x3& o) const {
    Matrix3x3 m;
    m.setValue(r_[0].dot(o.column(0)), r_[0].dot(o.column(1)), r_[0].dot(o.column(2)), r_[1].dot(o.column(0)),
               r_[1].dot(o.column(1)), r_[1].dot(o.column(2)), r_[2].dot(o.column(0)), r_[2].dot(o.column(1)),
               r_[2].dot(o.column(2)));
    return m;
  }
  void getRotation(Quaternion& q) const {  // Matrix3x3.h getRotation
    const tfScalar trace = r_[0].x() + r_[1].y() + r_[2].z();
    tfScalar t[4];
    if (trace > tfScalar(0.0)) {
      tfScalar s = sqrt(trace + tfScalar(1.0));
      t[3] = s * tfScalar(0.5);
      s = tfScalar(0.5) / s;
      t[0] = (r_[2].y() - r_[1].z()) * s;
      t[1] = (r_[0].z() - r_[2].x()) * s;
      t[2] = (r_[1].x() - r_[0].y()) * s;
    } else {
      const int i = r_[0].x() < r_[1].y() ? (r_[1].y() < r_[2].z() ? 2 : 1) : (r_[0].x() < r_[2].z() ? 2 : 0);
      const int j = (i + 1) % 3, k = (i + 2) % 3;
      tfScalar s = sqrt(r_[i].v_[i] - r_[j].v_[j] - r_[k].v_[k] + tfScalar(1.0));
      t[i] = s * tfScalar(0.5);
      s = tfScalar(0.5) / s;
      t[3] = (r_[k].v_[j] - r_[j].v_[k]) * s;
      t[j] = (r_[j].v_[i] + r_[i].v_[j]) * s;
      t[k] = (r_[k].v_[i] + r_[i].v_[k]) * s;
    }
    q = Quaternion(t[0], t[1], t[2], t[3]);
  }
  Vector3 r_[3];
};
class Transform {
 public:
  Transform() {}
  Transform(const Matrix3x3& b, const Vector3& c) : basis_(b), origin_(c) {}
  void setIdentity() {
    basis_.setIdentity();
    origin_ = Vector3(0, 0, 0);
  }
  void setOrigin(const Vector3& o) { origin_ = o; }
  void setBasis(const Matrix3x3& b) { basis_ = b; }
  void setRotation(const Quaternion& q) { basis_.setRotation(q); }
  const Vector3& getOrigin() const { return origin_; }
  const Matrix3x3& getBasis() const { return basis_; }
  Quaternion getRotation() const {
    Quaternion q;
    basis_.getRotation(q);
    return q;
  }
  // Transform.h: operator()(x) = Vector3(m_basis[0].dot(x) + m_origin.x(), m_basis[1].dot(x) + m_origin.y(), ...)
  Vector3 operator*(const Vector3& x) const {
    return Vector3(basis_[0].dot(x) + origin_.x(), basis_[1].dot(x) + origin_.y(), basis_[2].dot(x) + origin_.z());
  }
  Transform operator*(const Transform& t) const { return Transform(basis_ * t.basis_, (*this) * t.origin_); }
  Transform inverse() const {
    const Matrix3x3 inv = basis_.transpose();
    const Vector3 o = -origin_;
    return Transform(inv, Vector3(inv[0].dot(o), inv[1].dot(o), inv[2].dot(o)));
  }
 protected:
  Matrix3x3 basis_;
  Vector3 origin_;
};
class StampedTransform : public Transform {
 public:
  StampedTransform() {}
  StampedTransform(const Transform& t, const ros::Time& s, const std::string& f, const std::string& c)
      : Transform(t), stamp_(s), frame_id_(f), child_frame_id_(c) {}
  ros::Time stamp_;
  std::string frame_id_, child_frame_id_;
};
class TransformException : public std::runtime_error {
 public:
  explicit TransformException(const std::string& m) : std::runtime_error(m) {}
};
static inline double getYaw(const Quaternion& q) {  // transform_datatypes.h: Matrix3x3(q).getEulerYPR -> atan2 form restated in oracle/stubs
  return atan2(2.0 * (q.w_ * q.z_ + q.x_ * q.y_), 1.0 - 2.0 * (q.y_ * q.y_ + q.z_ * q.z_));
}
static inline void poseMsgToTF(const geometry_msgs::Pose& p, Transform& t) {
  t.setOrigin(Vector3(p.position.x, p.position.y, p.position.z));
  t.setRotation(Quaternion(p.orientation.x, p.orientation.y, p.orientation.z, p.orientation.w));
}
}  // namespace tf
