// TEST INFRASTRUCTURE ONLY -- private stand-in for the subset of Eigen3 that
// hector_mapping's header library uses.  Eigen3 is a third-party dependency of
// the reference (find_package(Eigen3 REQUIRED), hector_mapping/CMakeLists.txt:16,
// version not pinned; de-facto 3.3.x) and is absent from /root/reference and
// from this image.  This file exists so that the UNMODIFIED reference headers
// can be compiled straight from /root/reference (see oracle/Makefile ->
// oracle/_ref/) and used to pin the plain-C++ restatement in
// oracle/hector_oracle.cpp.  Nothing in the product path includes it.
//
// Arithmetic restated from Eigen 3.3.x (evaluation order matters for fp32
// bit-parity; each item names the Eigen source it follows):
//  * fixed-size redux (sum of n coefficients) is a balanced binary split:
//    sum3 = x0 + (x1 + x2), sum2 = x0 + x1           (Core/Redux.h, redux_novec_unroller)
//  * small fixed-size matrix products are coefficient based:
//    (A*B)(i,j) = sum_k A(i,k)*B(k,j) with the redux above   (Core/ProductEvaluators.h)
//  * Transform<Affine> * vector = translation + linear*v     (Geometry/Transform.h,
//    transform_right_product_impl case 2: res = t; res.noalias() += linear*v)
//  * Transform::inverse() (Affine): linear^-1 via compute_inverse, then
//    t' = (-linear^-1) * t                                   (Geometry/Transform.h)
//  * 2x2 / 3x3 inverse by cofactors times invdet = 1/det     (LU/InverseImpl.h)
//  * Rotation2D::toRotationMatrix uses std::sin/std::cos of the scalar type
//    (Geometry/Rotation2D.h); Translation*Rotation = isometry with t untouched
//  * DiagonalMatrix(Scaling) * Translation: linear = diag, t = diag * t
//    (Geometry/Translation.h, operator*(EigenBase, Translation))
//  * float -> int cast<>() and mixed-scalar constructors truncate (static_cast)
#ifndef ORACLE_MINI_EIGEN_H
#define ORACLE_MINI_EIGEN_H

// The reference relies on Eigen to drag these in (std::cout, memcpy, UINT_MAX,
// abs, pow): MapRepMultiMap.h:60,127, GridMapBase.h:202, OccGridMapBase.h:170,193.
#include <iostream>
#include <cstring>
#include <climits>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <cstddef>

#define EIGEN_MAKE_ALIGNED_OPERATOR_NEW

namespace Eigen {

enum TransformTraits { Isometry = 0x1, Affine = 0x2, AffineCompact = 0x10 | Affine, Projective = 0x20 };

namespace internal {
// balanced-split redux, Core/Redux.h redux_novec_unroller<Func, Derived, Start, Length>
template <typename T, int Start, int Length> struct redux_sum {
  enum { Half = Length / 2 };
  template <typename F> static inline T run(const F& f) {
    return redux_sum<T, Start, Half>::run(f) + redux_sum<T, Start + Half, Length - Half>::run(f);
  }
};
template <typename T, int Start> struct redux_sum<T, Start, 1> {
  template <typename F> static inline T run(const F& f) { return f(Start); }
};
template <typename T, int Start> struct redux_sum<T, Start, 0> {
  template <typename F> static inline T run(const F&) { return T(0); }
};
}  // namespace internal

template <typename T, int R, int C> class Array;
template <typename T, int R, int C> class Matrix;

template <typename M, int BR, int BC> class BlockRef {
 public:
  typedef typename M::Scalar Scalar;
  BlockRef(M& m, int r0, int c0) : m_(m), r0_(r0), c0_(c0) {}
  BlockRef& operator=(const Matrix<Scalar, BR, BC>& o) {
    for (int c = 0; c < BC; ++c)
      for (int r = 0; r < BR; ++r) m_(r0_ + r, c0_ + c) = o(r, c);
    return *this;
  }
  operator Matrix<Scalar, BR, BC>() const {
    Matrix<Scalar, BR, BC> out;
    for (int c = 0; c < BC; ++c)
      for (int r = 0; r < BR; ++r) out(r, c) = m_(r0_ + r, c0_ + c);
    return out;
  }
  Matrix<Scalar, BR, BC> eval() const { return Matrix<Scalar, BR, BC>(*this); }
  Matrix<Scalar, BR, BC> operator*(Scalar s) const { return eval() * s; }
  Matrix<Scalar, BR, BC> operator-(const Matrix<Scalar, BR, BC>& o) const { return eval() - o; }
  Matrix<Scalar, BR, BC> inverse() const { return eval().inverse(); }

 private:
  M& m_;
  int r0_, c0_;
};

// column-major fixed-size dense matrix (Eigen default storage order)
template <typename T, int R, int C> class Matrix {
 public:
  typedef T Scalar;
  enum { RowsAtCompileTime = R, ColsAtCompileTime = C, SizeAtCompileTime = R * C };
  T d[R * C];

  Matrix() {}
  template <typename A, typename B> Matrix(const A& x, const B& y) {
    static_assert(R * C == 2, "2-coefficient constructor");
    d[0] = static_cast<T>(x);
    d[1] = static_cast<T>(y);
  }
  template <typename A, typename B, typename D> Matrix(const A& x, const B& y, const D& z) {
    static_assert(R * C == 3, "3-coefficient constructor");
    d[0] = static_cast<T>(x);
    d[1] = static_cast<T>(y);
    d[2] = static_cast<T>(z);
  }
  template <typename A, typename B, typename D, typename E>
  Matrix(const A& x, const B& y, const D& z, const E& w) {
    static_assert(R * C == 4, "4-coefficient constructor");
    d[0] = static_cast<T>(x);
    d[1] = static_cast<T>(y);
    d[2] = static_cast<T>(z);
    d[3] = static_cast<T>(w);
  }
  Matrix(const Array<T, R, C>& a);

  static Matrix Zero() {
    Matrix m;
    for (int i = 0; i < R * C; ++i) m.d[i] = T(0);
    return m;
  }
  static Matrix Identity() {
    Matrix m = Zero();
    for (int i = 0; i < (R < C ? R : C); ++i) m(i, i) = T(1);
    return m;
  }
  void setZero() { *this = Zero(); }

  T& operator()(int r, int c) { return d[c * R + r]; }
  const T& operator()(int r, int c) const { return d[c * R + r]; }
  T& operator()(int i) { return d[i]; }
  const T& operator()(int i) const { return d[i]; }
  T& operator[](int i) { return d[i]; }
  const T& operator[](int i) const { return d[i]; }
  T& coeffRef(int r, int c) { return d[c * R + r]; }
  const T& coeff(int r, int c) const { return d[c * R + r]; }
  T& x() { return d[0]; }
  const T& x() const { return d[0]; }
  T& y() { return d[1]; }
  const T& y() const { return d[1]; }
  T& z() { return d[2]; }
  const T& z() const { return d[2]; }
  T& w() { return d[3]; }
  const T& w() const { return d[3]; }
  int rows() const { return R; }
  int cols() const { return C; }
  int size() const { return R * C; }
  T* data() { return d; }
  const T* data() const { return d; }

  template <int N> Matrix<T, N, 1> head() const {
    static_assert(C == 1 && N <= R, "head<N> on a column vector");
    Matrix<T, N, 1> out;
    for (int i = 0; i < N; ++i) out[i] = d[i];
    return out;
  }
  template <int N> Matrix<T, N, 1> start() const { return head<N>(); }
  template <int BR, int BC> BlockRef<Matrix, BR, BC> block(int r0, int c0) {
    return BlockRef<Matrix, BR, BC>(*this, r0, c0);
  }
  template <int BR, int BC> Matrix<T, BR, BC> block(int r0, int c0) const {
    Matrix<T, BR, BC> out;
    for (int c = 0; c < BC; ++c)
      for (int r = 0; r < BR; ++r) out(r, c) = (*this)(r0 + r, c0 + c);
    return out;
  }
  template <typename U> Matrix<U, R, C> cast() const {
    Matrix<U, R, C> out;
    for (int i = 0; i < R * C; ++i) out.d[i] = static_cast<U>(d[i]);
    return out;
  }
  Array<T, R, C>& array() { return *reinterpret_cast<Array<T, R, C>*>(this); }
  const Array<T, R, C>& array() const { return *reinterpret_cast<const Array<T, R, C>*>(this); }

  Matrix<T, C, R> transpose() const {
    Matrix<T, C, R> out;
    for (int c = 0; c < C; ++c)
      for (int r = 0; r < R; ++r) out(c, r) = (*this)(r, c);
    return out;
  }
  T sum() const {
    const T* p = d;
    return internal::redux_sum<T, 0, R * C>::run([p](int i) { return p[i]; });
  }
  T squaredNorm() const {
    const T* p = d;
    return internal::redux_sum<T, 0, R * C>::run([p](int i) { return p[i] * p[i]; });
  }
  T norm() const { return std::sqrt(squaredNorm()); }

  Matrix operator-() const {
    Matrix o;
    for (int i = 0; i < R * C; ++i) o.d[i] = -d[i];
    return o;
  }
  Matrix operator+(const Matrix& b) const {
    Matrix o;
    for (int i = 0; i < R * C; ++i) o.d[i] = d[i] + b.d[i];
    return o;
  }
  Matrix operator-(const Matrix& b) const {
    Matrix o;
    for (int i = 0; i < R * C; ++i) o.d[i] = d[i] - b.d[i];
    return o;
  }
  Matrix operator*(T s) const {
    Matrix o;
    for (int i = 0; i < R * C; ++i) o.d[i] = d[i] * s;
    return o;
  }
  Matrix operator/(T s) const {
    Matrix o;
    for (int i = 0; i < R * C; ++i) o.d[i] = d[i] / s;
    return o;
  }
  Matrix& operator+=(const Matrix& b) {
    for (int i = 0; i < R * C; ++i) d[i] += b.d[i];
    return *this;
  }
  Matrix& operator-=(const Matrix& b) {
    for (int i = 0; i < R * C; ++i) d[i] -= b.d[i];
    return *this;
  }
  Matrix& operator*=(T s) {
    for (int i = 0; i < R * C; ++i) d[i] *= s;
    return *this;
  }
  Matrix& operator/=(T s) {
    for (int i = 0; i < R * C; ++i) d[i] /= s;
    return *this;
  }
  bool operator==(const Matrix& b) const {
    for (int i = 0; i < R * C; ++i)
      if (!(d[i] == b.d[i])) return false;
    return true;
  }
  bool operator!=(const Matrix& b) const { return !(*this == b); }

  // coefficient-based product, Core/ProductEvaluators.h (lazy product coeff)
  template <int K> Matrix<T, R, K> operator*(const Matrix<T, C, K>& b) const {
    Matrix<T, R, K> o;
    for (int j = 0; j < K; ++j)
      for (int i = 0; i < R; ++i) {
        const Matrix* a = this;
        const Matrix<T, C, K>* bp = &b;
        o(i, j) = internal::redux_sum<T, 0, C>::run(
            [a, bp, i, j](int k) { return (*a)(i, k) * (*bp)(k, j); });
      }
    return o;
  }

  T determinant() const;
  Matrix inverse() const;
};

template <typename T, int R, int C> inline Matrix<T, R, C> operator*(T s, const Matrix<T, R, C>& m) {
  return m * s;
}

template <typename T, int R, int C>
inline std::ostream& operator<<(std::ostream& os, const Matrix<T, R, C>& m) {
  for (int r = 0; r < R; ++r) {
    for (int c = 0; c < C; ++c) os << (c ? " " : "") << m(r, c);
    if (r + 1 < R) os << "\n";
  }
  return os;
}

// coefficient-wise view; same storage as Matrix (array() reinterprets)
template <typename T, int R, int C> class Array {
 public:
  T d[R * C];
  Array operator+(T s) const {
    Array o;
    for (int i = 0; i < R * C; ++i) o.d[i] = d[i] + s;
    return o;
  }
  Array operator-(T s) const {
    Array o;
    for (int i = 0; i < R * C; ++i) o.d[i] = d[i] - s;
    return o;
  }
  Array& operator+=(T s) {
    for (int i = 0; i < R * C; ++i) d[i] += s;
    return *this;
  }
  Array& operator-=(T s) {
    for (int i = 0; i < R * C; ++i) d[i] -= s;
    return *this;
  }
  template <typename U> Array<U, R, C> cast() const {
    Array<U, R, C> o;
    for (int i = 0; i < R * C; ++i) o.d[i] = static_cast<U>(d[i]);
    return o;
  }
  Matrix<T, R, C> matrix() const { return Matrix<T, R, C>(*this); }
};

template <typename T, int R, int C> inline Matrix<T, R, C>::Matrix(const Array<T, R, C>& a) {
  for (int i = 0; i < R * C; ++i) d[i] = a.d[i];
}

namespace internal {
// LU/InverseImpl.h cofactor_3x3<MatrixType,i,j>
template <typename M> inline typename M::Scalar cofactor_3x3(const M& m, int i, int j) {
  const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
  return m.coeff(i1, j1) * m.coeff(i2, j2) - m.coeff(i1, j2) * m.coeff(i2, j1);
}
template <typename T, int N> struct inverse_impl;
template <typename T> struct inverse_impl<T, 1> {
  static T det(const Matrix<T, 1, 1>& m) { return m.coeff(0, 0); }
  static Matrix<T, 1, 1> inv(const Matrix<T, 1, 1>& m) {
    Matrix<T, 1, 1> r;
    r.coeffRef(0, 0) = T(1) / m.coeff(0, 0);
    return r;
  }
};
template <typename T> struct inverse_impl<T, 2> {
  // Core/Determinant (size 2) and LU/InverseImpl.h compute_inverse_size2_helper
  static T det(const Matrix<T, 2, 2>& m) {
    return m.coeff(0, 0) * m.coeff(1, 1) - m.coeff(1, 0) * m.coeff(0, 1);
  }
  static Matrix<T, 2, 2> inv(const Matrix<T, 2, 2>& m) {
    const T invdet = T(1) / det(m);
    Matrix<T, 2, 2> r;
    r.coeffRef(0, 0) = m.coeff(1, 1) * invdet;
    r.coeffRef(1, 0) = -m.coeff(1, 0) * invdet;
    r.coeffRef(0, 1) = -m.coeff(0, 1) * invdet;
    r.coeffRef(1, 1) = m.coeff(0, 0) * invdet;
    return r;
  }
};
template <typename T> struct inverse_impl<T, 3> {
  // LU/InverseImpl.h compute_inverse<MatrixType, ResultType, 3> + size3 helper;
  // Core/Determinant.h determinant_impl<Derived,3> (bruteforce_det3_helper)
  static T det3h(const Matrix<T, 3, 3>& m, int a, int b, int c) {
    return m.coeff(0, a) * (m.coeff(1, b) * m.coeff(2, c) - m.coeff(1, c) * m.coeff(2, b));
  }
  static T det(const Matrix<T, 3, 3>& m) {
    return det3h(m, 0, 1, 2) - det3h(m, 1, 0, 2) + det3h(m, 2, 0, 1);
  }
  static Matrix<T, 3, 3> inv(const Matrix<T, 3, 3>& m) {
    Matrix<T, 3, 1> cof0;
    cof0[0] = cofactor_3x3(m, 0, 0);
    cof0[1] = cofactor_3x3(m, 1, 0);
    cof0[2] = cofactor_3x3(m, 2, 0);
    // det = (cofactors_col0.cwiseProduct(matrix.col(0))).sum()  -> x0 + (x1 + x2)
    const T det = cof0[0] * m.coeff(0, 0) + (cof0[1] * m.coeff(1, 0) + cof0[2] * m.coeff(2, 0));
    const T invdet = T(1) / det;
    Matrix<T, 3, 3> r;
    r.coeffRef(0, 0) = cof0[0] * invdet;  // result.row(0) = cofactors_col0 * invdet
    r.coeffRef(0, 1) = cof0[1] * invdet;
    r.coeffRef(0, 2) = cof0[2] * invdet;
    r.coeffRef(1, 0) = cofactor_3x3(m, 0, 1) * invdet;
    r.coeffRef(1, 1) = cofactor_3x3(m, 1, 1) * invdet;
    r.coeffRef(1, 2) = cofactor_3x3(m, 2, 1) * invdet;
    r.coeffRef(2, 0) = cofactor_3x3(m, 0, 2) * invdet;
    r.coeffRef(2, 1) = cofactor_3x3(m, 1, 2) * invdet;
    r.coeffRef(2, 2) = cofactor_3x3(m, 2, 2) * invdet;
    return r;
  }
};
}  // namespace internal

template <typename T, int R, int C> inline T Matrix<T, R, C>::determinant() const {
  static_assert(R == C && R <= 3, "determinant: square, size <= 3 only");
  return internal::inverse_impl<T, R>::det(*this);
}
template <typename T, int R, int C> inline Matrix<T, R, C> Matrix<T, R, C>::inverse() const {
  static_assert(R == C && R <= 3, "inverse: square, size <= 3 only");
  return internal::inverse_impl<T, R>::inv(*this);
}

typedef Matrix<float, 2, 1> Vector2f;
typedef Matrix<float, 3, 1> Vector3f;
typedef Matrix<float, 4, 1> Vector4f;
typedef Matrix<double, 2, 1> Vector2d;
typedef Matrix<double, 3, 1> Vector3d;
typedef Matrix<int, 2, 1> Vector2i;
typedef Matrix<int, 3, 1> Vector3i;
typedef Matrix<float, 2, 2> Matrix2f;
typedef Matrix<float, 3, 3> Matrix3f;
typedef Matrix<float, 4, 4> Matrix4f;

// ---------------------------------------------------------------- Geometry
template <typename T, int Dim> class Translation {
 public:
  Matrix<T, Dim, 1> m_coeffs;
  Translation() {}
  Translation(const T& x, const T& y) {
    static_assert(Dim == 2, "2D translation");
    m_coeffs[0] = x;
    m_coeffs[1] = y;
  }
  Translation(const T& x, const T& y, const T& z) {
    static_assert(Dim == 3, "3D translation");
    m_coeffs[0] = x;
    m_coeffs[1] = y;
    m_coeffs[2] = z;
  }
  explicit Translation(const Matrix<T, Dim, 1>& v) : m_coeffs(v) {}
  const Matrix<T, Dim, 1>& vector() const { return m_coeffs; }
  const Matrix<T, Dim, 1>& translation() const { return m_coeffs; }
};

template <typename T> class Rotation2D {
 public:
  T m_angle;
  explicit Rotation2D(const T& a) : m_angle(a) {}
  T angle() const { return m_angle; }
  // Geometry/Rotation2D.h toRotationMatrix(): EIGEN_USING_STD_MATH(sin/cos) on Scalar
  Matrix<T, 2, 2> toRotationMatrix() const {
    using std::cos;
    using std::sin;
    const T sinA = sin(m_angle);
    const T cosA = cos(m_angle);
    Matrix<T, 2, 2> m;
    m(0, 0) = cosA;
    m(0, 1) = -sinA;
    m(1, 0) = sinA;
    m(1, 1) = cosA;
    return m;
  }
};

template <typename T, int Dim> class DiagonalMatrix {
 public:
  Matrix<T, Dim, 1> m_diag;
  DiagonalMatrix(const T& x, const T& y) {
    static_assert(Dim == 2, "2D scaling");
    m_diag[0] = x;
    m_diag[1] = y;
  }
  DiagonalMatrix(const T& x, const T& y, const T& z) {
    static_assert(Dim == 3, "3D scaling");
    m_diag[0] = x;
    m_diag[1] = y;
    m_diag[2] = z;
  }
  const Matrix<T, Dim, 1>& diagonal() const { return m_diag; }
};

template <typename T, int Dim, int Mode> class Transform {
 public:
  typedef Matrix<T, Dim, Dim> LinearMatrixType;
  typedef Matrix<T, Dim, 1> VectorType;
  LinearMatrixType m_linear;
  VectorType m_translation;

  Transform() {}
  template <int OtherMode> Transform(const Transform<T, Dim, OtherMode>& o)
      : m_linear(o.m_linear), m_translation(o.m_translation) {}

  const LinearMatrixType& linear() const { return m_linear; }
  LinearMatrixType& linear() { return m_linear; }
  const VectorType& translation() const { return m_translation; }
  VectorType& translation() { return m_translation; }

  Matrix<T, Dim + 1, Dim + 1> matrix() const {
    Matrix<T, Dim + 1, Dim + 1> m = Matrix<T, Dim + 1, Dim + 1>::Zero();
    for (int c = 0; c < Dim; ++c)
      for (int r = 0; r < Dim; ++r) m(r, c) = m_linear(r, c);
    for (int r = 0; r < Dim; ++r) m(r, Dim) = m_translation[r];
    m(Dim, Dim) = T(1);
    return m;
  }

  // Geometry/Transform.h transform_right_product_impl (Affine, Dim-row operand):
  //   res = translation; res.noalias() += linear * other   (lazy coefficient product)
  VectorType operator*(const VectorType& v) const {
    VectorType res(m_translation);
    const LinearMatrixType* l = &m_linear;
    const VectorType* vp = &v;
    for (int i = 0; i < Dim; ++i)
      res[i] += internal::redux_sum<T, 0, Dim>::run(
          [l, vp, i](int k) { return (*l)(i, k) * (*vp)[k]; });
    return res;
  }

  // Geometry/Transform.h Transform::inverse(hint = Mode)
  Transform inverse() const {
    Transform res;
    if (Mode == Isometry) {
      res.m_linear = m_linear.transpose();
    } else {
      res.m_linear = m_linear.inverse();
    }
    res.m_translation = (-res.m_linear) * m_translation;
    return res;
  }
};

typedef Transform<float, 2, Affine> Affine2f;
typedef Transform<float, 3, Affine> Affine3f;
typedef Transform<float, 2, Isometry> Isometry2f;
typedef Transform<float, 3, Isometry> Isometry3f;
typedef Translation<float, 2> Translation2f;
typedef Translation<float, 3> Translation3f;
typedef Rotation2D<float> Rotation2Df;
typedef DiagonalMatrix<float, 2> AlignedScaling2f;
typedef DiagonalMatrix<float, 3> AlignedScaling3f;

// Geometry/Translation.h: Translation * RotationBase -> *this * Isometry(r);
// Translation * Transform: res = t; res.pretranslate(m_coeffs)  (t.translation += coeffs)
template <typename T>
inline Transform<T, 2, Isometry> operator*(const Translation<T, 2>& t, const Rotation2D<T>& r) {
  Transform<T, 2, Isometry> res;
  res.m_linear = r.toRotationMatrix();
  res.m_translation = Matrix<T, 2, 1>::Zero();
  res.m_translation += t.m_coeffs;
  return res;
}

// Geometry/Translation.h: friend operator*(const EigenBase& linear, const Translation& t):
//   res.linear() = linear; res.translation() = linear * t.m_coeffs
template <typename T, int Dim>
inline Transform<T, Dim, Affine> operator*(const DiagonalMatrix<T, Dim>& s, const Translation<T, Dim>& t) {
  Transform<T, Dim, Affine> res;
  res.m_linear = Matrix<T, Dim, Dim>::Zero();
  for (int i = 0; i < Dim; ++i) {
    res.m_linear(i, i) = s.m_diag[i];
    res.m_translation[i] = s.m_diag[i] * t.m_coeffs[i];
  }
  return res;
}

}  // namespace Eigen

#endif  // ORACLE_MINI_EIGEN_H
