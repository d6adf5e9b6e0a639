// mini_eigen.h -- the small slice of Eigen's API that cpu_tsdf's library code (and this repo's C++
// host shell) uses, for machines without Eigen.  NOT Eigen: fixed-size, eager evaluation, no
// expression templates, no alignment tricks.
//
// Purpose: (1) lets the reference's own unmodified sources compile here into oracle/_ref (the parity
// oracle); (2) lets include/cpu_tsdf/*.h (the drop-in host shell) build and be tested without PCL/Eigen.
// With the real Eigen on the include path this directory is simply not used.
//
// Where Eigen's evaluation ORDER is observable in float results it is restated from Eigen 3.3.x (the
// series PCL >= 1.10 builds against) and marked [Eigen-recall]; none of it can be checked offline:
//   * reductions of n terms are complete-unrolled as a balanced tree: n=3 -> a0 + (a1 + a2),
//     n=4 -> (a0 + a1) + (a2 + a3)   (redux_novec_unroller)
//   * a matrix-vector product coefficient is such a reduction of the n products
//   * v /= s divides each coefficient (3.3; 3.2 multiplied by 1/s)
//   * normalize(): z = squaredNorm(); if (z > 0) v /= sqrt(z)
//   * Transform::inverse(Affine): 3x3 cofactor inverse times 1/det, translation = -(Rinv) * t
//   * Transform::rotation() runs an SVD in real Eigen; here it returns linear() (exact for rigid poses
//     up to the SVD's own rounding).
#pragma once

#include <cmath>
#include <cstddef>
#include <cstdlib>
#include <iomanip>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

#ifndef EIGEN_MAKE_ALIGNED_OPERATOR_NEW
#define EIGEN_MAKE_ALIGNED_OPERATOR_NEW
#endif

namespace Eigen {

template <typename T>
struct aligned_allocator : std::allocator<T> {
  template <typename U>
  struct rebind {
    typedef aligned_allocator<U> other;
  };
  aligned_allocator() {}
  template <typename U>
  aligned_allocator(const aligned_allocator<U> &) {}
};

namespace internal {
// balanced-tree reduction [Eigen-recall: redux_novec_unroller]
template <typename T>
inline T tree_sum(const T *a, int start, int len) {
  if (len == 1) return a[start];
  const int half = len / 2;
  return tree_sum(a, start, half) + tree_sum(a, start + half, len - half);
}
}  // namespace internal

template <typename T, int R, int C>
class Matrix;

template <typename T, int R, int C>
class CommaInit {
 public:
  CommaInit(Matrix<T, R, C> &m, T first) : m_(m), i_(0) { put(first); }
  CommaInit &operator,(T v) {
    put(v);
    return *this;
  }

 private:
  void put(T v) {
    m_(i_ / C, i_ % C) = v;
    ++i_;
  }
  Matrix<T, R, C> &m_;
  int i_;
};

template <typename T, int R, int C>
class ArrayWrap;

// Column-major fixed-size matrix / vector.
template <typename T, int R, int C>
class Matrix {
 public:
  typedef T Scalar;
  enum { RowsAtCompileTime = R, ColsAtCompileTime = C };
  Matrix() {
    for (int i = 0; i < R * C; ++i) d_[i] = T();
  }
  Matrix(int, int) {  // Eigen::Matrix<S,T,U>(rows, cols) -- sizes are fixed here
    for (int i = 0; i < R * C; ++i) d_[i] = T();
  }
  template <typename A, typename B, typename D>
  Matrix(A x, B y, D z) {
    static_assert(R * C == 3, "3-vector constructor");
    d_[0] = (T)x;
    d_[1] = (T)y;
    d_[2] = (T)z;
  }
  template <typename A, typename B, typename D, typename E>
  Matrix(A x, B y, D z, E w) {
    static_assert(R * C == 4, "4-vector constructor");
    d_[0] = (T)x;
    d_[1] = (T)y;
    d_[2] = (T)z;
    d_[3] = (T)w;
  }

  static Matrix Zero() { return Matrix(); }
  static Matrix Identity() {
    Matrix m;
    for (int i = 0; i < (R < C ? R : C); ++i) m(i, i) = T(1);
    return m;
  }
  static Matrix Constant(T v) {
    Matrix m;
    for (int i = 0; i < R * C; ++i) m.d_[i] = v;
    return m;
  }
  // Eigen: uniform in [-1, 1] from std::rand(), one call per coefficient [Eigen-recall]
  static Matrix Random() {
    Matrix m;
    for (int i = 0; i < R * C; ++i) m.d_[i] = (T)(2.0 * std::rand() / (double)RAND_MAX - 1.0);
    return m;
  }

  int rows() const { return R; }
  int cols() const { return C; }
  int size() const { return R * C; }
  T *data() { return d_; }
  const T *data() const { return d_; }

  T &operator()(int r, int c) { return d_[c * R + r]; }
  const T &operator()(int r, int c) const { return d_[c * R + r]; }
  T &operator()(int i) { return d_[i]; }
  const T &operator()(int i) const { return d_[i]; }
  T &operator[](int i) { return d_[i]; }
  const T &operator[](int i) const { return d_[i]; }
  T &coeffRef(int r, int c) { return (*this)(r, c); }
  const T &coeff(int r, int c) const { return (*this)(r, c); }
  T &x() { return d_[0]; }
  T &y() { return d_[1]; }
  T &z() { return d_[2]; }
  const T &x() const { return d_[0]; }
  const T &y() const { return d_[1]; }
  const T &z() const { return d_[2]; }

  CommaInit<T, R, C> operator<<(T first) { return CommaInit<T, R, C>(*this, first); }

  void setZero() {
    for (int i = 0; i < R * C; ++i) d_[i] = T();
  }
  void setIdentity() { *this = Identity(); }

  // topRightCorner<BR, BC>(): a writable view, enough for `m.topRightCorner<3,1>() *= s` (one product
  // per coefficient, as Eigen evaluates it)
  template <int BR, int BC>
  struct CornerRef {
    Matrix &m;
    CornerRef &operator*=(T s) {
      for (int c = 0; c < BC; ++c)
        for (int r = 0; r < BR; ++r) m(r, C - BC + c) = m(r, C - BC + c) * s;
      return *this;
    }
  };
  template <int BR, int BC>
  CornerRef<BR, BC> topRightCorner() {
    return CornerRef<BR, BC>{*this};
  }

  template <typename U>
  Matrix<U, R, C> cast() const {
    Matrix<U, R, C> m;
    for (int i = 0; i < R * C; ++i) m.data()[i] = (U)d_[i];
    return m;
  }

  Matrix operator+(const Matrix &o) const {
    Matrix m;
    for (int i = 0; i < R * C; ++i) m.d_[i] = d_[i] + o.d_[i];
    return m;
  }
  Matrix operator-(const Matrix &o) const {
    Matrix m;
    for (int i = 0; i < R * C; ++i) m.d_[i] = d_[i] - o.d_[i];
    return m;
  }
  Matrix operator-() const {
    Matrix m;
    for (int i = 0; i < R * C; ++i) m.d_[i] = -d_[i];
    return m;
  }
  Matrix operator*(T s) const {
    Matrix m;
    for (int i = 0; i < R * C; ++i) m.d_[i] = d_[i] * s;
    return m;
  }
  Matrix operator/(T s) const {
    Matrix m;
    for (int i = 0; i < R * C; ++i) m.d_[i] = d_[i] / s;
    return m;
  }
  friend Matrix operator*(T s, const Matrix &a) {
    Matrix m;
    for (int i = 0; i < R * C; ++i) m.d_[i] = s * a.d_[i];
    return m;
  }
  Matrix &operator+=(const Matrix &o) {
    for (int i = 0; i < R * C; ++i) d_[i] = d_[i] + o.d_[i];
    return *this;
  }
  Matrix &operator-=(const Matrix &o) {
    for (int i = 0; i < R * C; ++i) d_[i] = d_[i] - o.d_[i];
    return *this;
  }
  Matrix &operator*=(T s) {
    for (int i = 0; i < R * C; ++i) d_[i] = d_[i] * s;
    return *this;
  }
  Matrix &operator/=(T s) {  // [Eigen-recall 3.3] true division per coefficient
    for (int i = 0; i < R * C; ++i) d_[i] = d_[i] / s;
    return *this;
  }

  template <int K>
  Matrix<T, R, K> operator*(const Matrix<T, C, K> &o) const {
    Matrix<T, R, K> m;
    T prod[C];
    for (int r = 0; r < R; ++r)
      for (int k = 0; k < K; ++k) {
        for (int c = 0; c < C; ++c) prod[c] = (*this)(r, c) * o(c, k);
        m(r, k) = internal::tree_sum(prod, 0, C);
      }
    return m;
  }

  Matrix<T, C, R> transpose() const {
    Matrix<T, C, R> m;
    for (int r = 0; r < R; ++r)
      for (int c = 0; c < C; ++c) m(c, r) = (*this)(r, c);
    return m;
  }

  T dot(const Matrix &o) const {
    T prod[R * C];
    for (int i = 0; i < R * C; ++i) prod[i] = d_[i] * o.d_[i];
    return internal::tree_sum(prod, 0, R * C);
  }
  T squaredNorm() const { return dot(*this); }
  T norm() const { return std::sqrt(squaredNorm()); }
  void normalize() {
    const T z = squaredNorm();
    if (z > T(0)) *this /= std::sqrt(z);
  }
  Matrix normalized() const {
    Matrix m = *this;
    m.normalize();
    return m;
  }
  Matrix cross(const Matrix &o) const {
    static_assert(R * C == 3, "cross");
    return Matrix(d_[1] * o.d_[2] - d_[2] * o.d_[1], d_[2] * o.d_[0] - d_[0] * o.d_[2],
                  d_[0] * o.d_[1] - d_[1] * o.d_[0]);
  }
  T sum() const { return internal::tree_sum(d_, 0, R * C); }
  T minCoeff() const {
    T m = d_[0];
    for (int i = 1; i < R * C; ++i) m = d_[i] < m ? d_[i] : m;
    return m;
  }
  T maxCoeff() const {
    T m = d_[0];
    for (int i = 1; i < R * C; ++i) m = d_[i] > m ? d_[i] : m;
    return m;
  }
  bool operator==(const Matrix &o) const {
    for (int i = 0; i < R * C; ++i)
      if (d_[i] != o.d_[i]) return false;
    return true;
  }
  ArrayWrap<T, R, C> array() const;

  template <int BR, int BC>
  Matrix<T, BR, BC> block(int r0, int c0) const {
    Matrix<T, BR, BC> m;
    for (int r = 0; r < BR; ++r)
      for (int c = 0; c < BC; ++c) m(r, c) = (*this)(r0 + r, c0 + c);
    return m;
  }
  Matrix<T, R, 1> col(int c) const {
    Matrix<T, R, 1> m;
    for (int r = 0; r < R; ++r) m(r) = (*this)(r, c);
    return m;
  }

  // 3x3 inverse [Eigen-recall: compute_inverse_size3_helper]
  Matrix inverse() const {
    static_assert(R == 3 && C == 3, "only the 3x3 inverse is provided");
    const Matrix &a = *this;
    auto cof = [&](int i, int j) {
      const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
      return a(i1, j1) * a(i2, j2) - a(i1, j2) * a(i2, j1);
    };
    const T c0[3] = {cof(0, 0), cof(1, 0), cof(2, 0)};
    const T det = c0[0] * a(0, 0) + (c0[1] * a(1, 0) + c0[2] * a(2, 0));
    const T invdet = T(1) / det;
    Matrix r;
    for (int c = 0; c < 3; ++c) r(0, c) = c0[c] * invdet;
    r(1, 0) = cof(0, 1) * invdet;
    r(1, 1) = cof(1, 1) * invdet;
    r(2, 2) = cof(2, 2) * invdet;
    r(1, 2) = cof(2, 1) * invdet;
    r(2, 1) = cof(1, 2) * invdet;
    r(2, 0) = cof(0, 2) * invdet;
    return r;
  }

 private:
  T d_[R * C];
};

// operator<< for matrices [Eigen-recall: default IOFormat]: stream precision, columns right-aligned
// to the widest coefficient, " " between coefficients, "\n" between rows, no trailing newline.
template <typename T, int R, int C>
std::ostream &operator<<(std::ostream &s, const Matrix<T, R, C> &m) {
  std::streamsize width = 0;
  for (int j = 0; j < C; ++j)
    for (int i = 0; i < R; ++i) {
      std::stringstream ss;
      ss.copyfmt(s);
      ss << m(i, j);
      width = std::max<std::streamsize>(width, (std::streamsize)ss.str().length());
    }
  for (int i = 0; i < R; ++i) {
    if (i) s << "\n";
    for (int j = 0; j < C; ++j) {
      if (j) s << " ";
      if (width) s.width(width);
      s << m(i, j);
    }
  }
  return s;
}

// Coefficient-wise view (Eigen::Array).
template <typename T, int R, int C>
class ArrayWrap {
 public:
  typedef T Scalar;
  ArrayWrap() {}
  template <typename A, typename B, typename D>
  ArrayWrap(A x, B y, D z) : m_(x, y, z) {}
  explicit ArrayWrap(const Matrix<T, R, C> &m) : m_(m) {}
  T &operator[](int i) { return m_[i]; }
  const T &operator[](int i) const { return m_[i]; }
  T &operator()(int i) { return m_[i]; }
  const T &operator()(int i) const { return m_[i]; }
  ArrayWrap operator+(const ArrayWrap &o) const { return ArrayWrap(m_ + o.m_); }
  ArrayWrap operator-(const ArrayWrap &o) const { return ArrayWrap(m_ - o.m_); }
  ArrayWrap operator*(const ArrayWrap &o) const {
    ArrayWrap r;
    for (int i = 0; i < R * C; ++i) r[i] = m_[i] * o.m_[i];
    return r;
  }
  ArrayWrap operator/(const ArrayWrap &o) const {
    ArrayWrap r;
    for (int i = 0; i < R * C; ++i) r[i] = m_[i] / o.m_[i];
    return r;
  }
  ArrayWrap operator*(T s) const { return ArrayWrap(m_ * s); }
  friend ArrayWrap operator*(T s, const ArrayWrap &a) { return ArrayWrap(s * a.m_); }
  ArrayWrap &operator+=(const ArrayWrap &o) {
    m_ += o.m_;
    return *this;
  }
  ArrayWrap &operator-=(const ArrayWrap &o) {
    m_ -= o.m_;
    return *this;
  }
  ArrayWrap inverse() const {
    ArrayWrap r;
    for (int i = 0; i < R * C; ++i) r[i] = T(1) / m_[i];
    return r;
  }
  const Matrix<T, R, C> &matrix() const { return m_; }
  operator Matrix<T, R, C>() const { return m_; }
  template <typename U>
  ArrayWrap<U, R, C> cast() const {
    return ArrayWrap<U, R, C>(m_.template cast<U>());
  }

 private:
  Matrix<T, R, C> m_;
};

template <typename T, int R, int C>
ArrayWrap<T, R, C> Matrix<T, R, C>::array() const {
  return ArrayWrap<T, R, C>(*this);
}

template <typename T, int R, int C>
Matrix<T, R, C> operator+(const Matrix<T, R, C> &a, const ArrayWrap<T, R, C> &b) {
  return a + b.matrix();
}

typedef Matrix<float, 3, 1> Vector3f;
typedef Matrix<double, 3, 1> Vector3d;
typedef Matrix<int, 3, 1> Vector3i;
typedef Matrix<float, 4, 1> Vector4f;
typedef Matrix<double, 4, 1> Vector4d;
typedef Matrix<float, 3, 3> Matrix3f;
typedef Matrix<double, 3, 3> Matrix3d;
typedef Matrix<float, 4, 4> Matrix4f;
typedef Matrix<double, 4, 4> Matrix4d;
typedef ArrayWrap<float, 3, 1> Array3f;

// Writable view of 3 consecutive scalars inside a point struct (Eigen::Map<Vector3f>).
template <typename T, bool ARRAY = false>
class Map3 {
 public:
  typedef Matrix<T, 3, 1> Vec;
  explicit Map3(T *p) : p_(p) {}
  Map3 &operator=(const Vec &v) {
    p_[0] = v[0];
    p_[1] = v[1];
    p_[2] = v[2];
    return *this;
  }
  Map3 &operator=(const Map3 &o) {
    const T a = o.p_[0], b = o.p_[1], c = o.p_[2];
    p_[0] = a;
    p_[1] = b;
    p_[2] = c;
    return *this;
  }
  template <bool A2>
  Map3 &operator=(const Map3<const T, A2> &o) {
    return *this = o.eval();
  }
  Map3 &operator=(const ArrayWrap<T, 3, 1> &a) { return *this = a.matrix(); }
  Vec eval() const { return Vec(p_[0], p_[1], p_[2]); }
  operator Vec() const { return eval(); }
  operator ArrayWrap<T, 3, 1>() const { return ArrayWrap<T, 3, 1>(eval()); }
  T &operator[](int i) { return p_[i]; }
  T operator[](int i) const { return p_[i]; }
  T &operator()(int i) { return p_[i]; }
  T operator()(int i) const { return p_[i]; }
  Vec normalized() const { return eval().normalized(); }
  T norm() const { return eval().norm(); }
  T dot(const Vec &o) const { return eval().dot(o); }
  Vec operator+(const Vec &o) const { return eval() + o; }
  Vec operator-(const Vec &o) const { return eval() - o; }
  Vec operator*(T s) const { return eval() * s; }
  template <typename U>
  Matrix<U, 3, 1> cast() const {
    return eval().template cast<U>();
  }

 private:
  T *p_;
};

// Read-only view.
template <typename T, bool ARRAY>
class Map3<const T, ARRAY> {
 public:
  typedef Matrix<T, 3, 1> Vec;
  explicit Map3(const T *p) : p_(p) {}
  Vec eval() const { return Vec(p_[0], p_[1], p_[2]); }
  operator Vec() const { return eval(); }
  operator ArrayWrap<T, 3, 1>() const { return ArrayWrap<T, 3, 1>(eval()); }
  T operator[](int i) const { return p_[i]; }
  T operator()(int i) const { return p_[i]; }
  Vec normalized() const { return eval().normalized(); }
  T norm() const { return eval().norm(); }
  Vec operator+(const Vec &o) const { return eval() + o; }
  Vec operator-(const Vec &o) const { return eval() - o; }
  Vec operator*(T s) const { return eval() * s; }
  template <typename U>
  Matrix<U, 3, 1> cast() const {
    return eval().template cast<U>();
  }

 private:
  const T *p_;
};

enum TransformTraits { Isometry = 1, Affine = 2, AffineCompact = 0x10 | Affine, Projective = 0x20 };

// Transform<T,3,Affine>: 4x4 with last row (0,0,0,1).
template <typename T>
class Affine3 {
 public:
  typedef Matrix<T, 4, 4> MatrixType;
  typedef Matrix<T, 3, 3> LinearMatrixType;
  typedef Matrix<T, 3, 1> VectorType;
  Affine3() : m_(MatrixType::Identity()) {}
  Affine3(const MatrixType &m) : m_(m) {}  // NOLINT: Eigen allows Affine3d = Matrix4d
  static Affine3 Identity() { return Affine3(); }
  Affine3 &operator=(const MatrixType &m) {
    m_ = m;
    return *this;
  }
  MatrixType &matrix() { return m_; }
  const MatrixType &matrix() const { return m_; }
  T &operator()(int r, int c) { return m_(r, c); }
  const T &operator()(int r, int c) const { return m_(r, c); }
  LinearMatrixType linear() const { return m_.template block<3, 3>(0, 0); }
  // [deviation] real Eigen extracts the rotation with an SVD for Affine mode
  LinearMatrixType rotation() const { return linear(); }
  VectorType translation() const { return VectorType(m_(0, 3), m_(1, 3), m_(2, 3)); }
  void setLinear(const LinearMatrixType &l) {
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) m_(r, c) = l(r, c);
  }
  void setTranslation(const VectorType &t) {
    for (int r = 0; r < 3; ++r) m_(r, 3) = t[r];
  }
  template <typename U>
  Affine3<U> cast() const {
    return Affine3<U>(m_.template cast<U>());
  }
  // [Eigen-recall] Transform::inverse(Affine)
  Affine3 inverse(TransformTraits = Affine) const {
    const LinearMatrixType li = linear().inverse();
    const VectorType t = (-li) * translation();
    Affine3 r;
    r.setLinear(li);
    r.setTranslation(t);
    return r;
  }
  // [Eigen-recall] Affine * vector = linear * v + translation
  VectorType operator*(const VectorType &v) const { return linear() * v + translation(); }
  template <bool A>
  VectorType operator*(const Map3<T, A> &v) const {
    return (*this) * v.eval();
  }
  template <bool A>
  VectorType operator*(const Map3<const T, A> &v) const {
    return (*this) * v.eval();
  }
  Affine3 operator*(const Affine3 &o) const {
    Affine3 r;
    r.setLinear(linear() * o.linear());
    r.setTranslation(linear() * o.translation() + translation());
    return r;
  }
  MatrixType operator*(const MatrixType &o) const { return m_ * o; }

 private:
  MatrixType m_;
};

typedef Affine3<float> Affine3f;
typedef Affine3<double> Affine3d;

}  // namespace Eigen
