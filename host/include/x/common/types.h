// x/common/types.h -- value types of the host-side mirror of the xVIO API.
// Mirrors include/x/common/types.h:30-58 of the reference (index enums, Matrix/Vector typedefs)
// without Eigen: x::Matrix is a minimal column-major dense double matrix exposing the subset of
// Eigen::MatrixXd the Updater/VioUpdater/Ekf signatures need (rows/cols/data/operator()/size).
// With Eigen available a caller can pass MatrixXd::data() straight to the C ABI (include/xk.h).
#pragma once
#include <cmath>
#include <cstddef>
#include <vector>

namespace x {

enum elementSize { pose = 3, quaternion = 4, smallAngle = 3, feature = 3, bias_w = 3, bias_a = 3, velocity = 3 };
enum { kIdxP = 0, kIdxV = 3, kIdxQ = 6, kIdxBw = 9, kIdxBa = 12, kSizeCoreErr = 15, kSizeClone = 15 };

class Matrix {
 public:
  Matrix() = default;
  Matrix(int r, int c) : r_(r), c_(c), d_((size_t)r * c, 0.0) {}
  static Matrix Zero(int r, int c) { return Matrix(r, c); }
  static Matrix Identity(int r, int c) {
    Matrix m(r, c);
    for (int i = 0; i < (r < c ? r : c); ++i) m(i, i) = 1.0;
    return m;
  }
  int rows() const { return r_; }
  int cols() const { return c_; }
  size_t size() const { return d_.size(); }
  double *data() { return d_.data(); }
  const double *data() const { return d_.data(); }
  double &operator()(int i, int j = 0) { return d_[(size_t)i + (size_t)j * r_]; }
  double operator()(int i, int j = 0) const { return d_[(size_t)i + (size_t)j * r_]; }
  void resize(int r, int c) { r_ = r; c_ = c; d_.assign((size_t)r * c, 0.0); }
  Matrix &operator+=(const Matrix &o) {
    for (size_t i = 0; i < d_.size(); ++i) d_[i] += o.d_[i];
    return *this;
  }

 private:
  int r_ = 0, c_ = 0;
  std::vector<double> d_;
};
using Vectorx = Matrix;  // n x 1

struct Vector3 {
  double v[3] = {0, 0, 0};
  Vector3() = default;
  Vector3(double x, double y, double z) : v{x, y, z} {}
  double &operator()(int i) { return v[i]; }
  double operator()(int i) const { return v[i]; }
  double norm() const { return std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); }
};

// Hamilton quaternion, coefficient order of Eigen::Quaterniond::coeffs(): x, y, z, w.
struct Quaternion {
  double x_ = 0, y_ = 0, z_ = 0, w_ = 1;
  Quaternion() = default;
  Quaternion(double w, double x, double y, double z) : x_(x), y_(y), z_(z), w_(w) {}  // Eigen ctor order
  double &x() { return x_; } double &y() { return y_; } double &z() { return z_; } double &w() { return w_; }
  double x() const { return x_; } double y() const { return y_; } double z() const { return z_; } double w() const { return w_; }
  double norm() const { return std::sqrt(x_ * x_ + y_ * y_ + z_ * z_ + w_ * w_); }
  Quaternion normalized() const { double n = norm(); return Quaternion(w_ / n, x_ / n, y_ / n, z_ / n); }
  void normalize() { *this = normalized(); }
  Quaternion operator*(const Quaternion &b) const {
    return Quaternion(w_ * b.w_ - x_ * b.x_ - y_ * b.y_ - z_ * b.z_, w_ * b.x_ + x_ * b.w_ + y_ * b.z_ - z_ * b.y_,
                      w_ * b.y_ + y_ * b.w_ + z_ * b.x_ - x_ * b.z_, w_ * b.z_ + z_ * b.w_ + x_ * b.y_ - y_ * b.x_);
  }
  // q.toRotationMatrix() (unit q), row-major 3x3
  void toRotationMatrix(double r[9]) const {
    const double tx = 2 * x_, ty = 2 * y_, tz = 2 * z_;
    const double twx = tx * w_, twy = ty * w_, twz = tz * w_, txx = tx * x_, txy = ty * x_, txz = tz * x_;
    const double tyy = ty * y_, tyz = tz * y_, tzz = tz * z_;
    r[0] = 1 - (tyy + tzz); r[1] = txy - twz; r[2] = txz + twy;
    r[3] = txy + twz; r[4] = 1 - (txx + tzz); r[5] = tyz - twx;
    r[6] = txz - twy; r[7] = tyz + twx; r[8] = 1 - (txx + tyy);
  }
};

}  // namespace x
