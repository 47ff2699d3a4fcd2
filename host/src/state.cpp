// Mirror of src/x/ekf/state.cpp (reference) -- only what the update path touches.
#include "x/ekf/state.h"

using namespace x;

State::State(int n_poses, int n_features) {
  p_array_ = Matrix::Zero(n_poses * 3, 1);
  q_array_ = Matrix::Zero(n_poses * 4, 1);
  f_array_ = Matrix::Zero(n_features * 3, 1);
  const int n = kSizeCoreErr + n_poses * 6 + n_features * 3;
  cov_ = Matrix::Identity(n, n);
}

void State::getDynamicStates(double o[16]) const {
  for (int i = 0; i < 3; ++i) { o[i] = p_(i); o[3 + i] = v_(i); o[10 + i] = b_w_(i); o[13 + i] = b_a_(i); }
  o[6] = q_.x(); o[7] = q_.y(); o[8] = q_.z(); o[9] = q_.w();
}

Attitude State::computeCameraAttitude() const {
  const Quaternion q = q_.normalized() * q_ic_.normalized();
  return {q.x(), q.y(), q.z(), q.w()};
}

Vector3 State::computeCameraPosition() const {
  double r[9];
  q_.normalized().toRotationMatrix(r);
  return Vector3(p_(0) + r[0] * p_ic_(0) + r[1] * p_ic_(1) + r[2] * p_ic_(2),
                 p_(1) + r[3] * p_ic_(0) + r[4] * p_ic_(1) + r[5] * p_ic_(2),
                 p_(2) + r[6] * p_ic_(0) + r[7] * p_ic_(1) + r[8] * p_ic_(2));
}

Quaternion State::errorQuatFromSmallAngles(const double d[3]) {
  const double n = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
  if (n == 0.0) return Quaternion(1, 0, 0, 0);                      // state.cpp:274-275
  const double s = std::sin(0.5 * n), c = std::cos(0.5 * n);        // AngleAxisd(angle, axis) -> quaternion
  return Quaternion(c, d[0] / n * s, d[1] / n * s, d[2] / n * s);
}

void State::correct(const Vectorx &corr) {
  const int n_p = p_array_.rows(), n_f = f_array_.rows();
  for (int i = 0; i < 3; ++i) {
    p_(i) += corr(kIdxP + i);
    v_(i) += corr(kIdxV + i);
    b_w_(i) += corr(kIdxBw + i);
    b_a_(i) += corr(kIdxBa + i);
  }
  for (int i = 0; i < n_p; ++i) p_array_(i) += corr(kSizeCoreErr + i);
  for (int i = 0; i < n_f; ++i) f_array_(i) += corr(kSizeCoreErr + 2 * n_p + i);
  const double dth[3] = {corr(kIdxQ), corr(kIdxQ + 1), corr(kIdxQ + 2)};
  q_ = (q_ * errorQuatFromSmallAngles(dth)).normalized();
  for (int i = 0; i < n_p / 3; ++i) {
    const double d[3] = {corr(kSizeCoreErr + n_p + 3 * i), corr(kSizeCoreErr + n_p + 3 * i + 1),
                         corr(kSizeCoreErr + n_p + 3 * i + 2)};
    Quaternion qi(q_array_(4 * i + 3), q_array_(4 * i), q_array_(4 * i + 1), q_array_(4 * i + 2));
    qi = qi * errorQuatFromSmallAngles(d);
    qi.normalize();
    q_array_(4 * i) = qi.x(); q_array_(4 * i + 1) = qi.y(); q_array_(4 * i + 2) = qi.z(); q_array_(4 * i + 3) = qi.w();
  }
}
