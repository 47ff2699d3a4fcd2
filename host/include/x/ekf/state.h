// x/ekf/state.h -- mirror of x::State (include/x/ekf/state.h:36-338, src/x/ekf/state.cpp).
#pragma once
#include "x/common/types.h"
#include "x/vision/types.h"

namespace x {
class State {
 public:
  State() = default;
  State(int n_poses, int n_features);                                   // state.cpp:23-38
  static constexpr double kInvalid = -1.0;

  double getTime() const { return time_; }
  void setTime(double t) { time_ = t; }
  const Matrix &getCovariance() const { return cov_; }
  Matrix &getCovarianceRef() { return cov_; }                            // state.cpp:125
  void setCovariance(const Matrix &c) { cov_ = c; }
  const Matrix &getPositionArray() const { return p_array_; }
  const Matrix &getOrientationArray() const { return q_array_; }        // x,y,z,w per pose (state.cpp:235-240)
  const Matrix &getFeatureArray() const { return f_array_; }
  void setPositionArray(const Matrix &m) { p_array_ = m; }
  void setOrientationArray(const Matrix &m) { q_array_ = m; }
  void setFeatureArray(const Matrix &m) { f_array_ = m; }
  int nPosesMax() const { return p_array_.rows() / 3; }                   // state.cpp:165
  int nFeaturesMax() const { return f_array_.rows() / 3; }
  int nErrorStates() const {                                              // state.cpp:171-175
    return kSizeCoreErr + p_array_.rows() + (q_array_.rows() / 4) * 3 + f_array_.rows();
  }
  // p, v, q(xyzw), b_w, b_a (State::getDynamicStates, state.cpp:87-99)
  void getDynamicStates(double out16[16]) const;
  Attitude computeCameraAttitude() const;                                 // state.cpp:184-187
  Vector3 computeCameraPosition() const;                                  // state.cpp:189-191
  // additive on p,v,b_w,b_a,p_array,f_array; q <- (q * dq(dtheta)).normalized() (state.cpp:197-249)
  void correct(const Vectorx &correction);
  void setImu(double time, unsigned int seq, const Vector3 &w_m, const Vector3 &a_m) {   // state.cpp:145-151
    time_ = time; seq_ = seq; w_m_ = w_m; a_m_ = a_m;
  }
  void setStaticStatesFrom(const State &s) {                              // state.cpp:153-161
    b_w_ = s.b_w_; b_a_ = s.b_a_; q_ic_ = s.q_ic_; p_ic_ = s.p_ic_;
    p_array_ = s.p_array_; q_array_ = s.q_array_; f_array_ = s.f_array_;
  }
  void computeUnbiasedImuMeasurements(Vector3 &e_w, Vector3 &e_a) const {  // state.cpp:177-182
    for (int i = 0; i < 3; ++i) { e_w(i) = w_m_(i) - b_w_(i); e_a(i) = a_m_(i) - b_a_(i); }
  }

  double time_ = kInvalid;
  unsigned int seq_ = 0;
  Vector3 p_, v_;
  Quaternion q_;
  Vector3 b_w_, b_a_;
  Matrix p_array_, q_array_, f_array_;
  Matrix cov_;
  Quaternion q_ic_;
  Vector3 p_ic_;
  Vector3 w_m_, a_m_;

 private:
  static Quaternion errorQuatFromSmallAngles(const double dtheta[3]);    // state.cpp:273-283
};
}  // namespace x
