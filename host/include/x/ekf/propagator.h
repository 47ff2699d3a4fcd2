// x/ekf/propagator.h -- mirror of x::Propagator (include/x/ekf/propagator.h, src/x/ekf/propagator.cpp): IMU state
// propagation and the two 15 x 15 closed forms of the covariance propagation on the host; the n x n part
// (propagateCovarianceMatrices, propagator.cpp:166-205) runs on the covariance resident in HBM (xk_cov_propagate).
#pragma once
#include "x/ekf/state.h"

struct xk_handle;

namespace x {
struct ImuNoise {          // include/x/ekf/propagator.h (continuous-time standard deviations)
  double n_w = 0.0013, n_bw = 0.00013, n_a = 0.0083, n_ba = 0.00083;
};
struct CoreCovMatrix {     // Eigen::Matrix<double, 15, 15>, column-major
  double m[15 * 15];
  double &operator()(int i, int j) { return m[i + 15 * j]; }
  double operator()(int i, int j) const { return m[i + 15 * j]; }
  static CoreCovMatrix Identity() { CoreCovMatrix c; for (int k = 0; k < 225; ++k) c.m[k] = 0; for (int i = 0; i < 15; ++i) c(i, i) = 1; return c; }
  static CoreCovMatrix Zero() { CoreCovMatrix c; for (int k = 0; k < 225; ++k) c.m[k] = 0; return c; }
};

class Propagator {
 public:
  Propagator() = default;
  Propagator(const Vector3 &g, const ImuNoise &imu_noise) : g_(g), imu_noise_(imu_noise) {}
  virtual ~Propagator() = default;
  void set(const Vector3 &g, const ImuNoise &imu_noise) { g_ = g; imu_noise_ = imu_noise; }   // propagator.cpp:25-28
  void setEngine(xk_handle *xk) { xk_ = xk; }

  virtual void propagateState(const State &state_0, State &state_1) const;           // :30-51
  // Reference semantics (State owns the covariance): state_0.cov_ is uploaded, propagated on the device and
  // downloaded into state_1.cov_.
  virtual void propagateCovariance(const State &state_0, State &state_1) const;      // :53-71
  // The two closed forms for the step state_0 -> state_1 (what propagateCovariance hands to the device).
  void transition(const State &state_0, const State &state_1, CoreCovMatrix &f_d, CoreCovMatrix &q_d) const;

  void quaternionIntegrator(const Vector3 &e_w_0, const Vector3 &e_w_1, double dt, double out[16]) const;   // :73-97
  CoreCovMatrix discreteStateTransition(double dt, const Vector3 &e_w, const Vector3 &e_a, const Quaternion &q) const;   // :99-164
  // Discrete process noise.  The reference's version (:207-840) is 630 lines of machine-generated scalar code (147 of
  // 225 entries assigned, not symmetric); it cannot be restated without copying it and it is host-side 15 x 15
  // arithmetic, so a drop-in overrides this with the reference's own function.  The default
  // is the model that code was generated from: Q_d = int_0^dt F_d(t) G Q_c G^T F_d(t)^T dt, integrated exactly.
  virtual CoreCovMatrix discreteProcessNoiseCov(double dt, const Quaternion &q, const Vector3 &e_w, const Vector3 &e_a,
                                                double n_w, double n_bw, double n_a, double n_ba) const;

 protected:
  Vector3 g_{0.0, 0.0, -9.81};
  ImuNoise imu_noise_;
  xk_handle *xk_ = nullptr;
};
}  // namespace x
