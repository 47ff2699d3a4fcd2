// Mirror of src/x/ekf/propagator.cpp.  Small fixed-size algebra is spelled out (no Eigen in the mirror); the n x n
// covariance propagation goes through the C ABI to the device.
#include "x/ekf/propagator.h"

#include <atomic>
#include <cstdio>

#include <cmath>
#include <stdexcept>
#include <string>

#include "xk.h"

using namespace x;

namespace {
void check(xk_handle *h, int rc, const char *what) {
  if (rc != XK_OK) throw std::runtime_error(std::string(what) + ": " + xk_strerror(rc) + " (" + (h ? xk_last_error(h) : "") + ")");
}
struct M3 { double a[9]; };                                   // row-major 3 x 3
M3 eye3() { return {{1, 0, 0, 0, 1, 0, 0, 0, 1}}; }
M3 cross(const Vector3 &v) { return {{0, -v(2), v(1), v(2), 0, -v(0), -v(1), v(0), 0}}; }   // toCrossMatrix
M3 mul(const M3 &x, const M3 &y) {
  M3 r;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) r.a[3 * i + j] = x.a[3 * i] * y.a[j] + x.a[3 * i + 1] * y.a[3 + j] + x.a[3 * i + 2] * y.a[6 + j];
  return r;
}
M3 lin(double ca, const M3 &a, double cb, const M3 &b, double cc, const M3 &c) {   // ca a + cb b + cc c
  M3 r;
  for (int k = 0; k < 9; ++k) r.a[k] = ca * a.a[k] + cb * b.a[k] + cc * c.a[k];
  return r;
}
M3 scale(double s, const M3 &a) { M3 r; for (int k = 0; k < 9; ++k) r.a[k] = s * a.a[k]; return r; }
void put(CoreCovMatrix &f, int r0, int c0, const M3 &b) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) f(r0 + i, c0 + j) = b.a[3 * i + j];
}
void omega(const Vector3 &w, double o[16]) {                   // toOmegaMatrix, row-major 4 x 4
  const double x = w(0), y = w(1), z = w(2);
  const double m[16] = {0, z, -y, x, -z, 0, x, y, y, -x, 0, z, -x, -y, -z, 0};
  for (int k = 0; k < 16; ++k) o[k] = m[k];
}
void mul44(const double *a, const double *b, double *c) {
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      double s = 0;
      for (int k = 0; k < 4; ++k) s += a[4 * i + k] * b[4 * k + j];
      c[4 * i + j] = s;
    }
}
}  // namespace

void Propagator::quaternionIntegrator(const Vector3 &e_w_0, const Vector3 &e_w_1, double dt, double out[16]) const {
  double om1[16], om0[16], a[16], a_k[16], tmp[16], t2[16];
  omega(e_w_1, om1);
  omega(e_w_0, om0);
  const Vector3 mean((e_w_1(0) + e_w_0(0)) / 2.0, (e_w_1(1) + e_w_0(1)) / 2.0, (e_w_1(2) + e_w_0(2)) / 2.0);
  omega(mean, a);
  for (int k = 0; k < 16; ++k) { a[k] = a[k] * 0.5 * dt; a_k[k] = a[k]; out[k] = (k % 5 == 0) ? 1.0 : 0.0; }
  int fac = 1;
  for (int k = 1; k < 5; k++) {                                 // Taylor series of the matrix exponential, 4th order
    fac = fac * k;
    for (int i = 0; i < 16; ++i) out[i] = out[i] + a_k[i] / fac;
    mul44(a_k, a, tmp);
    for (int i = 0; i < 16; ++i) a_k[i] = tmp[i];
  }
  mul44(om1, om0, tmp);
  mul44(om0, om1, t2);
  for (int i = 0; i < 16; ++i) out[i] = out[i] + 1.0 / 48.0 * (tmp[i] - t2[i]) * dt * dt;
}

void Propagator::propagateState(const State &state_0, State &state_1) const {   // propagator.cpp:30-51
  state_1.setStaticStatesFrom(state_0);
  Vector3 e_w_1, e_a_1, e_w_0, e_a_0;
  state_1.computeUnbiasedImuMeasurements(e_w_1, e_a_1);
  state_0.computeUnbiasedImuMeasurements(e_w_0, e_a_0);
  const double dt = state_1.time_ - state_0.time_;
  double dq[16];
  quaternionIntegrator(e_w_0, e_w_1, dt, dq);
  const double c0[4] = {state_0.q_.x(), state_0.q_.y(), state_0.q_.z(), state_0.q_.w()};   // coeffs(): x, y, z, w
  double c1[4];
  for (int i = 0; i < 4; ++i) c1[i] = dq[4 * i] * c0[0] + dq[4 * i + 1] * c0[1] + dq[4 * i + 2] * c0[2] + dq[4 * i + 3] * c0[3];
  state_1.q_ = Quaternion(c1[3], c1[0], c1[1], c1[2]);
  state_1.q_.normalize();
  double r1[9], r0[9];
  state_1.q_.toRotationMatrix(r1);
  state_0.q_.toRotationMatrix(r0);
  for (int i = 0; i < 3; ++i) {
    const double dv = (r1[3 * i] * e_a_1(0) + r1[3 * i + 1] * e_a_1(1) + r1[3 * i + 2] * e_a_1(2) +
                       r0[3 * i] * e_a_0(0) + r0[3 * i + 1] * e_a_0(1) + r0[3 * i + 2] * e_a_0(2)) / 2.0;
    state_1.v_(i) = state_0.v_(i) + (dv + g_(i)) * dt;
  }
  for (int i = 0; i < 3; ++i) state_1.p_(i) = state_0.p_(i) + (state_1.v_(i) + state_0.v_(i)) / 2.0 * dt;
}

CoreCovMatrix Propagator::discreteStateTransition(double dt, const Vector3 &e_w, const Vector3 &e_a, const Quaternion &q) const {
  const M3 w_x = cross(e_w), a_x = cross(e_a), I = eye3();
  M3 c_q;
  q.toRotationMatrix(c_q.a);
  const double dt_2_f2 = dt * dt * 0.5;
  const double dt_3_f3 = dt_2_f2 * dt / 3.0;
  const double dt_4_f4 = dt_3_f3 * dt * 0.25;
  const double dt_5_f5 = dt_4_f4 * dt * 0.2;
  const M3 c_q_a_x = mul(c_q, a_x), w2 = mul(w_x, w_x);
  const M3 a = mul(c_q_a_x, lin(-dt_2_f2, I, dt_3_f3, w_x, -dt_4_f4, w2));
  const M3 b = mul(c_q_a_x, lin(dt_3_f3, I, -dt_4_f4, w_x, dt_5_f5, w2));
  const M3 d = scale(-1.0, a);
  const M3 e = lin(1.0, I, -dt, w_x, dt_2_f2, w2);
  const M3 f = lin(-dt, I, dt_2_f2, w_x, -dt_3_f3, w2);
  const M3 c = mul(c_q_a_x, f);
  CoreCovMatrix f_d = CoreCovMatrix::Identity();
  put(f_d, kIdxP, kIdxV, scale(dt, I));
  put(f_d, kIdxP, kIdxQ, a);
  put(f_d, kIdxP, kIdxBw, b);
  put(f_d, kIdxP, kIdxBa, scale(-dt_2_f2, c_q));
  put(f_d, kIdxV, kIdxQ, c);
  put(f_d, kIdxV, kIdxBw, d);
  put(f_d, kIdxV, kIdxBa, scale(-dt, c_q));
  put(f_d, kIdxQ, kIdxQ, e);
  put(f_d, kIdxQ, kIdxBw, f);
  return f_d;
}

static std::atomic<bool> g_model_qd_acknowledged{false};
void Propagator::acknowledgeModelProcessNoise(bool on) { g_model_qd_acknowledged.store(on); }

CoreCovMatrix Propagator::discreteProcessNoiseCov(double dt, const Quaternion &q, const Vector3 &e_w, const Vector3 &e_a,
                                                  double n_w, double n_bw, double n_a, double n_ba) const {
  if (!g_model_qd_acknowledged.exchange(true))
    fprintf(stderr, "x::Propagator: discreteProcessNoiseCov is the mirror's clean model (exact integral of F_d G Q_c G^T F_d^T), NOT the "
                    "reference's generated q_d (propagator.cpp:207-840): the propagated covariance differs from the upstream filter's. "
                    "A drop-in keeps the reference's function (override the virtual or setProcessNoiseFunction; INTEGRATION.md 3.6); "
                    "Propagator::acknowledgeModelProcessNoise() silences this.\n");
  // G Q_c G^T for G = [v: -C(q), theta: -I, b_w: I, b_a: I] is block diagonal and does not depend on q:
  // diag(0, n_a^2 I, n_w^2 I, n_bw^2 I, n_ba^2 I).  The integrand F_d(t) (.) F_d(t)^T is a polynomial in t.
  // (degree <= 10 in t: six Gauss-Legendre points are exact)
  static const double gx[6] = {-0.9324695142031521, -0.6612093864662645, -0.2386191860831969,
                               0.2386191860831969,  0.6612093864662645,  0.9324695142031521};
  static const double gw[6] = {0.1713244923791704, 0.3607615730481386, 0.4679139345726910,
                               0.4679139345726910, 0.3607615730481386, 0.1713244923791704};
  double dg[15];
  for (int i = 0; i < 3; ++i) { dg[i] = 0.0; dg[3 + i] = n_a * n_a; dg[6 + i] = n_w * n_w; dg[9 + i] = n_bw * n_bw; dg[12 + i] = n_ba * n_ba; }
  CoreCovMatrix Q = CoreCovMatrix::Zero();
  // F_d(t) is the identity in the bias rows and zero left of the diagonal blocks: only the structurally non-zero terms
  // of  sum_l F(i,l) dg[l] F(j,l)  are formed (the skipped ones are exact zeros, so the sums are the dense sums)
  for (int k = 0; k < 6; ++k) {
    const double t = 0.5 * dt * (gx[k] + 1.0), wk = 0.5 * dt * gw[k];
    const CoreCovMatrix F = discreteStateTransition(t, e_w, e_a, q);
    double Fd[9][16], Fr[9][16];                                       // rows of F, contiguous in l (F itself is column-major)
    for (int i = 0; i < 9; ++i)
      for (int l = 3; l < 15; ++l) { Fr[i][l] = F(i, l); Fd[i][l] = Fr[i][l] * dg[l]; }
    for (int i = 0; i < 9; ++i) {
      for (int j = i; j < 9; ++j) {
        const int lo = (j >= 6) ? 6 : 3, hi = (j >= 6) ? 12 : 15;   // attitude rows: columns theta, b_w only
        double s = 0.0;
        for (int l = lo; l < hi; ++l) s += Fd[i][l] * Fr[j][l];
        Q(i, j) += wk * s;
      }
      for (int j = 9; j < 15; ++j) Q(i, j) += wk * Fd[i][j];          // F(j, :) = e_j for the bias rows
    }
    for (int i = 9; i < 15; ++i) Q(i, i) += wk * dg[i];
  }
  for (int i = 0; i < 15; ++i)
    for (int j = 0; j < i; ++j) Q(i, j) = Q(j, i);
  return Q;
}

void Propagator::transition(const State &state_0, const State &state_1, CoreCovMatrix &f_d, CoreCovMatrix &q_d) const {
  Vector3 e_w_1, e_a_1;                                          // propagator.cpp:55-68 (only the state_1 measurements enter)
  state_1.computeUnbiasedImuMeasurements(e_w_1, e_a_1);
  const double dt = state_1.time_ - state_0.time_;
  f_d = discreteStateTransition(dt, e_w_1, e_a_1, state_1.q_);
  q_d = processNoise(dt, state_1.q_, e_w_1, e_a_1);
}

void Propagator::propagateCovariance(const State &state_0, State &state_1) const {   // propagator.cpp:53-71
  if (!xk_) throw std::runtime_error("Propagator::propagateCovariance: no device engine (setEngine)");
  CoreCovMatrix f_d, q_d;
  transition(state_0, state_1, f_d, q_d);
  const int n = state_0.cov_.rows();
  check(xk_, xk_upload_P(xk_, state_0.cov_.data(), n, n), "xk_upload_P");
  check(xk_, xk_cov_propagate(xk_, f_d.m, 15, q_d.m, 15), "xk_cov_propagate");
  if (state_1.cov_.rows() != n) state_1.cov_.resize(n, n);
  check(xk_, xk_download_P(xk_, state_1.cov_.data(), n, n), "xk_download_P");
}
