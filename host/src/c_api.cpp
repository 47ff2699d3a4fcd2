// Plain-C entry points into the host-only pieces of the mirror (no device work), so that the CPU test suite can check
// them against the oracle through ctypes: the IMU closed forms of x::Propagator, the SimpleState <-> payload bridge and
// SlamUpdate::computeInverseDepthsNew.
#include <stdexcept>
#include <vector>

#include "xk.h"
#include "x/ekf/propagator.h"
#include "x/ekf/updater.h"
#include "x/ekf/simple_state.h"
#include "x/vio/slam_update.h"

using namespace x;

extern "C" {
// f_d (15 x 15, column-major) of Propagator::discreteStateTransition (propagator.cpp:99-164)
void x_host_discrete_state_transition(double dt, const double *e_w, const double *e_a, const double *q_xyzw, double *out) {
  Propagator p;
  const CoreCovMatrix f = p.discreteStateTransition(dt, Vector3(e_w[0], e_w[1], e_w[2]), Vector3(e_a[0], e_a[1], e_a[2]),
                                                    Quaternion(q_xyzw[3], q_xyzw[0], q_xyzw[1], q_xyzw[2]));
  for (int k = 0; k < 225; ++k) out[k] = f.m[k];
}
// the mirror's default q_d model (see x/ekf/propagator.h)
void x_host_process_noise_model(double dt, const double *q_xyzw, const double *e_w, const double *e_a, double n_w, double n_bw,
                                double n_a, double n_ba, double *out) {
  Propagator p;
  Propagator::acknowledgeModelProcessNoise();          // (this entry point exists to look at the model)
  const CoreCovMatrix q = p.discreteProcessNoiseCov(dt, Quaternion(q_xyzw[3], q_xyzw[0], q_xyzw[1], q_xyzw[2]),
                                                    Vector3(e_w[0], e_w[1], e_w[2]), Vector3(e_a[0], e_a[1], e_a[2]), n_w, n_bw,
                                                    n_a, n_ba);
  for (int k = 0; k < 225; ++k) out[k] = q.m[k];
}
// 4 x 4 row-major (propagator.cpp:73-97)
void x_host_quaternion_integrator(const double *e_w_0, const double *e_w_1, double dt, double *out16) {
  Propagator p;
  p.quaternionIntegrator(Vector3(e_w_0[0], e_w_0[1], e_w_0[2]), Vector3(e_w_1[0], e_w_1[1], e_w_1[2]), dt, out16);
}
// s = [time, p3, v3, q4 (xyzw), b_w3, b_a3, w_m3, a_m3] (23 doubles); s1 arrives with time / w_m / a_m set
void x_host_propagate_state(const double *s0, double *s1, const double *g) {
  auto load = [](const double *s, State &st) {
    st.time_ = s[0];
    for (int i = 0; i < 3; ++i) { st.p_(i) = s[1 + i]; st.v_(i) = s[4 + i]; st.b_w_(i) = s[11 + i]; st.b_a_(i) = s[14 + i]; st.w_m_(i) = s[17 + i]; st.a_m_(i) = s[20 + i]; }
    st.q_ = Quaternion(s[10], s[7], s[8], s[9]);
  };
  State a(1, 0), b(1, 0);
  load(s0, a);
  load(s1, b);
  Propagator p(Vector3(g[0], g[1], g[2]), ImuNoise());
  p.propagateState(a, b);
  for (int i = 0; i < 3; ++i) { s1[1 + i] = b.p_(i); s1[4 + i] = b.v_(i); s1[11 + i] = b.b_w_(i); s1[14 + i] = b.b_a_(i); }
  s1[7] = b.q_.x(); s1[8] = b.q_.y(); s1[9] = b.q_.z(); s1[10] = b.q_.w();
}
// Propagator::transition (the two closed forms of one IMU step, no device work) with the discrete process noise injected through
// setProcessNoiseFunction; returns how often the injected function was called.  use_hook = 0: the default model instead
// (which announces itself on stderr once per process unless acknowledged).
int x_host_transition(const double *s0, const double *s1, const double *qd225, int use_hook, double *fd_out, double *qd_out) {
  auto load = [](const double *s, State &st) {
    st.time_ = s[0];
    for (int i = 0; i < 3; ++i) { st.p_(i) = s[1 + i]; st.v_(i) = s[4 + i]; st.b_w_(i) = s[11 + i]; st.b_a_(i) = s[14 + i]; st.w_m_(i) = s[17 + i]; st.a_m_(i) = s[20 + i]; }
    st.q_ = Quaternion(s[10], s[7], s[8], s[9]);
  };
  State a(1, 0), b(1, 0);
  load(s0, a);
  load(s1, b);
  Propagator p(Vector3(0, 0, -9.81), ImuNoise());
  int calls = 0;
  if (use_hook)
    p.setProcessNoiseFunction([&](double, const Quaternion &, const Vector3 &, const Vector3 &, double, double, double, double) {
      CoreCovMatrix q;
      for (int k = 0; k < 225; ++k) q.m[k] = qd225[k];
      ++calls;
      return q;
    });
  CoreCovMatrix f_d, q_d;
  p.transition(a, b, f_d, q_d);
  for (int k = 0; k < 225; ++k) { fd_out[k] = f_d.m[k]; qd_out[k] = q_d.m[k]; }
  return calls;
}
// One IMU step of the covariance through the mirror (Propagator::propagateCovariance: upload, xk_cov_propagate, download) with
// the discrete process noise INJECTED through setProcessNoiseFunction -- what a drop-in does with the reference's own q_d.
// s0 / s1 as above; qd225 column-major; P_in / P_out n x n column-major, n = 15 + 6 N + 3 M.  Needs a GPU (device 0).
int x_host_propagate_covariance_with_qd(const double *s0, const double *s1, const double *qd225, const double *P_in, int N, int M,
                                        double *P_out, double *fd_out, int *qd_calls) {
  auto load = [](const double *s, State &st) {
    st.time_ = s[0];
    for (int i = 0; i < 3; ++i) { st.p_(i) = s[1 + i]; st.v_(i) = s[4 + i]; st.b_w_(i) = s[11 + i]; st.b_a_(i) = s[14 + i]; st.w_m_(i) = s[17 + i]; st.a_m_(i) = s[20 + i]; }
    st.q_ = Quaternion(s[10], s[7], s[8], s[9]);
  };
  xk_handle *xk = nullptr;
  try {
    if (xk_create(0, N, M, 4, &xk) != XK_OK) return 2;
    State a(N, M), b(N, M);
    load(s0, a);
    load(s1, b);
    const int n = 15 + 6 * N + 3 * M;
    a.cov_.resize(n, n);
    for (int k = 0; k < n * n; ++k) a.cov_.data()[k] = P_in[k];
    Propagator p(Vector3(0, 0, -9.81), ImuNoise());
    p.setEngine(xk);
    int calls = 0;
    p.setProcessNoiseFunction([&](double, const Quaternion &, const Vector3 &, const Vector3 &, double, double, double, double) {
      CoreCovMatrix q;
      for (int k = 0; k < 225; ++k) q.m[k] = qd225[k];
      ++calls;
      return q;
    });
    CoreCovMatrix f_d, q_d;
    p.transition(a, b, f_d, q_d);
    for (int k = 0; k < 225; ++k) fd_out[k] = f_d.m[k];
    p.propagateCovariance(a, b);
    for (int k = 0; k < n * n; ++k) P_out[k] = b.cov_.data()[k];
    *qd_calls = calls;
    xk_destroy(xk);
    return 0;
  } catch (...) {
    if (xk) xk_destroy(xk);
    return 1;
  }
}
// Updater::update with a subclass that hands applyUpdate a DENSE h (its own rows, not VioUpdater's device-resident
// construction), covariance owned by the State (resident = 0) or resident on the device (1).  Needs a GPU (device 0).
namespace {
class DenseUpdater : public Updater {
 public:
  DenseUpdater(int N, int M, const double *H, int m, const double *res, const double *rdiag) : H_(H), m_(m), res_(res), rdiag_(rdiag) {
    if (xk_create(0, N, M, 4, &xk_) != XK_OK) throw std::runtime_error("xk_create");
  }
  ~DenseUpdater() override { xk_destroy(xk_); }
  double getTime() const override { return 0.0; }
 protected:
  void preProcess(const State &) override {}
  bool preUpdate(State &) override { return true; }
  bool preUpdateShortMsckf() override { return false; }
  bool preUpdateCI() override { return false; }
  void constructSlamCIUpdate(const State &, MatrixList &, MatrixList &, MatrixList &, MatrixList &) override {}
  void constructUpdate(const State &state, Matrix &h, Matrix &res, Matrix &r) override {
    const int n = state.nErrorStates();
    h = Matrix(m_, n); res = Matrix(m_, 1); r = Matrix::Zero(m_, m_);
    for (int k = 0; k < m_ * n; ++k) h.data()[k] = H_[k];
    for (int i = 0; i < m_; ++i) { res(i) = res_[i]; r(i, i) = rdiag_[i]; }
  }
  void constructShortMsckfUpdate(const State &, Matrix &, Matrix &, Matrix &) override {}
  void postUpdate(State &, const Matrix &) override {}
 private:
  const double *H_; int m_; const double *res_, *rdiag_;
};
}  // namespace
int x_host_dense_update(const double *P_in, int N, int M, const double *H, int m, const double *res, const double *rdiag, int resident,
                        double *P_out, double *core16_out) {
  try {
    const int n = 15 + 6 * N + 3 * M;
    DenseUpdater u(N, M, H, m, res, rdiag);
    u.setResident(resident != 0);
    State s(N, M);
    if (resident) {
      if (xk_upload_P(u.engine(), P_in, n, n) != XK_OK) return 2;
    } else {
      s.cov_.resize(n, n);
      for (int k = 0; k < n * n; ++k) s.cov_.data()[k] = P_in[k];
    }
    u.update(s);
    if (resident) { if (xk_download_P(u.engine(), P_out, n, n) != XK_OK) return 3; }
    else for (int k = 0; k < n * n; ++k) P_out[k] = s.cov_.data()[k];
    s.getDynamicStates(core16_out);
    return 0;
  } catch (...) { return 1; }
}
// SimpleState::fromPayload -> accessors -> toPayload; lists_out = [attitudes 4N | positions 3N] as the CI code reads them
int x_host_simple_state_roundtrip(const double *payload_in, int N, int M, double *payload_out, double *lists_out) {
  try {
    double id = 0, ts = 0;
    const SimpleState s = SimpleState::fromPayload(payload_in, N, M, &id, &ts);
    if (s.nPosesMax() != N || s.nFeaturesMax() != M || s.getErrorStateSize() != 15 + 6 * N + 3 * M || s.nErrorStates() != 15 + 6 * N + 3 * M) return 2;
    s.toPayload(id, ts, payload_out);
    const AttitudeList al = s.getCameraAttitudesList();
    const TranslationList tl = s.getCameraPositionsList();
    for (int i = 0; i < N; ++i) {
      lists_out[4 * i] = al[i].ax; lists_out[4 * i + 1] = al[i].ay; lists_out[4 * i + 2] = al[i].az; lists_out[4 * i + 3] = al[i].aw;
      lists_out[4 * N + 3 * i] = tl[i].tx; lists_out[4 * N + 3 * i + 1] = tl[i].ty; lists_out[4 * N + 3 * i + 2] = tl[i].tz;
    }
    return 0;
  } catch (...) { return 1; }
}
// last observations [n x 2] -> ivds [3n] (slam_update.cpp:216-242)
void x_host_inverse_depths_new(const double *last_obs, int n, double rho_0, double *ivds) {
  TrackList tl;
  for (int j = 0; j < n; ++j) { Track t; t.emplace_back(0.0, 0.0); t.emplace_back(last_obs[2 * j], last_obs[2 * j + 1]); tl.push_back(t); }
  Matrix out;
  SlamUpdate::computeInverseDepthsNew(tl, rho_0, out);
  for (int i = 0; i < 3 * n; ++i) ivds[i] = out(i);
}
}
