// The state ring of x::Ekf wrapping while the covariance is resident on the device: buffer_sz - 1 IMU steps without a
// vision update (before the first frame, during a tracking dropout).  The reference's enqueueInPlace() overwrites the
// oldest state (state_buffer.cpp); the mirror does the same and moves the device covariance one slot on first.  Checked
// against the same IMU stream through the reference-semantics mode (every State owns its covariance):
//   usage: xk_ring_wrap_example [n_steps] [buffer_sz]      prints "OK <max rel diff>" or "FAIL ..."
//          xk_ring_wrap_example during_update [buffer_sz]  the ring wraps onto the resident covariance WHILE an update is in
//                                                           flight on it (the IMU thread outruns a slow update): that update
//                                                           is discarded as in the reference (ekf.cpp:229-239)
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <stdexcept>
#include <string>

#include "x/ekf/ekf.h"
#include "x/vio/vio_updater.h"

using namespace x;

static Matrix tail_cov(bool resident, int n_steps, int bsz, int N, bool *old_idx_refused) {
  VioUpdater updater(0, N, 0, 8, 1e-3);
  Propagator prop(Vector3(0, 0, -9.81), ImuNoise());
  Propagator::acknowledgeModelProcessNoise();
  prop.setEngine(updater.engine());
  Ekf ekf(updater);
  ekf.set(bsz, State(N, 0), &prop, 0.0025);
  ekf.setResident(resident);
  State s0(N, 0);
  const int n = s0.nErrorStates();
  s0.cov_.resize(n, n);
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) s0.cov_(i, j) = (i == j ? 1e-2 * (1 + i % 7) : 1e-4 * std::cos(0.37 * (i + 1) * (j + 1)) * ((i + j) % 3 == 0));
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < i; ++j) s0.cov_(i, j) = s0.cov_(j, i);
  s0.time_ = 1.0;
  ekf.initializeFromState(s0);
  for (int k = 0; k <= n_steps; ++k) {
    const double t = 1.0 + 0.005 * k;
    const Vector3 w(0.2 * std::sin(3 * t), -0.1 * std::cos(2 * t), 0.05), a(0.3 * std::sin(t), 0.2, 9.81 + 0.1 * std::cos(5 * t));
    ekf.processImu(t, (unsigned)k, w, a);            // (the first call only records the measurement)
  }
  if (resident && old_idx_refused) {
    // a slot the ring never filled (only before the first wrap; afterwards the resident covariance belongs to the oldest
    // state and every slot up to the tail has one) holds no covariance: covarianceAt must say so, not wrap around
    *old_idx_refused = false;
    if (n_steps < bsz - 1) {
      try { (void)ekf.covarianceAt((n_steps + 1) % bsz); } catch (const std::out_of_range &) { *old_idx_refused = true; }
    }
  }
  return ekf.covarianceAt(-1);
}

// An updater whose update is "slow": while it runs (preProcess is the first thing Updater::update calls, without the Ekf's
// mutex) the IMU callback delivers a whole ring of samples.
struct SlowUpdater : VioUpdater {
  using VioUpdater::VioUpdater;
  Ekf *ekf = nullptr;
  int burst = 0;
  double t0 = 0;
  void preProcess(const State &) override {
    for (int k = 1; k <= burst; ++k) ekf->processImu(t0 + 0.005 * k, 1000u + k, Vector3(0.1, 0, 0), Vector3(0, 0.1, 9.81));
  }
};

// One update lapped by the IMU thread.  Returns the tail covariance afterwards; *discarded = the update came back as nullopt;
// *threw = Ekf::processImu refused (resident mode, no saved prior: more than kWrapMargin samples during one update).
static Matrix lapped_update(bool resident, int bsz, bool *discarded, bool *threw) {
  const int N = 4;
  SlowUpdater updater(0, N, 0, 8, 1e-3);
  Propagator prop(Vector3(0, 0, -9.81), ImuNoise());
  Propagator::acknowledgeModelProcessNoise();
  prop.setEngine(updater.engine());
  Ekf ekf(updater);
  ekf.set(bsz, State(N, 0), &prop, 0.0025);
  ekf.setResident(resident);
  State s0(N, 0);
  const int n = s0.nErrorStates();
  s0.cov_.resize(n, n);
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) s0.cov_(i, j) = i == j ? 1e-2 * (1 + i % 5) : 0.0;
  s0.time_ = 1.0;
  ekf.initializeFromState(s0);
  for (int k = 0; k <= 2; ++k) ekf.processImu(1.0 + 0.005 * k, (unsigned)k, Vector3(0.1, 0, 0), Vector3(0, 0.1, 9.81));
  VioMeasurement meas;
  meas.timestamp = 1.0 + 0.005 * 2;
  Track t;
  t.emplace_back(0.01, 0.02); t.emplace_back(0.011, 0.021);
  meas.msckf_tracks.push_back(t);
  updater.setWindow(2, {});
  updater.setMeasurement(meas);
  updater.ekf = &ekf; updater.burst = bsz + 1; updater.t0 = meas.timestamp;
  *discarded = *threw = false;
  try {
    *discarded = !ekf.processUpdateMeasurement().has_value();
  } catch (const std::runtime_error &e) {
    *threw = true;
    printf("threw: %s\n", e.what());
  }
  // the IMU stream goes on either way
  updater.burst = 0;
  ekf.processImu(meas.timestamp + 0.005 * (bsz + 2), 5000u, Vector3(0.1, 0, 0), Vector3(0, 0.1, 9.81));
  return *threw ? Matrix() : ekf.covarianceAt(-1);
}

// The ring wraps onto the slot of the update in flight.  The reference overwrites the slot and loses that update (ekf.cpp:229-239);
// so must the mirror with a resident covariance -- same nullopt, same covariances afterwards as the reference-semantics mode, in
// which every State owns its covariance.  A ring with more than kWrapMargin (32) free slots at the start of the update saves no
// prior: being lapped THEN (33+ IMU samples during one update) is the one case that throws.
static int wrap_during_update(int bsz) {
  bool d_res, t_res, d_ref, t_ref;
  const Matrix a = lapped_update(true, bsz, &d_res, &t_res);
  if (bsz - 1 >= 32) {
    if (t_res) { printf("OK threw (no saved prior at %d free slots)\n", bsz - 1); return 0; }
    printf("FAIL no exception although no prior was saved\n");
    return 1;
  }
  const Matrix b = lapped_update(false, bsz, &d_ref, &t_ref);
  if (t_res || t_ref) { printf("FAIL threw (resident %d, reference semantics %d)\n", (int)t_res, (int)t_ref); return 1; }
  if (!d_res || !d_ref) { printf("FAIL the lapped update was not discarded (resident %d, reference semantics %d)\n", (int)d_res, (int)d_ref); return 1; }
  double num = 0, den = 0;
  for (size_t i = 0; i < a.size(); ++i) { num += (a.data()[i] - b.data()[i]) * (a.data()[i] - b.data()[i]); den += b.data()[i] * b.data()[i]; }
  const double relv = std::sqrt(num / den);
  if (!(relv <= 1e-12)) { printf("FAIL rel=%.3e between the resident covariance and the reference semantics after a discarded update\n", relv); return 1; }
  printf("OK discarded, rel %.3e\n", relv);
  return 0;
}

int main(int argc, char **argv) {
  if (argc > 1 && std::string(argv[1]) == "during_update") {
    try { return wrap_during_update(argc > 2 ? atoi(argv[2]) : 6); }
    catch (const std::exception &e) { printf("FAIL exception outside the update: %s\n", e.what()); return 1; }
  }
  const int n_steps = argc > 1 ? atoi(argv[1]) : 23, bsz = argc > 2 ? atoi(argv[2]) : 6, N = 4;
  try {
    bool refused = false;
    const Matrix a = tail_cov(true, n_steps, bsz, N, &refused), b = tail_cov(false, n_steps, bsz, N, nullptr);
    double num = 0, den = 0;
    for (size_t i = 0; i < a.size(); ++i) { num += (a.data()[i] - b.data()[i]) * (a.data()[i] - b.data()[i]); den += b.data()[i] * b.data()[i]; }
    const double relv = std::sqrt(num / den);
    const bool wrapped = n_steps >= bsz, expect_refusal = n_steps < bsz - 1;
    if (relv <= 1e-12 && refused == expect_refusal) { printf("OK %.3e wrapped=%d refused_old=%d\n", relv, (int)wrapped, (int)refused); return 0; }
    printf("FAIL rel=%.3e refused_old=%d\n", relv, (int)refused);
    return 1;
  } catch (const std::exception &e) {
    printf("FAIL exception: %s\n", e.what());
    return 1;
  }
}
