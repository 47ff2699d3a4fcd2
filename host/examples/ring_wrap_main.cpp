// The state ring of x::Ekf wrapping while the covariance is resident on the device: buffer_sz - 1 IMU steps without a
// vision update (before the first frame, during a tracking dropout).  The reference's enqueueInPlace() overwrites the
// oldest state (state_buffer.cpp); the mirror does the same and moves the device covariance one slot on first.  Checked
// against the same IMU stream through the reference-semantics mode (every State owns its covariance):
//   usage: xk_ring_wrap_example [n_steps] [buffer_sz]      prints "OK <max rel diff>" or "FAIL ..."
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <stdexcept>

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

int main(int argc, char **argv) {
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
