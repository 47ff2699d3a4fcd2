// Mirror of src/x/ekf/ekf.cpp + state_buffer.cpp.
#include "x/ekf/ekf.h"

#include <cmath>
#include <stdexcept>
#include <string>

#include "xk.h"

using namespace x;

namespace {
void check(xk_handle *h, int rc, const char *what) {
  if (rc != XK_OK) throw std::runtime_error(std::string(what) + ": " + xk_strerror(rc) + " (" + (h ? xk_last_error(h) : "") + ")");
}
// 15 x 15 products for composing IMU steps, column-major with the inner loop down a column (contiguous: the compiler
// vectorises it; the earlier version skipped the structural zeros of f_d entry by entry with a strided inner loop and
// was three times slower).  b's zeros are skipped per entry (f_d on the right is ~55 % zeros).
CoreCovMatrix mul15(const CoreCovMatrix &a, const CoreCovMatrix &b) {          // a b
  CoreCovMatrix c = CoreCovMatrix::Zero();
  for (int j = 0; j < 15; ++j)
    for (int k = 0; k < 15; ++k) {
      const double bkj = b.m[k + 15 * j];
      if (bkj == 0.0) continue;
      const double *ak = a.m + 15 * k;
      double *cj = c.m + 15 * j;
      for (int i = 0; i < 15; ++i) cj[i] += ak[i] * bkj;
    }
  return c;
}
CoreCovMatrix mul15_nt(const CoreCovMatrix &a, const CoreCovMatrix &b) {       // a b^T
  CoreCovMatrix c = CoreCovMatrix::Zero();
  for (int j = 0; j < 15; ++j)
    for (int k = 0; k < 15; ++k) {
      const double bjk = b.m[j + 15 * k];
      if (bjk == 0.0) continue;
      const double *ak = a.m + 15 * k;
      double *cj = c.m + 15 * j;
      for (int i = 0; i < 15; ++i) cj[i] += ak[i] * bjk;
    }
  return c;
}
}  // namespace

void Ekf::set(int sz, const State &default_state, Propagator *propagator, double time_margin, double a_m_max, int delta_seq_imu) {
  buffer_.assign(sz, default_state);
  for (State &s : buffer_) s.time_ = State::kInvalid;
  f_d_.assign(sz, CoreCovMatrix::Identity());
  q_d_.assign(sz, CoreCovMatrix::Zero());
  propagator_ = propagator;
  time_margin_ = time_margin;
  a_m_max_ = a_m_max;
  delta_seq_imu_ = delta_seq_imu;
  tail_ = -1;
  n_valid_ = 0;
  init_status_ = kNotInitialized;
}

void Ekf::setResident(bool on) {
  resident_ = on;
  updater_.setResident(on);
  if (on)
    for (State &s : buffer_) s.cov_ = Matrix();      // the ring keeps no host covariances
}

void Ekf::initializeFromState(const State &init_state) {                    // ekf.cpp:43-64
  if (buffer_.empty()) throw std::runtime_error("The EKF state buffer must have non-zero size.");
  if (init_state.p_array_.rows() != buffer_[0].p_array_.rows() || init_state.q_array_.rows() != buffer_[0].q_array_.rows() ||
      init_state.f_array_.rows() != buffer_[0].f_array_.rows() ||
      (!resident_ && init_state.cov_.rows() != buffer_[0].cov_.rows()))
    throw std::runtime_error("init_bfr_mismatch");
  for (State &s : buffer_) s.time_ = State::kInvalid;
  tail_ = 0;
  n_valid_ = 1;
  buffer_[0] = init_state;
  if (resident_) {
    const int n = init_state.nErrorStates();
    // an init state WITHOUT a covariance means the handle's covariance is already the initial one (e.g. restored with
    // xk_snapshot_P); otherwise it is uploaded, once
    if (init_state.cov_.size() > 0) {
      if (init_state.cov_.rows() != n) throw std::runtime_error("init_bfr_mismatch");
      check(updater_.engine(), xk_upload_P(updater_.engine(), init_state.cov_.data(), n, n), "xk_upload_P");
    }
    buffer_[0].cov_ = Matrix();
    cov_idx_ = 0;
  }
  init_status_ = kStandBy;
}

std::optional<State> Ekf::processImu(double timestamp, unsigned int seq, const Vector3 &w_m, const Vector3 &a_m) {
  if (init_status_ == kNotInitialized) return std::nullopt;
  std::lock_guard<std::mutex> g(mutex_);
  State &last_state = buffer_[tail_];
  if (init_status_ == kStandBy) {                                         // first IMU message (:82-99)
    if (a_m.norm() < a_m_max_) {
      last_state.setImu(timestamp, seq, w_m, a_m);
      last_seq_ = seq;
      init_status_ = kInitialized;
      return last_state;
    }
    return std::nullopt;
  }
  if (timestamp <= last_state.time_) return std::nullopt;                 // IMU going back in time (:102-108)
  last_seq_ = seq;
  const Vector3 a_m_smoothed = a_m.norm() < a_m_max_ ? a_m : last_state.a_m_;   // accelerometer spikes (:119-129)
  const int next = (tail_ + 1) % (int)buffer_.size();                     // state_buffer_.enqueueInPlace()
  // The ring wraps onto the slot whose covariance is resident on the device (buffer_sz - 1 IMU steps without a vision update:
  // before the first frame, during a tracking dropout).  The reference's enqueueInPlace() just overwrites the oldest state;
  // so does this, after moving the device covariance one slot on (it then belongs to the oldest state that survives).
  // ... unless an update is in flight on that very covariance (updater_.update() runs WITHOUT the mutex, ekf.cpp:186-205, on the
  // same engine handle and stream): propagating it now would rewrite the prior under the update's feet, from a second thread
  // inside a handle that is not thread-safe.  The reference overwrites the slot and DISCARDS that update when it comes back (the
  // slot's time no longer matches, ekf.cpp:229-239: a warning and nullopt).  Same here: the slot is overwritten, the update is
  // marked as discarded, and the step that would have carried the covariance past the slot is composed on the host and applied
  // to the prior -- saved on the device when the update started, because the ring was about to wrap -- once the update has
  // returned (endResidentUpdate).  Only an update that started with more than kWrapMargin free slots and still got lapped
  // (that many IMU samples during ONE update) has no saved prior: that, and only that, is an error.
  const int sz_ring = (int)buffer_.size();
  const int cov_slot = update_invalidated_ ? cov_target_ : cov_idx_;
  if (resident_ && next == cov_slot && update_in_flight_) {
    if (!snapshot_valid_)
      throw std::runtime_error("Ekf::processImu: the state ring wrapped onto the resident covariance while an update is in flight "
                               "and more than kWrapMargin IMU samples arrived during that one update: unrecoverable");
    if (!update_invalidated_) {
      update_invalidated_ = true;
      deferred_phi_ = CoreCovMatrix::Identity();
      deferred_q_ = CoreCovMatrix::Zero();
    }
    const int to = (cov_slot + 1) % sz_ring;            // the oldest state that survives: its recorded step, before a later wrap reuses the slot
    const CoreCovMatrix fq = mul15_nt(mul15(f_d_[to], deferred_q_), f_d_[to]);
    for (int k = 0; k < 225; ++k) deferred_q_.m[k] = fq.m[k] + q_d_[to].m[k];
    deferred_phi_ = mul15(f_d_[to], deferred_phi_);
    cov_target_ = to;
  } else if (resident_ && !update_in_flight_ && next == cov_idx_ && !advanceDeviceCovariance((cov_idx_ + 1) % sz_ring)) {
    // (only with no update in flight: once an update has been invalidated, cov_slot is cov_target_ and cov_idx_ says nothing any
    //  more -- advancing the device covariance from this thread would use the handle the update thread is inside of)
    throw std::runtime_error("Ekf: cannot advance the resident covariance past the slot the ring overwrites");
  }
  State &next_state = buffer_[next];
  next_state.setImu(timestamp, seq, w_m, a_m_smoothed);
  if (!propagator_) throw std::runtime_error("Ekf::processImu: no propagator");
  propagator_->propagateState(last_state, next_state);
  if (resident_) propagator_->transition(last_state, next_state, f_d_[next], q_d_[next]);   // applied on the device later
  else propagator_->propagateCovariance(last_state, next_state);
  tail_ = next;
  if (n_valid_ < (int)buffer_.size()) ++n_valid_;
  return next_state;
}

void Ekf::pushPropagatedState(const State &s) {
  std::lock_guard<std::mutex> g(mutex_);
  if (resident_) throw std::runtime_error("Ekf::pushPropagatedState: use processImu with a resident covariance");
  tail_ = (tail_ + 1) % (int)buffer_.size();
  buffer_[tail_] = s;
  if (n_valid_ < (int)buffer_.size()) ++n_valid_;
  init_status_ = kInitialized;
}

int Ekf::closestIdx(double timestamp) const {                               // state_buffer.cpp:26-63
  int best = -1;
  double bd = 1e300;
  for (int i = 0; i < (int)buffer_.size(); ++i) {
    if (buffer_[i].time_ == State::kInvalid) continue;
    const double d = std::fabs(buffer_[i].time_ - timestamp);
    if (d < bd) { bd = d; best = i; }
  }
  if (best < 0 || bd > time_margin_) return -1;
  return best;
}

// resident mode: bring the device covariance from slot cov_idx_ to slot idx (forward in time only)
bool Ekf::advanceDeviceCovariance(int idx) {
  if (idx == cov_idx_) return true;
  const int sz = (int)buffer_.size();
  int steps = (idx - cov_idx_ + sz) % sz;
  // idx must lie between cov_idx_ and the tail
  if (steps > (tail_ - cov_idx_ + sz) % sz) return false;                   // older than the resident covariance
  xk_handle *xk = updater_.engine();
  if (!compose_steps_) {
    for (int i = (cov_idx_ + 1) % sz, s = 0; s < steps; i = (i + 1) % sz, ++s)
      check(xk, xk_cov_propagate(xk, f_d_[i].m, 15, q_d_[i].m, 15), "xk_cov_propagate");
  } else {
    // P_ii <- F P_ii F^T + Q over several steps is one congruence with Phi = F_k ... F_1 and
    // Q_tot = sum_j (F_k ... F_{j+1}) Q_j (.)^T; P_iv <- Phi P_iv likewise.  15 x 15 products on the host, one launch.
    CoreCovMatrix phi = CoreCovMatrix::Identity(), qt = CoreCovMatrix::Zero();
    for (int i = (cov_idx_ + 1) % sz, s = 0; s < steps; i = (i + 1) % sz, ++s) {
      const CoreCovMatrix fq = mul15_nt(mul15(f_d_[i], qt), f_d_[i]);                    // F Q F^T
      for (int k = 0; k < 225; ++k) qt.m[k] = fq.m[k] + q_d_[i].m[k];
      phi = mul15(f_d_[i], phi);
    }
    check(xk, xk_cov_propagate(xk, phi.m, 15, qt.m, 15), "xk_cov_propagate");
  }
  cov_idx_ = idx;
  return true;
}

Matrix Ekf::covarianceAt(int idx) {
  std::lock_guard<std::mutex> g(mutex_);
  if (idx < 0) idx = tail_;
  if (!resident_) return buffer_[idx].cov_;
  const int n = buffer_[idx].nErrorStates(), sz = (int)buffer_.size();
  // only slots between the resident covariance's and the tail have a covariance any more (older ones were consumed)
  if ((idx - cov_idx_ + sz) % sz > (tail_ - cov_idx_ + sz) % sz)
    throw std::out_of_range("Ekf::covarianceAt: that state is older than the covariance resident on the device");
  Matrix P(n, n);
  check(updater_.engine(), xk_download_P(updater_.engine(), P.data(), n, n), "xk_download_P");
  const int steps = (idx - cov_idx_ + sz) % sz;
  std::vector<double> tmp(15);
  for (int i = (cov_idx_ + 1) % sz, s = 0; s < steps; i = (i + 1) % sz, ++s) {   // propagator.cpp:166-205 on the host
    const CoreCovMatrix &F = f_d_[i], &Q = q_d_[i];
    // rows: P[0:15, :] <- F P[0:15, :]
    for (int c = 0; c < n; ++c) {
      for (int r = 0; r < 15; ++r) { double a = 0; for (int k = 0; k < 15; ++k) a += F(r, k) * P(k, c); tmp[r] = a; }
      for (int r = 0; r < 15; ++r) P(r, c) = tmp[r];
    }
    // columns: P[:, 0:15] <- P[:, 0:15] F^T   (the core block thereby gets F P_ii F^T)
    for (int r = 0; r < n; ++r) {
      for (int c = 0; c < 15; ++c) { double a = 0; for (int k = 0; k < 15; ++k) a += P(r, k) * F(c, k); tmp[c] = a; }
      for (int c = 0; c < 15; ++c) P(r, c) = tmp[c];
    }
    for (int r = 0; r < 15; ++r)
      for (int c = 0; c < 15; ++c) P(r, c) += Q(r, c);
  }
  return P;
}

// resident mode, under mutex_: the device covariance goes to slot idx, the guard is armed, and -- if the IMU thread could lap the
// ring during this update (fewer than kWrapMargin free slots) -- the prior is saved on the device (one asynchronous copy)
bool Ekf::beginResidentUpdate(int idx) {
  if (!advanceDeviceCovariance(idx)) return false;                           // measurement older than the last update
  const int sz = (int)buffer_.size(), live = (tail_ - idx + sz) % sz;
  update_in_flight_ = true;
  update_invalidated_ = false;
  snapshot_valid_ = false;
  if (sz - 1 - live < kWrapMargin) {
    check(updater_.engine(), xk_snapshot_P(updater_.engine(), 2), "xk_snapshot_P");   // (slot 2 / 3: the filter loop's own)
    snapshot_valid_ = true;
  }
  return true;
}

// resident mode, under mutex_, after the plugin has returned (or thrown).  false: the IMU thread overwrote the slot meanwhile, the
// update is discarded as in ekf.cpp:229-239 -- the prior comes back and takes the IMU steps that passed over the slot.
bool Ekf::endResidentUpdate() {
  update_in_flight_ = false;
  if (!update_invalidated_) return true;
  update_invalidated_ = false;
  xk_handle *xk = updater_.engine();
  check(xk, xk_snapshot_P(xk, 3), "xk_snapshot_P");
  check(xk, xk_cov_propagate(xk, deferred_phi_.m, 15, deferred_q_.m, 15), "xk_cov_propagate");
  cov_idx_ = cov_target_;
  return false;
}

std::optional<State> Ekf::processUpdateMeasurement() {                       // ekf.cpp:179-213
  if (init_status_ != kInitialized) return std::nullopt;
  int idx;
  { std::lock_guard<std::mutex> g(mutex_); idx = closestIdx(updater_.getTime()); }
  if (idx < 0) return std::nullopt;
  if (resident_) {
    std::lock_guard<std::mutex> g(mutex_);
    if (!beginResidentUpdate(idx)) return std::nullopt;
  }
  State update_state = buffer_[idx];        // copy, not under the lock (as the reference); no covariance when resident
  try {
    updater_.update(update_state);          // <- plugin call; the mutex is NOT held
  } catch (...) {
    std::lock_guard<std::mutex> g(mutex_);
    if (resident_) (void)endResidentUpdate();
    throw;
  }
  bool ok;
  {
    std::lock_guard<std::mutex> g(mutex_);
    ok = (!resident_ || endResidentUpdate()) && repropagateFromStateAtIdx(update_state, idx);
  }
  if (ok) return update_state;
  return std::nullopt;                      // (the reference: "state buffer overwritten during update", the update is lost)
}

std::optional<State> Ekf::processOthersMeasurement(double timestamp) {       // ekf.cpp:143-176
  if (init_status_ != kInitialized) return std::nullopt;
  int idx;
  { std::lock_guard<std::mutex> g(mutex_); idx = closestIdx(timestamp); }
  if (idx < 0) return std::nullopt;
  if (resident_) {
    std::lock_guard<std::mutex> g(mutex_);
    if (!beginResidentUpdate(idx)) return std::nullopt;
  }
  State update_state = buffer_[idx];
  try {
    updater_.collaborativeUpdate(update_state);
  } catch (...) {
    std::lock_guard<std::mutex> g(mutex_);
    if (resident_) (void)endResidentUpdate();
    throw;
  }
  bool ok;
  {
    std::lock_guard<std::mutex> g(mutex_);
    ok = (!resident_ || endResidentUpdate()) && repropagateFromStateAtIdx(update_state, idx);
  }
  if (ok) return update_state;
  return std::nullopt;
}

bool Ekf::repropagateFromStateAtIdx(const State &state, int idx) {           // ekf.cpp:227-255
  if (buffer_[idx].getTime() != state.getTime()) return false;               // slot overwritten by the IMU thread
  buffer_[idx] = state;
  int curr = idx;
  while (curr != tail_) {
    const int next = (curr + 1) % (int)buffer_.size();
    if (propagator_) {
      propagator_->propagateState(buffer_[curr], buffer_[next]);
      if (resident_) propagator_->transition(buffer_[curr], buffer_[next], f_d_[next], q_d_[next]);
      else propagator_->propagateCovariance(buffer_[curr], buffer_[next]);
    }
    curr = next;
  }
  return true;
}
