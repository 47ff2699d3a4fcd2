// Mirror of src/x/ekf/ekf.cpp + state_buffer.cpp for the update side of the filter loop.
#include "x/ekf/ekf.h"

#include <cmath>
#include <stdexcept>

using namespace x;

void Ekf::set(int sz, const State &default_state, Propagator *propagator, double time_margin) {
  buffer_.assign(sz, default_state);
  for (State &s : buffer_) s.time_ = State::kInvalid;
  propagator_ = propagator;
  time_margin_ = time_margin;
  tail_ = -1;
  n_valid_ = 0;
}

void Ekf::initializeFromState(const State &init_state) {                    // ekf.cpp:43-64
  if (buffer_.empty()) throw std::runtime_error("The EKF state buffer must have non-zero size.");
  if (init_state.p_array_.rows() != buffer_[0].p_array_.rows() || init_state.q_array_.rows() != buffer_[0].q_array_.rows() ||
      init_state.f_array_.rows() != buffer_[0].f_array_.rows() || init_state.cov_.rows() != buffer_[0].cov_.rows())
    throw std::runtime_error("init_bfr_mismatch");
  for (State &s : buffer_) s.time_ = State::kInvalid;
  tail_ = 0;
  n_valid_ = 1;
  buffer_[0] = init_state;
  initialized_ = true;
}

void Ekf::pushPropagatedState(const State &s) {
  std::lock_guard<std::mutex> g(mutex_);
  tail_ = (tail_ + 1) % (int)buffer_.size();
  buffer_[tail_] = s;
  if (n_valid_ < (int)buffer_.size()) ++n_valid_;
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

std::optional<State> Ekf::processUpdateMeasurement() {                       // ekf.cpp:179-213
  if (!initialized_) return std::nullopt;
  int idx;
  { std::lock_guard<std::mutex> g(mutex_); idx = closestIdx(updater_.getTime()); }
  if (idx < 0) return std::nullopt;
  State update_state = buffer_[idx];        // copy, not under the lock (as the reference)
  updater_.update(update_state);            // <- plugin call; the mutex is NOT held
  bool ok;
  { std::lock_guard<std::mutex> g(mutex_); ok = repropagateFromStateAtIdx(update_state, idx); }
  if (ok) return update_state;
  return std::nullopt;
}

std::optional<State> Ekf::processOthersMeasurement(double timestamp) {       // ekf.cpp:143-176
  if (!initialized_) return std::nullopt;
  int idx;
  { std::lock_guard<std::mutex> g(mutex_); idx = closestIdx(timestamp); }
  if (idx < 0) return std::nullopt;
  State update_state = buffer_[idx];
  updater_.collaborativeUpdate(update_state);
  bool ok;
  { std::lock_guard<std::mutex> g(mutex_); ok = repropagateFromStateAtIdx(update_state, idx); }
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
      propagator_->propagateCovariance(buffer_[curr], buffer_[next]);
    }
    curr = next;
  }
  return true;
}
