// x/ekf/ekf.h -- mirror of x::Ekf (include/x/ekf/ekf.h, src/x/ekf/ekf.cpp): the filter loop that owns
// the time-sorted state ring and calls the Updater plugin.  IMU propagation is the step BEFORE the
// path (out of scope now) and is injected through x::Propagator.
#pragma once
#include <mutex>
#include <optional>
#include <vector>

#include "x/ekf/updater.h"

namespace x {
class Propagator {                       // include/x/ekf/propagator.h
 public:
  virtual ~Propagator() = default;
  virtual void propagateState(const State &curr, State &next) = 0;
  virtual void propagateCovariance(const State &curr, State &next) = 0;
};

class Ekf {
 public:
  explicit Ekf(Updater &updater) : updater_(updater) {}                      // ekf.cpp:25
  void set(int state_buffer_sz, const State &default_state, Propagator *propagator, double time_margin = 0.005);
  void initializeFromState(const State &init_state);                        // ekf.cpp:43-64
  // buffers a propagated state produced by the caller's IMU integration (stand-in for Ekf::processImu)
  void pushPropagatedState(const State &s);
  std::optional<State> processUpdateMeasurement();                          // ekf.cpp:179-213
  std::optional<State> processOthersMeasurement(double timestamp);          // ekf.cpp:143-176
  const State &tail() const { return buffer_[tail_]; }

 private:
  int closestIdx(double timestamp) const;                                   // state_buffer.cpp:26-63
  bool repropagateFromStateAtIdx(const State &state, int idx);              // ekf.cpp:227-255
  Updater &updater_;
  Propagator *propagator_ = nullptr;
  std::vector<State> buffer_;
  int tail_ = -1, n_valid_ = 0;
  double time_margin_ = 0.005;
  std::mutex mutex_;
  bool initialized_ = false;
};
}  // namespace x
