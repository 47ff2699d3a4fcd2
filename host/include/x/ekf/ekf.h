// x/ekf/ekf.h -- mirror of x::Ekf (include/x/ekf/ekf.h, src/x/ekf/ekf.cpp): the filter loop that owns the time-sorted
// state ring, propagates it with the IMU and calls the Updater plugin.
//
// Two ways of keeping the covariance:
//  * reference semantics (default): every State of the ring owns an n x n covariance on the host; each covariance
//    operation uploads it, runs on the device and downloads it again.
//  * resident (setResident(true)): ONE covariance lives in HBM -- the covariance at ring slot `cov_idx_`.  IMU steps
//    only record their 15 x 15 transition (f_d, q_d) in the slot; processUpdateMeasurement advances the device
//    covariance to the slot being updated (Propagator::propagateCovarianceMatrices per recorded step, or the steps
//    composed into one, see setComposeSteps), lets the Updater work on it in place (manage -> constructUpdate ->
//    applyCI / applyUpdate -> postUpdate) and re-propagates the later STATES on the host (ekf.cpp:227-255), refreshing
//    their recorded transitions.  No n x n matrix crosses PCIe in a frame; covarianceAt() fetches one on demand.
#pragma once
#include <mutex>
#include <optional>
#include <vector>

#include "x/ekf/propagator.h"
#include "x/ekf/updater.h"

namespace x {
class Ekf {
 public:
  explicit Ekf(Updater &updater) : updater_(updater) {}                      // ekf.cpp:25
  void set(int state_buffer_sz, const State &default_state, Propagator *propagator, double time_margin = 0.005,
           double a_m_max = 50.0, int delta_seq_imu = 1);                     // ekf.cpp:32-41
  void setResident(bool on);
  void setComposeSteps(bool on) { compose_steps_ = on; }
  void initializeFromState(const State &init_state);                        // ekf.cpp:43-64
  std::optional<State> processImu(double timestamp, unsigned int seq, const Vector3 &w_m, const Vector3 &a_m);   // ekf.cpp:66-141
  // buffers a propagated state produced by the caller's own IMU integration (reference-semantics mode only)
  void pushPropagatedState(const State &s);
  std::optional<State> processUpdateMeasurement();                          // ekf.cpp:179-213
  std::optional<State> processOthersMeasurement(double timestamp);          // ekf.cpp:143-176
  const State &tail() const { return buffer_[tail_]; }
  // resident mode: the covariance of ring slot `idx` (-1 = the tail), computed from the device covariance and the
  // recorded steps WITHOUT moving the device covariance (a download plus 15-row strip products on the host)
  Matrix covarianceAt(int idx = -1);

 private:
  enum InitStatus { kNotInitialized, kStandBy, kInitialized };
  int closestIdx(double timestamp) const;                                   // state_buffer.cpp:26-63
  bool repropagateFromStateAtIdx(const State &state, int idx);              // ekf.cpp:227-255
  bool advanceDeviceCovariance(int idx);                                    // resident: cov_idx_ -> idx
  Updater &updater_;
  Propagator *propagator_ = nullptr;
  std::vector<State> buffer_;
  std::vector<CoreCovMatrix> f_d_, q_d_;   // resident: transition INTO slot i from its predecessor
  int tail_ = -1, n_valid_ = 0;
  int cov_idx_ = -1;                       // resident: the slot whose covariance the device holds
  double time_margin_ = 0.005, a_m_max_ = 50.0;
  int delta_seq_imu_ = 1;
  unsigned int last_seq_ = 0;
  std::mutex mutex_;
  InitStatus init_status_ = kNotInitialized;
  bool resident_ = false, compose_steps_ = true;
  bool update_in_flight_ = false;          // resident: updater_.update() is running on the device covariance (guarded by mutex_)
  // resident: the ring wrapped onto the slot of the update in flight (ekf.cpp:229-239: the reference overwrites the slot and discards
  // that update when it comes back).  The prior was saved on the device before the update started (only when the ring was within
  // kWrapMargin slots of wrapping); the IMU steps that pass over the slot meanwhile are composed here and applied to the restored
  // prior once the update has returned.
  static constexpr int kWrapMargin = 32;
  bool snapshot_valid_ = false, update_invalidated_ = false;
  int cov_target_ = -1;
  CoreCovMatrix deferred_phi_, deferred_q_;
  bool beginResidentUpdate(int idx);       // under mutex_: advance the covariance, arm the guard, save the prior if a wrap is possible
  bool endResidentUpdate();                // under mutex_: false = the update was discarded (prior restored, deferred steps applied)
};
}  // namespace x
