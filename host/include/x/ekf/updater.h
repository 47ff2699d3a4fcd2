// x/ekf/updater.h -- mirror of the abstract plugin x::Updater (include/x/ekf/updater.h:37-232).
// Same public protocol and the same protected hooks; the two Kalman-algebra members
// (applyUpdate / applyCI, src/x/ekf/updater.cpp:117-161) are the ones redirected to the C ABI.
//
// The reference switches the construct* signatures with the MULTI_UAV macro (updater.h:127-220).  The mirror keeps
// BOTH sets of hooks and picks the branch of update() at run time (setMultiUav): the single-agent branch iterates
// (IEKF, updater.cpp:99-110), the MULTI_UAV branch takes the CI lists out of constructUpdate, applies applyCI per entry
// and only then applyUpdate (updater.cpp:84-97).
//
// Resident covariance (setResident): the covariance of the state being updated lives on the device for the whole
// frame -- StateManager::manage, constructUpdate, applyCI, applyUpdate and postUpdate all work on the handle's
// covariance and State::cov_ is neither uploaded nor downloaded; only the n-vector corrections come back.
#pragma once
#include <memory>
#include <vector>

#include "x/ekf/state.h"

struct xk_handle;

namespace x {
using MatrixList = std::vector<std::shared_ptr<Matrix>>;

class Updater {
 public:
  virtual ~Updater() = default;
  virtual double getTime() const = 0;
  void update(State &state);                 // updater.cpp:39-115
  void collaborativeUpdate(State &state);    // updater.cpp:22-36
  void setMultiUav(bool on) { multi_uav_ = on; }
  bool multiUav() const { return multi_uav_; }
  void setResident(bool on) { resident_ = on; }
  bool resident() const { return resident_; }
  xk_handle *engine() const { return xk_; }
  // host-side wall time of the last update() by section, microseconds (0 preUpdate/manage, 1 constructUpdate, 2 applyUpdate,
  // 3 postUpdate); the device works asynchronously underneath, so the waiting lands in applyUpdate
  const double *profileUs() const { return prof_us_; }

 protected:
  int iekf_iter_{1};
  bool multi_uav_{false};
  bool resident_{false};
  xk_handle *xk_{nullptr};                   // device engine (include/xk.h), owned by the concrete updater
  double prof_us_[4] = {0, 0, 0, 0};
  // The pass Updater::update is about to run (updater.cpp:99-110): what applyUpdate will be called with is known before
  // constructUpdate, so a subclass can queue the whole pass at once (xk_build_compress_update_pass_async)
  const double *pass_correction_total_{nullptr};   // n doubles, the corrections of the passes so far (:140); nullptr outside the loop
  bool pass_cov_update_{true};
  bool compressed_on_device_{false};         // constructUpdate left [T_H|z] (and, unless CI ran, the prior) resident

  void applyUpdate(State &state, const Matrix &H, const Matrix &res, const Matrix &R, Matrix &correction_total,
                   bool cov_update = true);  // updater.cpp:117-141
  void applyCI(State &state, Matrix &ci_P, const Matrix &H, const Matrix &res, Matrix &S);  // updater.cpp:144-161

  virtual void preProcess(const State &state) = 0;
  virtual bool preUpdate(State &state) = 0;
  virtual bool preUpdateShortMsckf() = 0;
  virtual bool preUpdateCI() = 0;
  virtual void constructSlamCIUpdate(const State &state, MatrixList &S_list, MatrixList &P_list, MatrixList &H_list,
                                     MatrixList &res_list) = 0;
  // single-agent signatures (updater.h:196-220)
  virtual void constructUpdate(const State &state, Matrix &h, Matrix &res, Matrix &r) = 0;
  virtual void constructShortMsckfUpdate(const State &state, Matrix &h, Matrix &res, Matrix &r) = 0;
  // MULTI_UAV signatures (updater.h:127-192); the defaults forward to the single-agent ones with empty lists
  virtual void constructUpdate(const State &state, Matrix &h, Matrix &res, Matrix &r, MatrixList &S_list,
                               MatrixList &P_list, MatrixList &H_list, MatrixList &res_list) {
    (void)S_list; (void)P_list; (void)H_list; (void)res_list;
    constructUpdate(state, h, res, r);
  }
  virtual void constructShortMsckfUpdate(const State &state, Matrix &h, Matrix &res, Matrix &r, MatrixList &S_list,
                                         MatrixList &P_list, MatrixList &H_list, MatrixList &res_list) {
    (void)S_list; (void)P_list; (void)H_list; (void)res_list;
    constructShortMsckfUpdate(state, h, res, r);
  }
  virtual void postUpdate(State &state, const Matrix &correction) = 0;

 private:
  bool prior_stale_on_device_{false};        // an applyCI rewrote state.cov_ after constructUpdate staged the prior
};
}  // namespace x
