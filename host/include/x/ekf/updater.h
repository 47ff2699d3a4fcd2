// x/ekf/updater.h -- mirror of the abstract plugin x::Updater (include/x/ekf/updater.h:37-232).
// Same public protocol and the same protected hooks; the two Kalman-algebra members
// (applyUpdate / applyCI, src/x/ekf/updater.cpp:117-161) are the ones redirected to the C ABI.
// Single-agent signatures (the reference changes them under MULTI_UAV, updater.h:127-220); the
// collaborative entry point takes the CI lists explicitly.
#pragma once
#include <memory>
#include <vector>

#include "x/ekf/state.h"

struct xk_handle;

namespace x {
class Updater {
 public:
  virtual ~Updater() = default;
  virtual double getTime() const = 0;
  void update(State &state);                 // updater.cpp:39-115 (non-MULTI_UAV branch, IEKF loop)
  void collaborativeUpdate(State &state);    // updater.cpp:22-36

 protected:
  int iekf_iter_{1};
  xk_handle *xk_{nullptr};                   // device engine (include/xk.h), owned by the concrete updater
  bool compressed_on_device_{false};         // constructUpdate left [T_H|z] and the prior resident

  void applyUpdate(State &state, const Matrix &H, const Matrix &res, const Matrix &R, Matrix &correction_total,
                   bool cov_update = true);  // updater.cpp:117-141
  void applyCI(State &state, Matrix &ci_P, const Matrix &H, const Matrix &res, Matrix &S);  // updater.cpp:144-161

  virtual void preProcess(const State &state) = 0;
  virtual bool preUpdate(State &state) = 0;
  virtual bool preUpdateShortMsckf() = 0;
  virtual bool preUpdateCI() = 0;
  virtual void constructSlamCIUpdate(const State &state, std::vector<std::shared_ptr<Matrix>> &S_list,
                                     std::vector<std::shared_ptr<Matrix>> &P_list,
                                     std::vector<std::shared_ptr<Matrix>> &H_list,
                                     std::vector<std::shared_ptr<Matrix>> &res_list) = 0;
  virtual void constructUpdate(const State &state, Matrix &h, Matrix &res, Matrix &r) = 0;
  virtual void constructShortMsckfUpdate(const State &state, Matrix &h, Matrix &res, Matrix &r) = 0;
  virtual void postUpdate(State &state, const Matrix &correction) = 0;
};
}  // namespace x
