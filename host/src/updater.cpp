// Mirror of src/x/ekf/updater.cpp.  Orchestration is the reference's; the Kalman algebra goes
// through the C ABI (include/xk.h) into the HIP kernels.
#include "x/ekf/updater.h"

#include <chrono>
#include <stdexcept>
#include <string>

#include "xk.h"

using namespace x;

static double usSince(std::chrono::steady_clock::time_point t0) {
  return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
}
static void check(xk_handle *h, int rc, const char *what) {
  if (rc != XK_OK) throw std::runtime_error(std::string(what) + ": " + xk_strerror(rc) + " (" + xk_last_error(h) + ")");
}

void Updater::collaborativeUpdate(State &state) {      // updater.cpp:22-36
  if (preUpdateCI()) {
    MatrixList S_list, P_list, H_list, res_list;
    constructSlamCIUpdate(state, S_list, P_list, H_list, res_list);
    for (size_t i = 0; i < P_list.size(); i++) applyCI(state, *P_list[i], *H_list[i], *res_list[i], *S_list[i]);
  }
}

void Updater::update(State &state) {                   // updater.cpp:39-115
  Matrix h, res, r;
  Matrix correction = Matrix::Zero(state.nErrorStates(), 1);
  preProcess(state);
  // before sliding the pose window with the new prior, the short MSCKF tracks are used (:50-73)
  const bool short_update_requested = preUpdateShortMsckf();
  if (short_update_requested) {
    if (multi_uav_) {
      MatrixList S_list, P_list, H_list, res_list;
      constructShortMsckfUpdate(state, h, res, r, S_list, P_list, H_list, res_list);
      for (size_t j = 0; j < P_list.size(); j++) applyCI(state, *P_list[j], *H_list[j], *res_list[j], *S_list[j]);
      // (the MULTI_UAV build stops here: it does not apply the short-track rows themselves, updater.cpp:56-66)
      compressed_on_device_ = false;
      prior_stale_on_device_ = false;
    } else {
      constructShortMsckfUpdate(state, h, res, r);
      if (h.size() > 0) applyUpdate(state, h, res, r, correction, true);
    }
  }
  auto t0 = std::chrono::steady_clock::now();
  const bool update_requested = preUpdate(state);
  prof_us_[0] = usSince(t0);
  prof_us_[1] = prof_us_[2] = prof_us_[3] = 0;
  if (update_requested) {
    correction = Matrix::Zero(state.nErrorStates(), 1);
    if (multi_uav_) {                                  // :84-97: CI entries first, then the regular update, no IEKF loop
      MatrixList S_list, P_list, H_list, res_list;
      constructUpdate(state, h, res, r, S_list, P_list, H_list, res_list);
      for (size_t j = 0; j < P_list.size(); j++) applyCI(state, *P_list[j], *H_list[j], *res_list[j], *S_list[j]);
      if (h.size() > 0) applyUpdate(state, h, res, r, correction, true);
    } else {
      for (int i = 0; i < iekf_iter_; i++) {
        const bool is_last_iter = i == iekf_iter_ - 1;
        t0 = std::chrono::steady_clock::now();
        pass_correction_total_ = correction.data();
        pass_cov_update_ = is_last_iter;
        constructUpdate(state, h, res, r);
        pass_correction_total_ = nullptr;
        prof_us_[1] += usSince(t0);
        t0 = std::chrono::steady_clock::now();
        if (h.size() > 0) applyUpdate(state, h, res, r, correction, is_last_iter);
        prof_us_[2] += usSince(t0);
      }
    }
    t0 = std::chrono::steady_clock::now();
    postUpdate(state, correction);
    prof_us_[3] = usSince(t0);
  }
}

void Updater::applyUpdate(State &state, const Matrix &H, const Matrix &res, const Matrix &R, Matrix &correction_total,
                          const bool cov_update) {     // updater.cpp:117-141
  Matrix &P = state.getCovarianceRef();
  const int n = state.nErrorStates();
  Matrix correction(n, 1);
  if (compressed_on_device_) {
    // constructUpdate left the compressed [T_H | z] resident in HBM.  The prior is the handle's covariance: what
    // constructUpdate staged, unless an applyCI has rewritten state.cov_ since (H and res stay linearised at the
    // staged prior, as in the reference, where constructUpdate runs before the applyCI loop).
    if (prior_stale_on_device_) {
      check(xk_, xk_upload_P(xk_, P.data(), n, n), "xk_upload_P");
      prior_stale_on_device_ = false;
    }
    check(xk_, xk_apply_update(xk_, correction_total.data(), cov_update ? 1 : 0, correction.data()), "xk_apply_update");
    if (cov_update && !resident_) check(xk_, xk_download_P(xk_, P.data(), n, n), "xk_download_P");
    compressed_on_device_ = false;
    for (int i = 0; i < n; ++i) correction_total(i) += correction(i);                     // :140
  } else {
    // A dense h (a subclass that builds its own rows instead of going through buildAndCompress).  With a resident covariance
    // there is no host copy to update: it is fetched, updated by the dense route and sent back -- two n x n transfers, the
    // price of leaving the device-resident construction, not an error.
    Matrix fetched;
    Matrix *Pd = &P;
    if (resident_) {
      fetched = Matrix(n, n);
      check(xk_, xk_download_P(xk_, fetched.data(), n, n), "xk_download_P");
      Pd = &fetched;
    }
    std::vector<double> rdiag(H.rows());
    for (int i = 0; i < H.rows(); ++i) rdiag[i] = R(i, i);
    check(xk_, xk_apply_update_dense(xk_, Pd->data(), n, n, H.data(), H.rows(), H.rows(), res.data(), rdiag.data(),
                                     correction_total.data(), cov_update ? 1 : 0, correction.data()),
          "xk_apply_update_dense");  // adds correction to correction_total (:140)
    if (resident_ && cov_update) check(xk_, xk_upload_P(xk_, Pd->data(), n, n), "xk_upload_P");
  }
  state.correct(correction);                                                                // :137
}

void Updater::applyCI(State &state, Matrix &ci_P, const Matrix &H, const Matrix &res, Matrix &S) {  // updater.cpp:144-161
  Matrix &P = state.getCovarianceRef();
  const int n = state.nErrorStates();
  Matrix correction(n, 1);
  if (resident_) {
    // P = sym((I - K H) ci_P) becomes the handle's covariance; it does not come back to the host
    check(xk_, xk_apply_ci_resident(xk_, ci_P.data(), n, n, H.data(), H.rows(), H.rows(), res.data(), S.data(), S.rows(),
                                    correction.data()), "xk_apply_ci_resident");
  } else {
    check(xk_, xk_apply_ci(xk_, P.data(), n, ci_P.data(), n, n, H.data(), H.rows(), H.rows(), res.data(), S.data(),
                           S.rows(), correction.data()), "xk_apply_ci");
    prior_stale_on_device_ = compressed_on_device_;   // the staged prior no longer is state.cov_
  }
  state.correct(correction);
}
