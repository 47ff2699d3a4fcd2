// x/vio/state_manager.h -- mirror of x::StateManager (include/x/vio/state_manager.h, src/x/vio/state_manager.cpp)
// restricted to what runs every frame immediately before the visual update: manage() (state_manager.cpp:31-149)
// with its covariance operations on the GPU, and the window list accessors.
//
// The reference builds a dense n x n Jacobian for every operation and evaluates J * cov * J^T with two dense
// products.  Here the same J's are assembled row by row in CSR (they are identities / shifts apart from a few
// 3-row blocks) and applied to the covariance resident on the device (xk_cov_congruence, include/xk.h).
#pragma once
#include <vector>

#include "x/ekf/state.h"
#include "xk.h"

namespace x {
class StateManager {
 public:
  StateManager(int n_poses_max, int n_features_max, xk_handle *xk)
      : n_poses_max_(n_poses_max), n_features_max_(n_features_max), anchor_idxs_(n_features_max, -1), xk_(xk) {}

  void setEngine(xk_handle *xk) { xk_ = xk; }
  void clear();                                                          // state_manager.cpp:23-29
  // Removes the listed persistent features, slides the window if it is full (re-anchoring the features of
  // the oldest pose first) and augments state and covariance with the current camera pose.
  // resident = false: state.cov_ is uploaded first and downloaded afterwards (reference semantics: State owns
  // the covariance).  resident = true: the handle's covariance is the live one and state.cov_ is not touched.
  void manage(State &state, std::vector<unsigned int> del_feat_idx, bool resident = false);   // :31-149

  // After the update (VioUpdater::postUpdate, vio_updater.cpp:425-446): turn this frame's MSCKF-SLAM tracks into
  // persistent features (state_manager.cpp:151-174) / add standard SLAM features with a prior on rho (:176-197).
  // The covariance blocks are written on the device (xk_init_*_features); state.cov_ follows unless resident.
  void initMsckfSlamFeatures(State &state, int n_new, const Matrix &correction, double sigma_img, bool resident = false);
  void initStandardSlamFeatures(State &state, const Matrix &new_features, double sigma_img, double sigma_rho_0,
                                bool resident = false);

  int getNPoses() const { return n_poses_; }
  int getNFeatures() const { return n_features_; }
  const std::vector<int> &getAnchorIdxs() const { return anchor_idxs_; }
  // bookkeeping of a filter that is already running (the feature front end that fills it is out of scope)
  void restore(int n_poses, int n_features, const std::vector<int> &anchor_idxs, bool filled_before) {
    n_poses_ = n_poses; n_features_ = n_features; anchor_idxs_ = anchor_idxs; stateHasBeenFilledBefore_ = filled_before;
  }
  bool stateHasBeenFilledBefore() const { return stateHasBeenFilledBefore_; }

 private:
  // sparse row-wise Jacobian under construction
  struct Csr {
    std::vector<int> rp{0}, ci;
    std::vector<double> v;
    void entry(int c, double x) { ci.push_back(c); v.push_back(x); }
    void endRow() { rp.push_back((int)ci.size()); }
  };
  void apply(const Csr &J);                                              // queue P <- J P J^T
  void flush();                                                          // one congruence with the product of the queued J's
  void removeFeatureCov(unsigned int idx, int n);                        // :76-107
  void reparametrizeFeatures(const Matrix &atts_old, const Matrix &poss_old, Matrix &features);   // :351-482
  void slideWindow(Matrix &atts, Matrix &poss, int n);                   // :484-537
  void augmentCovariance(const State &state, int pos, int n);            // :273-349
  void addFeatureStates(State &state, const double *new_features, int n_new_states);   // :199-226 (state part)

  int n_poses_max_, n_features_max_;
  int n_poses_ = 0, n_features_ = 0;
  std::vector<int> anchor_idxs_;
  bool stateHasBeenFilledBefore_ = false;
  xk_handle *xk_;
  Csr pending_;            // product of the operations queued by the current manage() call (empty = identity)
  bool has_pending_ = false;
  Csr scratch_;            // apply(): the product under construction, swapped with pending_
  std::vector<double> acc_;
  std::vector<char> used_;
  std::vector<int> cols_;
};
}  // namespace x
