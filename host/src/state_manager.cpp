// Mirror of src/x/vio/state_manager.cpp:23-537 (manage and the four covariance operations it calls).
#include "x/vio/state_manager.h"

#include <algorithm>
#include <stdexcept>
#include <string>

namespace x {
namespace {
void check(xk_handle *h, int rc, const char *what) {
  if (rc != XK_OK) throw std::runtime_error(std::string(what) + ": " + xk_strerror(rc) + " (" + xk_last_error(h) + ")");
}
void rotOf(const Matrix &atts, int pose, double r[9]) {   // Quaternion(x,y,z,w).normalized().toRotationMatrix()
  Quaternion q(atts(4 * pose + 3), atts(4 * pose), atts(4 * pose + 1), atts(4 * pose + 2));
  q.normalized().toRotationMatrix(r);
}
void skew(const double v[3], double s[9]) {               // x::Skew, tools.h:57-65
  s[0] = 0; s[1] = -v[2]; s[2] = v[1];
  s[3] = v[2]; s[4] = 0; s[5] = -v[0];
  s[6] = -v[1]; s[7] = v[0]; s[8] = 0;
}
void mul33(const double a[9], const double b[9], double c[9]) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) c[3 * i + j] = a[3 * i] * b[j] + a[3 * i + 1] * b[3 + j] + a[3 * i + 2] * b[6 + j];
}
void transpose33(const double a[9], double t[9]) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) t[3 * i + j] = a[3 * j + i];
}
}  // namespace

void StateManager::clear() {
  n_poses_ = 0;
  n_features_ = 0;
  anchor_idxs_.assign(n_features_max_, -1);
  stateHasBeenFilledBefore_ = false;
}

// The operations of one manage() call are composed on the host -- J_total = J_k ... J_2 J_1, each factor a
// permutation / identity with at most a few 15-entry rows -- and applied to the device covariance once.
void StateManager::apply(const Csr &J) {
  if (!has_pending_) { pending_ = J; has_pending_ = true; return; }
  const int n = (int)J.rp.size() - 1;
  Csr &out = scratch_;                                   // (members: no allocation per call once they have grown)
  out.rp.clear(); out.ci.clear(); out.v.clear(); out.rp.push_back(0);
  std::vector<double> &acc = acc_;
  std::vector<char> &used = used_;
  std::vector<int> &cols = cols_;
  acc.assign(n, 0.0);
  used.assign(n, 0);
  for (int r = 0; r < n; ++r) {
    cols.clear();
    for (int ia = J.rp[r]; ia < J.rp[r + 1]; ++ia) {
      const int a = J.ci[ia];
      for (int ib = pending_.rp[a]; ib < pending_.rp[a + 1]; ++ib) {
        const int b = pending_.ci[ib];
        if (!used[b]) { used[b] = 1; cols.push_back(b); }
        acc[b] += J.v[ia] * pending_.v[ib];
      }
    }
    std::sort(cols.begin(), cols.end());
    for (int b : cols) { out.entry(b, acc[b]); acc[b] = 0.0; used[b] = 0; }
    out.endRow();
  }
  std::swap(pending_, scratch_);
}

void StateManager::flush() {
  if (!has_pending_) return;
  check(xk_, xk_cov_congruence(xk_, pending_.rp.data(), pending_.ci.data(), pending_.v.data(), (int)pending_.ci.size()),
        "xk_cov_congruence");
  has_pending_ = false;
}

void StateManager::manage(State &state, std::vector<unsigned int> del_feat_idx, bool resident) {
  Matrix att = state.getOrientationArray(), pos = state.getPositionArray();       // :34-35
  const Attitude cae = state.computeCameraAttitude();                              // :38
  const Vector3 cpe = state.computeCameraPosition();                               // :39
  Matrix new_features = state.getFeatureArray();                                  // :45
  const int n = state.nErrorStates();
  if (!resident) check(xk_, xk_upload_P(xk_, state.getCovariance().data(), n, n), "xk_upload_P");   // :42

  // persistent feature removal, highest index first (:52-112)
  std::sort(del_feat_idx.begin(), del_feat_idx.end());
  for (size_t i = del_feat_idx.size(); i; --i) {
    const unsigned int idx = del_feat_idx[i - 1];
    const int n1 = n_features_ - (int)idx - 1;
    for (int k = 0; k < 3 * n1; ++k) new_features(3 * idx + k) = new_features(3 * (idx + 1) + k);   // :64-65
    for (int k = 0; k < 3; ++k) new_features(3 * (n_features_ - 1) + k) = 0.0;                        // :66
    anchor_idxs_.erase(anchor_idxs_.begin() + idx);                                                  // :71-72
    anchor_idxs_.push_back(-1);
    removeFeatureCov(idx, n);
    --n_features_;
  }
  if (n_poses_ == n_poses_max_) {                                                 // :119-125
    reparametrizeFeatures(att, pos, new_features);
    slideWindow(att, pos, n);
  }
  att(4 * n_poses_) = cae.ax; att(4 * n_poses_ + 1) = cae.ay; att(4 * n_poses_ + 2) = cae.az; att(4 * n_poses_ + 3) = cae.aw;   // :133
  for (int k = 0; k < 3; ++k) pos(3 * n_poses_ + k) = cpe(k);                     // :134
  augmentCovariance(state, n_poses_, n);                                          // :137
  ++n_poses_;
  // the window lists as they stand now are what constructUpdate stages next: handed over first, they travel with the
  // congruence operand instead of in a host-to-device copy of their own (xk_stage_window keeps identical lists for free)
  if (resident && n_poses_ >= 2) check(xk_, xk_stage_window(xk_, att.data(), pos.data(), n_poses_), "xk_stage_window");
  flush();
  if (!resident) check(xk_, xk_download_P(xk_, state.getCovarianceRef().data(), n, n), "xk_download_P");   // :145
  state.setOrientationArray(att);
  state.setPositionArray(pos);
  state.setFeatureArray(new_features);
}

// Rows/cols of feature idx are dropped, everything behind them moves up by three, the last three rows/cols
// become zero (:76-107) -- a congruence with a shift matrix.
void StateManager::removeFeatureCov(unsigned int idx, int n) {
  const int idx0 = kSizeClone + n_poses_max_ * 6 + (int)idx * 3;
  Csr J;
  for (int r = 0; r < n; ++r) {
    if (r < idx0) J.entry(r, 1.0);
    else if (r + 3 < n) J.entry(r + 3, 1.0);
    J.endRow();
  }
  apply(J);
}

void StateManager::reparametrizeFeatures(const Matrix &atts_old, const Matrix &poss_old, Matrix &features) {
  const int n = kSizeClone + n_poses_max_ * 6 + n_features_max_ * 3, idx1 = n_poses_max_ - 1;
  double R_old[9], R_new[9], R_newT[9], RnTRo[9];
  rotOf(atts_old, 0, R_old);
  rotOf(atts_old, idx1, R_new);
  transpose33(R_new, R_newT);
  mul33(R_newT, R_old, RnTRo);
  // (nothing anchored in the oldest pose -- the usual frame: J = I, and none of the bookkeeping below is needed)
  bool any = false;
  for (int j = 0; j < n_features_ && !any; ++j) any = anchor_idxs_[j] == 0;
  if (!any) return;
  // rows of J that differ from the identity: feature j -> five 3x3 blocks (:455-482)
  std::vector<std::vector<std::pair<int, double>>> special(n);
  std::vector<char> is_special(n, 0);
  for (int j = 0; j < n_features_; ++j) {
    if (anchor_idxs_[j] != 0) continue;                                                        // :367-371
    const double al = features(3 * j), be = features(3 * j + 1), rho = features(3 * j + 2);
    const double v[3] = {al, be, 1.0};
    double tmp[3], np_[3];
    for (int r = 0; r < 3; ++r)
      tmp[r] = -poss_old(3 * idx1 + r) + poss_old(r) + 1.0 / rho * (R_old[3 * r] * v[0] + R_old[3 * r + 1] * v[1] + R_old[3 * r + 2] * v[2]);
    for (int r = 0; r < 3; ++r) np_[r] = R_newT[3 * r] * tmp[0] + R_newT[3 * r + 1] * tmp[1] + R_newT[3 * r + 2] * tmp[2];   // Eq. 38, :402-406
    const double rho_new = 1.0 / np_[2], al_new = np_[0] * rho_new, be_new = np_[1] * rho_new;
    features(3 * j) = al_new; features(3 * j + 1) = be_new; features(3 * j + 2) = rho_new;   // :413
    anchor_idxs_[j] = idx1;                                                                   // :416
    double S[9], J_att_old[9], J_att_new[9], J_pos_old[9], J_pos_new[9], J_feat_old[9], mat[9], t9[9];
    skew(v, S);
    mul33(RnTRo, S, t9);
    for (int k = 0; k < 9; ++k) J_att_old[k] = -1.0 / rho * t9[k];                            // :424-427
    skew(np_, J_att_new);                                                                     // :430-436
    for (int k = 0; k < 9; ++k) { J_pos_old[k] = R_newT[k]; J_pos_new[k] = -R_newT[k]; }      // :439-444
    const double m_old[9] = {1, 0, -al / rho, 0, 1, -be / rho, 0, 0, -1.0 / rho};
    mul33(RnTRo, m_old, t9);
    for (int k = 0; k < 9; ++k) J_feat_old[k] = 1.0 / rho * t9[k];                            // :447-453
    // A_j (3 x n) is zero except five blocks; later writes win where blocks coincide (:458-473)
    std::vector<std::pair<int, const double *>> blocks = {
        {kSizeCoreErr + 3 * idx1, J_pos_new}, {kSizeCoreErr + 3 * idx1 + 3 * n_poses_max_, J_att_new},
        {kSizeCoreErr, J_pos_old}, {kSizeCoreErr + 3 * n_poses_max_, J_att_old},
        {kSizeCoreErr + 6 * n_poses_max_ + 3 * j, J_feat_old}};
    const double m_new[9] = {1, 0, -al_new, 0, 1, -be_new, 0, 0, -rho_new};
    for (int k = 0; k < 9; ++k) mat[k] = rho_new * m_new[k];
    const int n1 = kSizeCoreErr + 6 * n_poses_max_ + 3 * j;
    for (int r = 0; r < 3; ++r) { is_special[n1 + r] = 1; special[n1 + r].clear(); }
    std::vector<int> owner(n, -1);   // which block owns a column after all assignments
    for (size_t b = 0; b < blocks.size(); ++b)
      for (int c = 0; c < 3; ++c) owner[blocks[b].first + c] = (int)b;
    for (int col = 0; col < n; ++col) {
      if (owner[col] < 0) continue;
      const double *B = blocks[owner[col]].second;
      const int c = col - blocks[owner[col]].first;
      for (int r = 0; r < 3; ++r)   // J.block(n1,0,3,n) = rho_new * mat * A_j  (:481)
        special[n1 + r].push_back({col, mat[3 * r] * B[c] + mat[3 * r + 1] * B[3 + c] + mat[3 * r + 2] * B[6 + c]});
    }
  }
  if (std::find(is_special.begin(), is_special.end(), 1) == is_special.end()) return;   // no feature anchored in the oldest pose: J = I
  Csr J;
  for (int r = 0; r < n; ++r) {
    if (!is_special[r]) J.entry(r, 1.0);
    else for (auto &e : special[r]) J.entry(e.first, e.second);
    J.endRow();
  }
  apply(J);                                                                                    // :485
}

void StateManager::slideWindow(Matrix &atts, Matrix &poss, int n) {
  for (int k = 0; k < (n_poses_max_ - 1) * 4; ++k) atts(k) = atts(k + 4);                      // :488-493
  for (int k = 0; k < (n_poses_max_ - 1) * 3; ++k) poss(k) = poss(k + 3);
  for (int k = 0; k < 4; ++k) atts((n_poses_max_ - 1) * 4 + k) = 0.0;
  for (int k = 0; k < 3; ++k) poss((n_poses_max_ - 1) * 3 + k) = 0.0;
  // left_mult (:498-520): identity on core and features, pose blocks shifted up by one, last slot zero;
  // right_mult is its transpose (:510,522-530)
  const int w = (n_poses_max_ - 1) * 3, p0 = kSizeClone, a0 = kSizeClone + 3 * n_poses_max_, f0 = kSizeClone + 6 * n_poses_max_;
  Csr J;
  for (int r = 0; r < n; ++r) {
    if (r < p0 || r >= f0) J.entry(r, 1.0);
    else if (r < p0 + w) J.entry(r + 3, 1.0);
    else if (r >= a0 && r < a0 + w) J.entry(r + 3, 1.0);
    J.endRow();
  }
  apply(J);                                                                                    // :524
  for (int i = 0; i < n_features_; ++i) --anchor_idxs_[i];                                     // :527-529
  --n_poses_;                                                                                  // :532
}

void StateManager::augmentCovariance(const State &state, int pos, int n) {
  const int N3 = 3 * n_poses_max_, prow = kSizeClone + 3 * pos, arow = kSizeClone + N3 + 3 * pos;
  // columns of the new pose are deleted from P before the product (:326-339): fold that into J
  auto dead = [&](int c) { return (c >= prow && c < prow + 3) || (c >= arow && c < arow + 3); };
  auto unit_row = [&](int r) {                                                                 // :276-303
    if (stateHasBeenFilledBefore_) return true;
    if (r < kSizeClone + 3 * (pos + 1)) return true;
    if (r >= kSizeClone + N3 && r < kSizeClone + N3 + 3 * (pos + 1)) return true;
    return r >= kSizeClone + 2 * N3 && r < kSizeClone + 2 * N3 + 3 * n_features_;
  };
  double R[9], S[9], RS[9], Ric[9], RicT[9];
  state.q_.normalized().toRotationMatrix(R);
  const double pic[3] = {state.p_ic_(0), state.p_ic_(1), state.p_ic_(2)};
  skew(pic, S);
  mul33(R, S, RS);                                                                             // :312-315
  Quaternion qc(state.q_ic_.w(), -state.q_ic_.x(), -state.q_ic_.y(), -state.q_ic_.z());      // conjugate()
  qc.normalized().toRotationMatrix(Ric);                                                       // :318-322
  (void)RicT;
  Csr J;
  for (int r = 0; r < n; ++r) {
    if (r >= prow && r < prow + 3) {
      const int i = r - prow;
      J.entry(i, 1.0);                                        // d cam position / d imu position
      for (int c = 0; c < 3; ++c) J.entry(6 + c, -RS[3 * i + c]);   // ... / d imu attitude
      // (the unit entry J(r,r) multiplies a deleted column)
    } else if (r >= arow && r < arow + 3) {
      const int i = r - arow;
      for (int c = 0; c < 3; ++c) J.entry(6 + c, Ric[3 * i + c]);
    } else if (unit_row(r) && !dead(r)) {
      J.entry(r, 1.0);
    }
    J.endRow();
  }
  if (pos + 1 == n_poses_max_) stateHasBeenFilledBefore_ = true;                               // :342-343
  apply(J);                                                                                    // :346-347
}
// state_manager.cpp:199-226 without the covariance blocks (the device wrote those)
void StateManager::addFeatureStates(State &state, const double *new_features, int n_new_states) {
  Matrix features = state.getFeatureArray();
  for (int k = 0; k < n_new_states; ++k) features(3 * n_features_ + k) = new_features[k];   // :206-207
  state.setFeatureArray(features);
  const int n_new = n_new_states / 3;
  for (int i = 0; i < n_new; ++i) anchor_idxs_[n_features_ + i] = n_poses_ - 1;               // :219-221
  n_features_ += n_new;                                                                     // :223
}

void StateManager::initMsckfSlamFeatures(State &state, int n_new, const Matrix &correction, double sigma_img, bool resident) {
  if (n_new == 0) return;
  const int n = state.nErrorStates();
  if (!resident) check(xk_, xk_upload_P(xk_, state.getCovariance().data(), n, n), "xk_upload_P");   // :156
  std::vector<double> f(3 * (size_t)n_new);
  const int rc = xk_init_msckf_slam_features(xk_, n_features_, correction.data(), sigma_img, f.data());   // :158-171
  if (rc == XK_ESINGULAR) throw std::runtime_error(xk_last_error(xk_));   // H2 singular: camera hovering (:158-160)
  check(xk_, rc, "xk_init_msckf_slam_features");
  if (!resident) check(xk_, xk_download_P(xk_, state.getCovarianceRef().data(), n, n), "xk_download_P");
  addFeatureStates(state, f.data(), 3 * n_new);                                                       // :173
}

void StateManager::initStandardSlamFeatures(State &state, const Matrix &new_features, double sigma_img, double sigma_rho_0,
                                            bool resident) {
  const int n_new_states = (int)new_features.size();                                                  // :180
  if (n_new_states == 0) return;
  const int n = state.nErrorStates();
  if (!resident) check(xk_, xk_upload_P(xk_, state.getCovariance().data(), n, n), "xk_upload_P");
  check(xk_, xk_init_standard_slam_features(xk_, n_features_, n_new_states / 3, sigma_img, sigma_rho_0),
        "xk_init_standard_slam_features");                                                           // :183-193
  if (!resident) check(xk_, xk_download_P(xk_, state.getCovarianceRef().data(), n, n), "xk_download_P");
  addFeatureStates(state, new_features.data(), n_new_states);                                         // :196
}
}  // namespace x
