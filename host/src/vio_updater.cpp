// Mirror of the hot-path half of src/x/vio/vio_updater.cpp.
#include "x/vio/vio_updater.h"

#include "x/vio/state_manager.h"

#include <stdexcept>
#include <string>

#include "xk.h"

using namespace x;

static void check(xk_handle *h, int rc, const char *what) {
  if (rc != XK_OK) throw std::runtime_error(std::string(what) + ": " + xk_strerror(rc) + " (" + (h ? xk_last_error(h) : "") + ")");
}

VioUpdater::VioUpdater(int device, int n_poses_max, int n_feat_max, int k_max, double sigma_img, double sigma_landmark,
                       double ci_slam_w, int iekf_iter)
    : n_poses_max_(n_poses_max), n_feat_max_(n_feat_max), k_max_(k_max), sigma_img_(sigma_img),
      sigma_landmark_(sigma_landmark), ci_slam_w_(ci_slam_w) {
  iekf_iter_ = iekf_iter;
  check(nullptr, xk_create(device, n_poses_max, n_feat_max, k_max, &xk_), "xk_create");  // no CPU fallback
  n_poses_ = n_poses_max;
}

VioUpdater::~VioUpdater() { xk_destroy(xk_); }

// StateManager::convertCameraAttitudesToList / PositionsToList (state_manager.cpp:539-584):
// the first n_poses_ slots of the window arrays.
void VioUpdater::windowLists(const State &state, std::vector<double> &q, std::vector<double> &p) const {
  const Matrix &qa = state.getOrientationArray(), &pa = state.getPositionArray();
  q.assign(qa.data(), qa.data() + 4 * n_poses_);
  p.assign(pa.data(), pa.data() + 3 * n_poses_);
}

void VioUpdater::constructUpdate(const State &state, Matrix &h, Matrix &res, Matrix &r) {
  const int n = state.nErrorStates();
  std::vector<double> q, p;
  windowLists(state, q, p);
  check(xk_, xk_stage_window(xk_, q.data(), p.data(), n_poses_), "xk_stage_window");
  // tracks -> CSR
  const TrackList &tr = measurement_.msckf_tracks;
  std::vector<int> off(tr.size() + 1, 0);
  std::vector<double> obs;
  for (size_t k = 0; k < tr.size(); ++k) {
    off[k + 1] = off[k] + (int)tr[k].size();
    for (const Feature &f : tr[k]) { obs.push_back(f.getX()); obs.push_back(f.getY()); }
  }
  check(xk_, xk_stage_tracks(xk_, off.data(), obs.data(), (int)tr.size()), "xk_stage_tracks");
  const TrackList &st = measurement_.slam_tracks;
  const int M = (int)st.size();
  std::vector<double> z(2 * (size_t)M);
  std::vector<int> tsz(M);
  for (int j = 0; j < M; ++j) {
    tsz[j] = (int)st[j].size();
    z[2 * j] = st[j].back().getX();   // SlamUpdate uses the newest observation only (slam_update.cpp:100-103)
    z[2 * j + 1] = st[j].back().getY();
  }
  check(xk_, xk_stage_slam(xk_, state.getFeatureArray().data(), anchor_idxs_.data(), tsz.data(), z.data(), M),
        "xk_stage_slam");
  {   // MSCKF-SLAM tracks (vio_updater.cpp:311-321): their rows sit between the MSCKF and the SLAM rows (:413-419)
    const TrackList &mt = measurement_.new_msckf_slam_tracks;
    std::vector<int> moff(mt.size() + 1, 0);
    std::vector<double> mobs;
    for (size_t k = 0; k < mt.size(); ++k) {
      moff[k + 1] = moff[k] + (int)mt[k].size();
      for (const Feature &f : mt[k]) { mobs.push_back(f.getX()); mobs.push_back(f.getY()); }
    }
    check(xk_, xk_stage_msckf_slam(xk_, moff.data(), mobs.data(), (int)mt.size()), "xk_stage_msckf_slam");
  }
  check(xk_, xk_upload_P(xk_, state.getCovariance().data(), n, n), "xk_upload_P");   // Matrix P = state.getCovariance()
  inlier_msckf_.assign(tr.size(), 0);
  inlier_slam_.assign(M, 0);
  check(xk_, xk_msckf_build(xk_, sigma_img_, inlier_msckf_.data(), nullptr, inlier_slam_.data(), nullptr),
        "xk_msckf_build");
  h = Matrix::Zero(n, n);
  res = Matrix::Zero(n, 1);
  check(xk_, xk_qr_compress(xk_, h.data(), n, res.data()), "xk_qr_compress");       // applyQRDecomposition
  r = Matrix::Zero(n, n);
  for (int i = 0; i < n; ++i) r(i, i) = sigma_img_ * sigma_img_;                       // vio_updater.cpp:508-509
  compressed_on_device_ = true;
}

void VioUpdater::constructSlamCIUpdate(const State &state, std::vector<std::shared_ptr<Matrix>> &S_list,
                                       std::vector<std::shared_ptr<Matrix>> &P_list,
                                       std::vector<std::shared_ptr<Matrix>> &H_list,
                                       std::vector<std::shared_ptr<Matrix>> &res_list) {
  const int n = state.nErrorStates();
  std::vector<double> q, p;
  windowLists(state, q, p);
  const Matrix &P = state.getCovariance();
  for (const SlamMatchInput &m : measurement_.slam_matches) {   // MultiSlamUpdate ctor loop, multi_slam_update.cpp:46-58
    std::vector<double> oq, op;
    for (const Attitude &a : m.other_C_q_G) { oq.push_back(a.ax); oq.push_back(a.ay); oq.push_back(a.az); oq.push_back(a.aw); }
    for (const Translation &t : m.other_G_p_C) { op.push_back(t.tx); op.push_back(t.ty); op.push_back(t.tz); }
    auto H = std::make_shared<Matrix>(3, n), res = std::make_shared<Matrix>(3, 1), S = std::make_shared<Matrix>(3, 3);
    auto Pj = std::make_shared<Matrix>(n, n);
    int inl = 0;
    double gamma = 0;
    const int rc = xk_multi_slam_match(xk_, q.data(), p.data(), n_poses_, state.getFeatureArray().data(),
                                       anchor_idxs_[m.current_feature_id], m.current_feature_id, P.data(), n, n,
                                       n_poses_max_, oq.data(), op.data(), (int)m.other_G_p_C.size(),
                                       m.other_features.data(), m.other_anchor_idxs[m.received_feature_id],
                                       m.received_feature_id, m.other_cov.data(), m.other_cov.rows(),
                                       m.other_cov.rows(), m.other_n_poses_max, sigma_landmark_, ci_slam_w_, &inl,
                                       &gamma, H->data(), 3, res->data(), S->data(), Pj->data(), n);
    if (rc == XK_EINVAL) throw std::runtime_error(xk_last_error(xk_));  // same places the reference throws
    check(xk_, rc, "xk_multi_slam_match");
    if (inl) { H_list.push_back(H); S_list.push_back(S); res_list.push_back(res); P_list.push_back(Pj); }
  }
  measurement_.slam_matches.clear();                                    // tracker_.cleanSlamMatches()
}

// MSCKF-SLAM feature initialisation after the update (vio_updater.cpp:425-435); the standard-SLAM branch
// (:437-446) needs the tracker's new_slam_std_trks_ and is reachable through StateManager directly.
void VioUpdater::postUpdate(State &state, const Matrix &correction) {
  const int n_new = (int)measurement_.new_msckf_slam_tracks.size();
  if (n_new == 0) return;
  const int n_existing = (int)measurement_.slam_tracks.size();
  StateManager sm(n_poses_max_, n_feat_max_, xk_);
  std::vector<int> anchors = anchor_idxs_;
  anchors.resize(n_feat_max_, -1);
  sm.restore(n_poses_, n_existing, anchors, true);
  sm.initMsckfSlamFeatures(state, n_new, correction, sigma_img_);
  anchor_idxs_ = sm.getAnchorIdxs();
}
