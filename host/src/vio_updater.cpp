// Mirror of the hot-path half of src/x/vio/vio_updater.cpp.
#include "x/vio/vio_updater.h"

#include <algorithm>
#include <cstring>
#include <stdexcept>
#include <string>

#include "x/vio/slam_update.h"
#include "xk.h"

using namespace x;

static void check(xk_handle *h, int rc, const char *what) {
  if (rc != XK_OK) throw std::runtime_error(std::string(what) + ": " + xk_strerror(rc) + " (" + (h ? xk_last_error(h) : "") + ")");
}

// one pass over the tracker's lists, straight into the engine's pinned staging memory
static void stageTracks(xk_handle *xk, const TrackList &tr) {
  int n_obs = 0;
  for (const Track &t : tr) n_obs += (int)t.size();
  int *off = nullptr;
  double *o = nullptr;
  check(xk, xk_stage_tracks_begin(xk, (int)tr.size(), n_obs, &off, &o), "xk_stage_tracks_begin");
  // a Track is a contiguous vector of Features and a Feature is its two normalised coordinates: one memcpy per track
  static_assert(sizeof(Feature) == 2 * sizeof(double), "Feature = (x, y)");
  off[0] = 0;
  for (size_t k = 0; k < tr.size(); ++k) {
    const size_t L = tr[k].size();
    off[k + 1] = off[k] + (int)L;
    if (L) memcpy(o, static_cast<const void *>(tr[k].data()), L * sizeof(Feature));
    o += 2 * L;
  }
  check(xk, xk_stage_tracks_end(xk), "xk_stage_tracks_end");
}

static void toCsr(const TrackList &tr, std::vector<int> &off, std::vector<double> &obs) {
  off.assign(tr.size() + 1, 0);
  for (size_t k = 0; k < tr.size(); ++k) off[k + 1] = off[k] + (int)tr[k].size();
  obs.resize(2 * (size_t)off[tr.size()]);
  double *o = obs.data();
  for (const Track &t : tr)
    for (const Feature &f : t) { *o++ = f.getX(); *o++ = f.getY(); }
}

VioUpdater::VioUpdater(int device, int n_poses_max, int n_feat_max, int k_max, double sigma_img, double sigma_landmark,
                       double ci_slam_w, int iekf_iter, double ci_msckf_w, double rho_0, double sigma_rho_0)
    : n_poses_max_(n_poses_max), n_feat_max_(n_feat_max), k_max_(k_max), state_manager_(n_poses_max, n_feat_max, nullptr),
      sigma_img_(sigma_img), sigma_landmark_(sigma_landmark), ci_slam_w_(ci_slam_w), ci_msckf_w_(ci_msckf_w),
      rho_0_(rho_0), sigma_rho_0_(sigma_rho_0) {
  iekf_iter_ = iekf_iter;
  check(nullptr, xk_create(device, n_poses_max, n_feat_max, k_max, &xk_), "xk_create");  // no CPU fallback
  state_manager_.setEngine(xk_);
  state_manager_.restore(n_poses_max, 0, std::vector<int>(n_feat_max, -1), true);
}

VioUpdater::~VioUpdater() { xk_destroy(xk_); }

void VioUpdater::setWindow(int n_poses, const std::vector<int> &anchor_idxs, bool filled_before) {
  std::vector<int> a = anchor_idxs;
  const int n_features = (int)a.size();
  a.resize(n_feat_max_, -1);
  state_manager_.restore(n_poses, n_features, a, filled_before);
}

// With a resident covariance the tracks of the frame go to the device as soon as they arrive (an asynchronous copy that
// overlaps the host work before the update: covariance propagation, manage()); constructUpdate finds them staged.
void VioUpdater::stageMeasurementEarly() {
  tracks_staged_ = false;
  if (!resident_ || !measurement_.msckf_short_tracks.empty()) return;   // (the short-track update stages its own tracks first)
  stageTracks(xk_, measurement_.msckf_tracks);
  tracks_staged_ = true;
}

// Manage vision states to be added, removed, re-parametrised or slid (vio_updater.cpp:200-207)
bool VioUpdater::preUpdate(State &state) {
  if (manage_window_) state_manager_.manage(state, measurement_.lost_slam_track_idxs, resident_);
  return !(measurement_.msckf_tracks.empty() && measurement_.slam_tracks.empty() && measurement_.new_slam_std_tracks.empty() &&
           measurement_.new_msckf_slam_tracks.empty());
}

// StateManager::convertCameraAttitudesToList / PositionsToList (state_manager.cpp:539-584):
// the first n_poses slots of the window arrays.
void VioUpdater::windowLists(const State &state, std::vector<double> &q, std::vector<double> &p) const {
  const Matrix &qa = state.getOrientationArray(), &pa = state.getPositionArray();
  const int np = state_manager_.getNPoses();
  q.assign(qa.data(), qa.data() + 4 * np);
  p.assign(pa.data(), pa.data() + 3 * np);
}

void VioUpdater::buildAndCompress(const State &state, const TrackList &tr, bool with_slam, Matrix &h, Matrix &res, Matrix &r) {
  const int n = state.nErrorStates();
  std::vector<double> q, p;
  windowLists(state, q, p);
  check(xk_, xk_stage_window(xk_, q.data(), p.data(), state_manager_.getNPoses()), "xk_stage_window");
  if (!(tracks_staged_ && &tr == &measurement_.msckf_tracks)) stageTracks(xk_, tr);
  tracks_staged_ = false;
  const TrackList &st = measurement_.slam_tracks;
  const int M = with_slam ? (int)st.size() : 0;
  std::vector<double> z(2 * (size_t)M);
  std::vector<int> tsz(M);
  for (int j = 0; j < M; ++j) {
    tsz[j] = (int)st[j].size();
    z[2 * j] = st[j].back().getX();   // SlamUpdate uses the newest observation only (slam_update.cpp:100-103)
    z[2 * j + 1] = st[j].back().getY();
  }
  check(xk_, xk_stage_slam(xk_, state.getFeatureArray().data(), state_manager_.getAnchorIdxs().data(), tsz.data(), z.data(), M),
        "xk_stage_slam");
  {   // MSCKF-SLAM tracks (vio_updater.cpp:311-321): their rows sit between the MSCKF and the SLAM rows (:413-419)
    static const TrackList none;
    const TrackList &mt = with_slam ? measurement_.new_msckf_slam_tracks : none;
    std::vector<int> moff;
    std::vector<double> mobs;
    toCsr(mt, moff, mobs);
    check(xk_, xk_stage_msckf_slam(xk_, moff.data(), mobs.data(), (int)mt.size()), "xk_stage_msckf_slam");
  }
  if (!resident_) check(xk_, xk_upload_P(xk_, state.getCovariance().data(), n, n), "xk_upload_P");   // Matrix P = state.getCovariance()
  inlier_msckf_.assign(tr.size(), 0);
  inlier_slam_.assign(M, 0);
  if (resident_) {
    // Nothing waits for the device here: the per-feature build and the compression are queued behind the staging copies
    // (and behind the covariance propagation / manage() of this frame); [T_H | z] stays in HBM, xk_apply_update consumes
    // it there and its synchronisation also brings the gate flags (fetched in postUpdate).  h only has to be non-empty
    // for Updater::update.
    // Single agent, one pass (updater.cpp:99-110 with iekf_iter = 1): nothing comes between constructUpdate and
    // applyUpdate(correction_total = 0, cov_update = true), so the Kalman update is queued with the rows -- inside the compression
    // launch where the geometry allows it.  IEKF passes (iekf_iter > 1): Updater::update says before constructUpdate what the
    // pass's applyUpdate will be called with (correction_total so far, cov_update = last pass), so they are queued whole too.
    // Only the MULTI_UAV order (applyCI entries rewrite the covariance in between, :84-97) keeps the two-call form.
    if (!multi_uav_ && (iekf_iter_ == 1 || !with_slam))   // (the short-track update, with_slam = false, is always one pass: updater.cpp:50-74)
      check(xk_, xk_build_compress_update_async(xk_, sigma_img_), "xk_build_compress_update_async");
    else if (!multi_uav_ && pass_correction_total_)
      check(xk_, xk_build_compress_update_pass_async(xk_, sigma_img_, pass_correction_total_, pass_cov_update_ ? 1 : 0),
            "xk_build_compress_update_pass_async");
    else
      check(xk_, xk_build_compress_async(xk_, sigma_img_), "xk_build_compress_async");
    flags_pending_ = true;
    h = Matrix::Zero(1, 1);
    res = Matrix::Zero(1, 1);
    r = Matrix::Zero(1, 1);
  } else {
    check(xk_, xk_msckf_build(xk_, sigma_img_, inlier_msckf_.data(), nullptr, inlier_slam_.data(), nullptr), "xk_msckf_build");
    h = Matrix::Zero(n, n);
    res = Matrix::Zero(n, 1);
    check(xk_, xk_qr_compress(xk_, h.data(), n, res.data()), "xk_qr_compress");       // applyQRDecomposition
    r = Matrix::Zero(n, n);
    for (int i = 0; i < n; ++i) r(i, i) = sigma_img_ * sigma_img_;                       // vio_updater.cpp:508-509
  }
  compressed_on_device_ = true;
}

// MsckfUpdate::preProcessOneTrack, MULTI_UAV part (msckf_update.cpp:96-279), for every track with matches: the
// matches of a track are consumed (erased from the list, :136-137) whether or not a CI entry results.
void VioUpdater::buildMsckfCiLists(const State &state, const TrackList &tracks, MatrixList &S_list, MatrixList &P_list,
                                   MatrixList &H_list, MatrixList &res_list) {
  MsckfMatches &all = measurement_.msckf_matches;
  if (all.empty()) return;
  const int n = state.nErrorStates(), np = state_manager_.getNPoses();
  std::vector<double> q, p;
  windowLists(state, q, p);
  // every CI entry is linearised at, and scaled from, the PRIOR of this update (cov_s, msckf_update.cpp:27-63)
  Matrix prior;
  const Matrix *P = &state.getCovariance();
  if (resident_) {
    prior = Matrix(n, n);
    check(xk_, xk_download_P(xk_, prior.data(), n, n), "xk_download_P");
    P = &prior;
  }
  for (const Track &track : tracks) {
    std::vector<MsckfMatch> mine;
    for (size_t i = 0; i < all.size();) {
      if (all[i].id_current_track == track.getId()) { mine.push_back(all[i]); all.erase(all.begin() + i); }
      else ++i;
    }
    if (mine.empty()) continue;
    const int k = (int)mine.size();
    std::vector<double> obs;
    for (const Feature &f : track) { obs.push_back(f.getX()); obs.push_back(f.getY()); }
    std::vector<std::vector<double>> mo(k), mq(k), mp(k);
    std::vector<Matrix> mP(k);
    std::vector<const double *> po(k), pq(k), pp(k), pP(k);
    std::vector<int> mL(k), mnp(k), mn(k);
    for (int i = 0; i < k; ++i) {
      const SimpleState &ss = *mine[i].state;
      for (const Feature &f : *mine[i].received_track_ptr) { mo[i].push_back(f.getX()); mo[i].push_back(f.getY()); }
      for (const Attitude &a : ss.getCameraAttitudesList()) { mq[i].push_back(a.ax); mq[i].push_back(a.ay); mq[i].push_back(a.az); mq[i].push_back(a.aw); }
      for (const Translation &t : ss.getCameraPositionsList()) { mp[i].push_back(t.tx); mp[i].push_back(t.ty); mp[i].push_back(t.tz); }
      mL[i] = (int)mine[i].received_track_ptr->size();
      mnp[i] = ss.nPosesMax();
      mn[i] = ss.getErrorStateSize();
      po[i] = mo[i].data(); pq[i] = mq[i].data(); pp[i] = mp[i].data(); pP[i] = ss.covariance().data();
    }
    auto H = std::make_shared<Matrix>(3 * k, n), res = std::make_shared<Matrix>(3 * k, 1), S = std::make_shared<Matrix>(3 * k, 3 * k);
    auto Pj = std::make_shared<Matrix>(n, n);
    int self_inl = 0, has_ci = 0;
    double self_gamma = 0, ci_gamma = 0;
    const int rc = xk_msckf_ci_track(xk_, obs.data(), (int)track.size(), q.data(), p.data(), np, P->data(), n, n, n_poses_max_,
                                     sigma_img_, k, po.data(), mL.data(), pq.data(), pp.data(), mnp.data(), pP.data(), mn.data(),
                                     ci_msckf_w_, &self_inl, &self_gamma, &has_ci, &ci_gamma, H->data(), 3 * k, res->data(),
                                     S->data(), 3 * k, Pj->data(), n);
    if (rc == XK_EINVAL) throw std::runtime_error(xk_last_error(xk_));   // the places the reference throws (ci.cpp:59-62)
    check(xk_, rc, "xk_msckf_ci_track");
    if (has_ci) { S_list.push_back(S); P_list.push_back(Pj); H_list.push_back(H); res_list.push_back(res); }   // :269-272
  }
  n_ci_entries_ = (int)P_list.size();
}

void VioUpdater::constructUpdate(const State &state, Matrix &h, Matrix &res, Matrix &r) {
  buildAndCompress(state, measurement_.msckf_tracks, true, h, res, r);
}

void VioUpdater::constructShortMsckfUpdate(const State &state, Matrix &h, Matrix &res, Matrix &r) {   // vio_updater.cpp:218-264
  buildAndCompress(state, measurement_.msckf_short_tracks, false, h, res, r);
}

void VioUpdater::constructUpdate(const State &state, Matrix &h, Matrix &res, Matrix &r, MatrixList &S_list, MatrixList &P_list,
                                 MatrixList &H_list, MatrixList &res_list) {
  n_ci_entries_ = 0;
  // (the CI entries and the stacked rows are both built from the prior; the entries first, because with a resident
  //  covariance they need the prior on the host once)
  buildMsckfCiLists(state, measurement_.msckf_tracks, S_list, P_list, H_list, res_list);
  buildAndCompress(state, measurement_.msckf_tracks, true, h, res, r);
}

void VioUpdater::constructShortMsckfUpdate(const State &state, Matrix &h, Matrix &res, Matrix &r, MatrixList &S_list,
                                           MatrixList &P_list, MatrixList &H_list, MatrixList &res_list) {
  n_ci_entries_ = 0;
  buildMsckfCiLists(state, measurement_.msckf_short_tracks, S_list, P_list, H_list, res_list);
  // (the MULTI_UAV build drops the short tracks' own rows, updater.cpp:56-66, so they are not built)
  h = Matrix(); res = Matrix(); r = Matrix();
}

void VioUpdater::constructSlamCIUpdate(const State &state, MatrixList &S_list, MatrixList &P_list, MatrixList &H_list,
                                       MatrixList &res_list) {
  const int n = state.nErrorStates();
  std::vector<double> q, p;
  windowLists(state, q, p);
  Matrix prior;
  const Matrix *P = &state.getCovariance();
  if (resident_) {
    prior = Matrix(n, n);
    check(xk_, xk_download_P(xk_, prior.data(), n, n), "xk_download_P");
    P = &prior;
  }
  const std::vector<int> &anchors = state_manager_.getAnchorIdxs();
  for (const SlamMatch &m : measurement_.slam_matches) {   // MultiSlamUpdate ctor loop, multi_slam_update.cpp:46-58
    const SimpleState &ss = *m.state;
    std::vector<double> oq, op;
    for (const Attitude &a : ss.getCameraAttitudesList()) { oq.push_back(a.ax); oq.push_back(a.ay); oq.push_back(a.az); oq.push_back(a.aw); }
    for (const Translation &t : ss.getCameraPositionsList()) { op.push_back(t.tx); op.push_back(t.ty); op.push_back(t.tz); }
    const Vectorx of = ss.getFeatureState();
    auto H = std::make_shared<Matrix>(3, n), res = std::make_shared<Matrix>(3, 1), S = std::make_shared<Matrix>(3, 3);
    auto Pj = std::make_shared<Matrix>(n, n);
    int inl = 0;
    double gamma = 0;
    const int rc = xk_multi_slam_match(xk_, q.data(), p.data(), state_manager_.getNPoses(), state.getFeatureArray().data(),
                                       anchors[m.current_feature_id], m.current_feature_id, P->data(), n, n, n_poses_max_,
                                       oq.data(), op.data(), ss.nPosesMax(), of.data(), ss.getAnchorIdat(m.received_feature_id),
                                       m.received_feature_id, ss.covariance().data(), ss.getErrorStateSize(),
                                       ss.getErrorStateSize(), ss.nPosesMax(), sigma_landmark_, ci_slam_w_, &inl, &gamma,
                                       H->data(), 3, res->data(), S->data(), Pj->data(), n);
    if (rc == XK_EINVAL) throw std::runtime_error(xk_last_error(xk_));  // same places the reference throws
    check(xk_, rc, "xk_multi_slam_match");
    if (inl) { H_list.push_back(H); S_list.push_back(S); res_list.push_back(res); P_list.push_back(Pj); }
  }
  measurement_.slam_matches.clear();                                    // tracker_.cleanSlamMatches()
}

// Feature initialisation after the update (vio_updater.cpp:425-446)
void VioUpdater::postUpdate(State &state, const Matrix &correction) {
  if (flags_pending_) {   // resident mode: the gate results arrived with applyUpdate's synchronisation
    check(xk_, xk_fetch_flags(xk_, inlier_msckf_.data(), nullptr, inlier_slam_.data(), nullptr), "xk_fetch_flags");
    flags_pending_ = false;
  }
  const int n_new = (int)measurement_.new_msckf_slam_tracks.size();
  if (n_new > 0) state_manager_.initMsckfSlamFeatures(state, n_new, correction, sigma_img_, resident_);   // :428-434
  if (!measurement_.new_slam_std_tracks.empty()) {                                                        // :437-446
    Matrix features_slam_std;
    SlamUpdate::computeInverseDepthsNew(measurement_.new_slam_std_tracks, rho_0_, features_slam_std);
    state_manager_.initStandardSlamFeatures(state, features_slam_std, sigma_img_, sigma_rho_0_, resident_);
  }
}
