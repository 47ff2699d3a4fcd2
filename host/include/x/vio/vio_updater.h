// x/vio/vio_updater.h -- mirror of the concrete updater x::VioUpdater (include/x/vio/vio_updater.h:35,
// src/x/vio/vio_updater.cpp) restricted to the hot path: measurements arrive as ready-made track lists
// (the tracker / track manager / state manager front end is out of scope), constructUpdate runs the
// MSCKF + MSCKF-SLAM + SLAM builders and the QR compression on the GPU, postUpdate initialises the new
// persistent features.
#pragma once
#include <memory>
#include <vector>

#include "x/ekf/updater.h"

namespace x {
struct SlamMatchInput {            // one SLAM-SLAM match against a received SimpleState (vision/types.h:83-116)
  int current_feature_id, received_feature_id;
  AttitudeList other_C_q_G;
  TranslationList other_G_p_C;
  Matrix other_features;           // 3*M_other x 1
  std::vector<int> other_anchor_idxs;
  Matrix other_cov;
  int other_n_poses_max;
};

struct VioMeasurement {
  double timestamp = 0.0;
  TrackList msckf_tracks;          // full or short tracks ending at the current frame
  TrackList slam_tracks;           // one per persistent feature, newest observation last
  TrackList new_msckf_slam_tracks; // tracks whose landmark becomes a persistent feature this frame (vio_updater.cpp:176)
  std::vector<SlamMatchInput> slam_matches;
};

class VioUpdater : public Updater {
 public:
  VioUpdater(int device, int n_poses_max, int n_feat_max, int k_max, double sigma_img, double sigma_landmark = 0.1,
             double ci_slam_w = 0.4, int iekf_iter = 1);
  ~VioUpdater() override;
  VioUpdater(const VioUpdater &) = delete;

  void setMeasurement(const VioMeasurement &m) { measurement_ = m; }
  // window occupancy and SLAM anchors, kept by StateManager in the reference (state_manager.h)
  void setWindow(int n_poses, const std::vector<int> &anchor_idxs) { n_poses_ = n_poses; anchor_idxs_ = anchor_idxs; }
  const std::vector<int> &getAnchorIdxs() const { return anchor_idxs_; }
  double getTime() const override { return measurement_.timestamp; }
  const std::vector<int> &getMsckfInlierFlags() const { return inlier_msckf_; }
  const std::vector<int> &getSlamInlierFlags() const { return inlier_slam_; }
  xk_handle *engine() const { return xk_; }   // for the StateManager mirror, which works on the same resident covariance

 protected:
  void preProcess(const State &) override {}
  bool preUpdate(State &) override { return !(measurement_.msckf_tracks.empty() && measurement_.slam_tracks.empty()); }
  bool preUpdateShortMsckf() override { return false; }
  bool preUpdateCI() override { return !measurement_.slam_matches.empty(); }
  void constructSlamCIUpdate(const State &state, std::vector<std::shared_ptr<Matrix>> &S_list,
                             std::vector<std::shared_ptr<Matrix>> &P_list, std::vector<std::shared_ptr<Matrix>> &H_list,
                             std::vector<std::shared_ptr<Matrix>> &res_list) override;   // vio_updater.cpp:81-115
  void constructUpdate(const State &state, Matrix &h, Matrix &res, Matrix &r) override;   // vio_updater.cpp:267-423
  void constructShortMsckfUpdate(const State &state, Matrix &h, Matrix &res, Matrix &r) override {
    constructUpdate(state, h, res, r);                                                    // vio_updater.cpp:218-264
  }
  void postUpdate(State &state, const Matrix &correction) override;                       // vio_updater.cpp:425-446

 private:
  void windowLists(const State &state, std::vector<double> &q, std::vector<double> &p) const;  // state_manager.cpp:539-584
  VioMeasurement measurement_;
  int n_poses_max_, n_feat_max_, k_max_, n_poses_ = 0;
  std::vector<int> anchor_idxs_;
  double sigma_img_, sigma_landmark_, ci_slam_w_;
  std::vector<int> inlier_msckf_, inlier_slam_;
};
}  // namespace x
