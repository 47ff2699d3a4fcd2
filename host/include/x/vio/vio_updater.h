// x/vio/vio_updater.h -- mirror of the concrete updater x::VioUpdater (include/x/vio/vio_updater.h:35,
// src/x/vio/vio_updater.cpp) restricted to the hot path: measurements arrive as ready-made track lists and match
// lists (the tracker / track manager / place-recognition front end is out of scope); preUpdate runs
// StateManager::manage, constructUpdate runs the MSCKF + MSCKF-SLAM + SLAM builders, the MSCKF-MSCKF CI block and the
// QR compression on the GPU, postUpdate initialises the new persistent features.
#pragma once
#include <memory>
#include <vector>

#include "x/ekf/simple_state.h"
#include "x/ekf/updater.h"
#include "x/vio/state_manager.h"

namespace x {
struct VioMeasurement {
  double timestamp = 0.0;
  TrackList msckf_tracks;          // full-length tracks ending at the current frame (track_manager.cpp:376-386)
  TrackList msckf_short_tracks;    // short tracks that just ended: used BEFORE the window slides (vio_updater.cpp:218-264)
  TrackList slam_tracks;           // one per persistent feature, newest observation last
  TrackList new_msckf_slam_tracks; // tracks whose landmark becomes a persistent feature this frame (vio_updater.cpp:176)
  TrackList new_slam_std_tracks;   // tracks initialised as standard SLAM features with a depth prior (:437-446)
  std::vector<unsigned int> lost_slam_track_idxs;   // persistent features to remove in manage() (vio_updater.cpp:170,202)
  MsckfMatches msckf_matches;      // this agent's MSCKF tracks also seen by other agents (place_recognition.cpp:137)
  SlamMatches slam_matches;        // persistent features matched to other agents' (processed in collaborativeUpdate)
};

class VioUpdater : public Updater {
 public:
  VioUpdater(int device, int n_poses_max, int n_feat_max, int k_max, double sigma_img, double sigma_landmark = 0.1,
             double ci_slam_w = 0.4, int iekf_iter = 1, double ci_msckf_w = 0.05, double rho_0 = 0.5,
             double sigma_rho_0 = 0.4);
  ~VioUpdater() override;
  VioUpdater(const VioUpdater &) = delete;

  void setMeasurement(const VioMeasurement &m) { measurement_ = m; stageMeasurementEarly(); }
  void setMeasurement(VioMeasurement &&m) { measurement_ = std::move(m); stageMeasurementEarly(); }   // the tracker hands its lists over
  // Window occupancy and SLAM anchors live in the StateManager (state_manager.h).  manage_window = true makes
  // preUpdate call StateManager::manage (vio_updater.cpp:200-202: slide, re-anchor, augment with the current camera
  // pose); with false the window arrives ready-made in the State (a snapshot taken after manage()).
  void setWindow(int n_poses, const std::vector<int> &anchor_idxs, bool filled_before = true);
  void setManageWindow(bool on) { manage_window_ = on; }
  StateManager &stateManager() { return state_manager_; }
  const std::vector<int> &getAnchorIdxs() const { return state_manager_.getAnchorIdxs(); }
  double getTime() const override { return measurement_.timestamp; }
  const std::vector<int> &getMsckfInlierFlags() const { return inlier_msckf_; }
  const std::vector<int> &getSlamInlierFlags() const { return inlier_slam_; }
  int ciEntriesOfLastUpdate() const { return n_ci_entries_; }

 protected:
  void preProcess(const State &) override {}
  bool preUpdate(State &state) override;                                                   // vio_updater.cpp:200-207
  bool preUpdateShortMsckf() override { return !measurement_.msckf_short_tracks.empty(); }  // :209-215
  bool preUpdateCI() override { return !measurement_.slam_matches.empty(); }
  void constructSlamCIUpdate(const State &state, MatrixList &S_list, MatrixList &P_list, MatrixList &H_list,
                             MatrixList &res_list) override;                               // vio_updater.cpp:81-115
  void constructUpdate(const State &state, Matrix &h, Matrix &res, Matrix &r) override;   // :267-423
  void constructShortMsckfUpdate(const State &state, Matrix &h, Matrix &res, Matrix &r) override;   // :218-264
  void constructUpdate(const State &state, Matrix &h, Matrix &res, Matrix &r, MatrixList &S_list, MatrixList &P_list,
                       MatrixList &H_list, MatrixList &res_list) override;                 // MULTI_UAV, :295-305
  void constructShortMsckfUpdate(const State &state, Matrix &h, Matrix &res, Matrix &r, MatrixList &S_list,
                                 MatrixList &P_list, MatrixList &H_list, MatrixList &res_list) override;
  void postUpdate(State &state, const Matrix &correction) override;                       // :425-446

 private:
  void windowLists(const State &state, std::vector<double> &q, std::vector<double> &p) const;  // state_manager.cpp:539-584
  // stage + per-feature build + QR compression of `tracks` (+ SLAM / MSCKF-SLAM rows when `with_slam`)
  void buildAndCompress(const State &state, const TrackList &tracks, bool with_slam, Matrix &h, Matrix &res, Matrix &r);
  // the MSCKF-MSCKF CI block (msckf_update.cpp:96-279) for every track of `tracks` that has matches; consumes them
  void buildMsckfCiLists(const State &state, const TrackList &tracks, MatrixList &S_list, MatrixList &P_list,
                         MatrixList &H_list, MatrixList &res_list);
  VioMeasurement measurement_;
  int n_poses_max_, n_feat_max_, k_max_;
  StateManager state_manager_;
  bool manage_window_ = false;
  double sigma_img_, sigma_landmark_, ci_slam_w_, ci_msckf_w_, rho_0_, sigma_rho_0_;
  std::vector<int> inlier_msckf_, inlier_slam_;
  int n_ci_entries_ = 0;
  bool flags_pending_ = false;
  bool tracks_staged_ = false;       // resident mode: measurement_.msckf_tracks already sit on the device
  void stageMeasurementEarly();
};
}  // namespace x
