// x/ekf/simple_state.h -- mirror of x::SimpleState (include/x/ekf/simple_state.h:30-75, src/x/ekf/simple_state.cpp): the
// snapshot of another agent's filter (state vectors, full covariance, SLAM anchors) that travels with a CI message.
// fromPayload / toPayload bridge it to the flat RCCL payload of xk_pack_payload (include/xk.h).
#pragma once
#include <vector>

#include "x/common/types.h"
#include "x/vision/types.h"

namespace x {
class SimpleState {
 public:
  SimpleState() = delete;
  SimpleState(Vectorx dynamic_state, const Vectorx &positions_state, Vectorx orientations_state, Vectorx features_state,
              Matrix cov, std::vector<int> anchor_idxs);                        // simple_state.cpp:22-31

  int nPosesMax() const { return n_poses_; }
  int nFeaturesMax() const { return features_state_.rows() / 3; }
  Vectorx getDynamicState() const { return dynamic_state_; }
  Vectorx getPositionState() const { return positions_state_; }
  Vectorx getOrientationState() const { return orientations_state_; }
  Vectorx getFeatureState() const { return features_state_; }
  Matrix getCovariance() const { return cov_; }
  const Matrix &covariance() const { return cov_; }                             // (no copy; not in the reference)
  std::vector<int> getAnchorIdxs() const { return anchor_idxs_; }
  int getErrorStateSize() const { return cov_.cols(); }
  Quaternion getRotation() const { return rotation_; }
  Vector3 getTranslation() const { return translation_; }
  int getAnchorIdat(int id) const { return anchor_idxs_[id]; }
  Vector3 getLastPose() const {
    const int r = positions_state_.rows();
    return Vector3(positions_state_(r - 3), positions_state_(r - 2), positions_state_(r - 1));
  }
  int nErrorStates() const {                                                     // simple_state.h:56-59
    return dynamic_state_.rows() - 1 + positions_state_.rows() + orientations_state_.rows() / 4 * 3 + features_state_.rows();
  }
  AttitudeList getCameraAttitudesList() const;                                  // simple_state.cpp:33-49
  TranslationList getCameraPositionsList() const;                               // :51-66

  // flat payload: [agent_id, timestamp, n_poses, M, dyn16, anchors(M), 4 pad | p(3N) | q(4N) | f(3M) | cov(n*n)]
  static long payloadDoubles(int n_poses_max, int n_features_max);
  static SimpleState fromPayload(const double *payload, int n_poses_max, int n_features_max, double *agent_id = nullptr,
                                 double *timestamp = nullptr);
  void toPayload(double agent_id, double timestamp, double *payload) const;

 private:
  const Vectorx dynamic_state_, positions_state_, orientations_state_, features_state_;
  const Quaternion rotation_ = Quaternion(1, 0, 0, 0);
  const Vector3 translation_ = Vector3(0, 0, 0);
  const std::vector<int> anchor_idxs_;
  const Matrix cov_;
  const int n_poses_ = -1;
};
}  // namespace x
