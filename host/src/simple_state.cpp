// Mirror of src/x/ekf/simple_state.cpp plus the bridge to the flat inter-agent payload (include/xk.h, xk_pack_payload).
#include "x/ekf/simple_state.h"

#include <stdexcept>

using namespace x;

SimpleState::SimpleState(Vectorx dynamic_state, const Vectorx &positions_state, Vectorx orientations_state,
                         Vectorx features_state, Matrix cov, std::vector<int> anchor_idxs)
    : dynamic_state_(std::move(dynamic_state)), positions_state_(positions_state),
      orientations_state_(std::move(orientations_state)), features_state_(std::move(features_state)),
      anchor_idxs_(std::move(anchor_idxs)), cov_(std::move(cov)), n_poses_(positions_state.rows() / 3) {}

AttitudeList SimpleState::getCameraAttitudesList() const {                      // simple_state.cpp:33-49
  AttitudeList l(n_poses_, Attitude());
  for (int i = 0; i < n_poses_; i++)
    l[i] = Attitude{orientations_state_(4 * i), orientations_state_(4 * i + 1), orientations_state_(4 * i + 2),
                    orientations_state_(4 * i + 3)};
  return l;
}

TranslationList SimpleState::getCameraPositionsList() const {                   // :51-66 (translation_ is zero)
  TranslationList l(n_poses_, Translation());
  for (int i = 0; i < n_poses_; i++)
    l[i] = Translation{positions_state_(3 * i) + translation_(0), positions_state_(3 * i + 1) + translation_(1),
                       positions_state_(3 * i + 2) + translation_(2)};
  return l;
}

// hdr[8] {agent_id, timestamp, N, M, n, n_poses_valid, 0, 0} dyn[16] pos[3N] att[4N] feat[3M] anchors[M] cov[n*n]
long SimpleState::payloadDoubles(int N, int M) {
  const long n = kSizeCoreErr + 6L * N + 3L * M;
  return 24 + 3L * N + 4L * N + 3L * M + M + n * n;
}

SimpleState SimpleState::fromPayload(const double *b, int N, int M, double *agent_id, double *timestamp) {
  const int n = kSizeCoreErr + 6 * N + 3 * M;
  if ((int)b[2] != N || (int)b[3] != M || (int)b[4] != n) throw std::runtime_error("SimpleState::fromPayload: layout mismatch");
  if (agent_id) *agent_id = b[0];
  if (timestamp) *timestamp = b[1];
  Vectorx dyn(16, 1), pos(3 * N, 1), att(4 * N, 1), feat(3 * M, 1);
  Matrix cov(n, n);
  const double *p = b + 8;
  for (int i = 0; i < 16; ++i) dyn(i) = p[i];
  p += 16;
  for (int i = 0; i < 3 * N; ++i) pos(i) = p[i];
  p += 3 * N;
  for (int i = 0; i < 4 * N; ++i) att(i) = p[i];
  p += 4 * N;
  for (int i = 0; i < 3 * M; ++i) feat(i) = p[i];
  p += 3 * M;
  std::vector<int> anchors(M);
  for (int i = 0; i < M; ++i) anchors[i] = (int)p[i];
  p += M;
  for (size_t i = 0; i < (size_t)n * n; ++i) cov.data()[i] = p[i];
  return SimpleState(dyn, pos, att, feat, cov, anchors);
}

void SimpleState::toPayload(double agent_id, double timestamp, double *b) const {
  const int N = n_poses_, M = nFeaturesMax(), n = cov_.rows();
  b[0] = agent_id; b[1] = timestamp; b[2] = N; b[3] = M; b[4] = n; b[5] = N; b[6] = b[7] = 0;
  double *p = b + 8;
  for (int i = 0; i < 16; ++i) p[i] = dynamic_state_(i);
  p += 16;
  for (int i = 0; i < 3 * N; ++i) p[i] = positions_state_(i);
  p += 3 * N;
  for (int i = 0; i < 4 * N; ++i) p[i] = orientations_state_(i);
  p += 4 * N;
  for (int i = 0; i < 3 * M; ++i) p[i] = features_state_(i);
  p += 3 * M;
  for (int i = 0; i < M; ++i) p[i] = i < (int)anchor_idxs_.size() ? anchor_idxs_[i] : -1;
  p += M;
  for (size_t i = 0; i < (size_t)n * n; ++i) p[i] = cov_.data()[i];
}
