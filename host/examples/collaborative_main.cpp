// Drives the collaborative (SLAM-SLAM covariance-intersection) route through the mirrored reference API:
//   Ekf::processOthersMeasurement (ekf.cpp:143-176) -> Updater::collaborativeUpdate (updater.cpp:22-36)
//   -> VioUpdater::constructSlamCIUpdate (vio_updater.cpp:81-123: one MultiSlamUpdate::processOneMatch per SlamMatch, every
//      entry built from the SAME state and prior) -> applyCI per inlier entry, each REPLACING the covariance (Q6)
//   -> State::correct per entry -> re-propagation
// with the covariance owned by the State (argv[3] = 0) or resident on the device (1).
//   in : N M n_others n_matches sigma_landmark ci_slam_w | own: q[4N] p[3N] f[3M] anchors[M] P[n*n]
//        | per other: q[4N] p[3N] f[3M] anchors[M] P[n*n] | per match: current_feature_id received_feature_id other
//   out: P_post[n*n] | p_array[3N] q_array[4N] f_array[3M] | core16 | n_slam_matches_left
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <vector>

#include "x/ekf/ekf.h"
#include "x/vio/vio_updater.h"

using namespace x;

static std::vector<double> slurp(const char *path) {
  FILE *f = fopen(path, "rb");
  if (!f) { perror(path); exit(2); }
  fseek(f, 0, SEEK_END);
  long sz = ftell(f);
  fseek(f, 0, SEEK_SET);
  std::vector<double> v(sz / sizeof(double));
  if (fread(v.data(), sizeof(double), v.size(), f) != v.size()) exit(2);
  fclose(f);
  return v;
}

int main(int argc, char **argv) {
  if (argc < 3) { fprintf(stderr, "usage: %s in.bin out.bin [resident]\n", argv[0]); return 2; }
  const bool resident = argc > 3 && atoi(argv[3]) != 0;
  const std::vector<double> in = slurp(argv[1]);
  size_t at = 0;
  const int N = (int)in[at++], M = (int)in[at++], n_others = (int)in[at++], n_matches = (int)in[at++];
  const double sigma_landmark = in[at++], ci_slam_w = in[at++];
  const int n = kSizeCoreErr + 6 * N + 3 * M;
  State s(N, M);
  s.setTime(1.0);
  std::vector<int> own_anchors(M);
  std::vector<std::shared_ptr<SimpleState>> others(n_others);
  for (int a = -1; a < n_others; ++a) {
    Vectorx dyn(16, 1), pos(3 * N, 1), att(4 * N, 1), feat(3 * M, 1);
    Matrix cov(n, n);
    std::vector<int> anchors(M);
    dyn(9) = 1.0;
    for (int i = 0; i < 4 * N; ++i) att(i) = in[at++];
    for (int i = 0; i < 3 * N; ++i) pos(i) = in[at++];
    for (int i = 0; i < 3 * M; ++i) feat(i) = in[at++];
    for (int i = 0; i < M; ++i) anchors[i] = (int)in[at++];
    for (size_t i = 0; i < (size_t)n * n; ++i) cov.data()[i] = in[at++];
    if (a < 0) { s.q_array_ = att; s.p_array_ = pos; s.f_array_ = feat; s.cov_ = cov; own_anchors = anchors; }
    else others[a] = std::make_shared<SimpleState>(dyn, pos, att, feat, cov, anchors);
  }
  VioMeasurement meas;
  meas.timestamp = 1.0;
  for (int m = 0; m < n_matches; ++m) {
    const int cur = (int)in[at++], recv = (int)in[at++], o = (int)in[at++];
    meas.slam_matches.emplace_back(o + 1, cur, recv, others[o]);
  }

  VioUpdater updater(0, N, M, 4, 1e-3, sigma_landmark, ci_slam_w);
  updater.setMultiUav(true);
  updater.setWindow(N, own_anchors);
  updater.setMeasurement(meas);
  Ekf ekf(updater);
  ekf.set(4, State(N, M), nullptr, 0.02);
  ekf.setResident(resident);
  ekf.initializeFromState(s);
  ekf.processImu(1.0, 0, Vector3(0, 0, 0), Vector3(0, 0, 9.81));
  std::optional<State> post = ekf.processOthersMeasurement(1.0);
  if (!post) { fprintf(stderr, "no collaborative update applied\n"); return 3; }
  const Matrix P = resident ? ekf.covarianceAt(-1) : post->cov_;

  FILE *f = fopen(argv[2], "wb");
  fwrite(P.data(), sizeof(double), (size_t)n * n, f);
  fwrite(post->p_array_.data(), sizeof(double), 3 * N, f);
  fwrite(post->q_array_.data(), sizeof(double), 4 * N, f);
  fwrite(post->f_array_.data(), sizeof(double), 3 * M, f);
  double dyn[16];
  post->getDynamicStates(dyn);
  fwrite(dyn, sizeof(double), 16, f);
  fclose(f);
  printf("ok n=%d matches=%d resident=%d\n", n, n_matches, (int)resident);
  return 0;
}
