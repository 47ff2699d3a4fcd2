// Drives ONE visual update through the mirrored reference API:
//   Ekf::processUpdateMeasurement -> Updater::update -> VioUpdater::constructUpdate -> applyUpdate
// Input/outputs are flat little-endian double files written/read by tests/test_gpu_host_cpp.py.
//   in : N M K n_poses sigma_img | q[4N] p[3N] | L_k[K] | obs[2*sum L] | feat[3M] anchors[M] zlast[2M] tsz[M] | P[n*n]
//        optionally followed by  K2 M_used | L2_k[K2] | obs2[2*sum L2]:  K2 MSCKF-SLAM tracks (features initialised in
//        postUpdate) and the number of feature slots in use (M is then the capacity)
//   out: P_post[n*n] | p_array[3N] q_array[4N] f_array[3M] | inlier_msckf[K]
// Optional arguments: iekf_iter (Updater::update's IEKF loop, updater.cpp:99-110; default 1), resident (1: the covariance
// stays on the device, Ekf::setResident), core.bin (16 doubles p v q[xyzw] b_w b_a: the core state the update corrects
// through the cross-covariances).  With any of them the output is followed by the 16 dynamic states of the posterior.
#include <cstdio>
#include <cstdlib>
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
  if (argc < 3) { fprintf(stderr, "usage: %s in.bin out.bin [iekf_iter [resident [core.bin]]]\n", argv[0]); return 2; }
  const int iekf_iter = argc > 3 ? atoi(argv[3]) : 1;
  const bool resident = argc > 4 && atoi(argv[4]) != 0;
  const std::vector<double> in = slurp(argv[1]);
  size_t at = 0;
  const int N = (int)in[at++], M = (int)in[at++], K = (int)in[at++], n_poses = (int)in[at++];
  const double sigma_img = in[at++];
  const int n = kSizeCoreErr + 6 * N + 3 * M;
  State s(N, M);
  s.setTime(1.0);
  if (argc > 5) {
    const std::vector<double> core = slurp(argv[5]);
    if (core.size() != 16) { fprintf(stderr, "core.bin: 16 doubles expected\n"); return 2; }
    for (int i = 0; i < 3; ++i) { s.p_(i) = core[i]; s.v_(i) = core[3 + i]; s.b_w_(i) = core[10 + i]; s.b_a_(i) = core[13 + i]; }
    s.q_ = Quaternion(core[9], core[6], core[7], core[8]);
  }
  for (int i = 0; i < 4 * N; ++i) s.q_array_(i) = in[at + i];
  at += 4 * N;
  for (int i = 0; i < 3 * N; ++i) s.p_array_(i) = in[at + i];
  at += 3 * N;
  VioMeasurement meas;
  meas.timestamp = 1.0;
  std::vector<int> L(K);
  for (int k = 0; k < K; ++k) L[k] = (int)in[at++];
  for (int k = 0; k < K; ++k) {
    Track t;
    for (int i = 0; i < L[k]; ++i) { t.emplace_back(in[at], in[at + 1]); at += 2; }
    meas.msckf_tracks.push_back(t);
  }
  for (int i = 0; i < 3 * M; ++i) s.f_array_(i) = in[at + i];
  at += 3 * M;
  std::vector<int> anchors(M);
  for (int j = 0; j < M; ++j) anchors[j] = (int)in[at++];
  std::vector<double> zl(in.begin() + at, in.begin() + at + 2 * M);
  at += 2 * M;
  for (int j = 0; j < M; ++j) {
    Track t((size_t)in[at + j] - 1, Feature(0, 0));
    t.emplace_back(zl[2 * j], zl[2 * j + 1]);
    meas.slam_tracks.push_back(t);
  }
  at += M;
  for (size_t i = 0; i < (size_t)n * n; ++i) s.cov_.data()[i] = in[at + i];
  at += (size_t)n * n;
  if (at < in.size()) {
    const int K2 = (int)in[at++], M_used = (int)in[at++];
    std::vector<int> L2(K2);
    for (int k = 0; k < K2; ++k) L2[k] = (int)in[at++];
    for (int k = 0; k < K2; ++k) {
      Track t;
      for (int i = 0; i < L2[k]; ++i) { t.emplace_back(in[at], in[at + 1]); at += 2; }
      meas.new_msckf_slam_tracks.push_back(t);
    }
    meas.slam_tracks.resize(M_used);
    anchors.resize(M_used);
  }

  VioUpdater updater(0, N, M, K > 0 ? K : 1, sigma_img, 0.1, 0.4, iekf_iter);
  updater.setWindow(n_poses, anchors);
  updater.setMeasurement(meas);
  Ekf ekf(updater);
  ekf.set(4, State(N, M), nullptr, 0.02);
  ekf.setResident(resident);
  ekf.initializeFromState(s);
  ekf.processImu(1.0, 0, Vector3(0, 0, 0), Vector3(0, 0, 9.81));   // first IMU message: stand-by -> initialised (ekf.cpp:82-93)
  std::optional<State> post = ekf.processUpdateMeasurement();
  if (!post) { fprintf(stderr, "no update applied\n"); return 3; }

  const Matrix P = resident ? ekf.covarianceAt(-1) : post->cov_;
  FILE *f = fopen(argv[2], "wb");
  fwrite(P.data(), sizeof(double), (size_t)n * n, f);
  fwrite(post->p_array_.data(), sizeof(double), 3 * N, f);
  fwrite(post->q_array_.data(), sizeof(double), 4 * N, f);
  fwrite(post->f_array_.data(), sizeof(double), 3 * M, f);
  for (int k = 0; k < K; ++k) { double v = updater.getMsckfInlierFlags()[k]; fwrite(&v, sizeof(double), 1, f); }
  if (argc > 3) {
    double dyn[16];
    post->getDynamicStates(dyn);
    fwrite(dyn, sizeof(double), 16, f);
  }
  fclose(f);
  printf("ok n=%d K=%d M=%d iekf_iter=%d resident=%d\n", n, K, M, iekf_iter, (int)resident);
  return 0;
}
