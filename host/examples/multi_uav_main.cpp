// Drives ONE MULTI_UAV visual update through the mirrored reference API (updater.cpp:84-97):
//   Ekf::processUpdateMeasurement -> Updater::update -> VioUpdater::constructUpdate (stacked rows + MSCKF-MSCKF CI lists)
//   -> applyCI per list entry -> applyUpdate -> State::correct
// with the covariance owned by the State (reference semantics) or resident on the device (argv[3] = 1).
// A negative K means |K| regular tracks FOLLOWED BY short tracks (vio_updater.cpp:218-264; the MULTI_UAV build only takes
// their CI entries, updater.cpp:52-66): the header then carries n_short right after ci_msckf_w, the short tracks' lengths and
// observations follow the regular ones, and a match whose track index is >= K refers to short track (index - K).
//   in : N K n_agents n_matches sigma_img ci_msckf_w [n_short] | per agent: q[4N] p[3N] P[n*n] | L_k[K (+ n_short)] | obs[2*sum L]
//        | per match: track agent L obs[2L]
//   out: P_post[n*n] | p_array[3N] q_array[4N] | p v q(xyzw) b_w b_a [16] | n_ci | inlier_msckf[K]
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
  const int N = (int)in[at++], Kin = (int)in[at++], n_agents = (int)in[at++], n_matches = (int)in[at++];
  const double sigma_img = in[at++], ci_msckf_w = in[at++];
  const int K = Kin < 0 ? -Kin : Kin, n_short = Kin < 0 ? (int)in[at++] : 0;
  const int n = kSizeCoreErr + 6 * N;
  std::vector<std::shared_ptr<SimpleState>> others(n_agents);
  State s(N, 0);
  s.setTime(1.0);
  for (int a = 0; a < n_agents; ++a) {
    Vectorx dyn(16, 1), pos(3 * N, 1), att(4 * N, 1), feat(0, 1);
    Matrix cov(n, n);
    dyn(9) = 1.0;                                          // unit quaternion (x, y, z, w at 6..9)
    for (int i = 0; i < 4 * N; ++i) att(i) = in[at + i];
    at += 4 * N;
    for (int i = 0; i < 3 * N; ++i) pos(i) = in[at + i];
    at += 3 * N;
    for (size_t i = 0; i < (size_t)n * n; ++i) cov.data()[i] = in[at + i];
    at += (size_t)n * n;
    if (a == 0) { s.q_array_ = att; s.p_array_ = pos; s.cov_ = cov; }
    else others[a] = std::make_shared<SimpleState>(dyn, pos, att, feat, cov, std::vector<int>());
  }
  VioMeasurement meas;
  meas.timestamp = 1.0;
  std::vector<int> L(K + n_short);
  for (int k = 0; k < K + n_short; ++k) L[k] = (int)in[at++];
  for (int k = 0; k < K + n_short; ++k) {
    Track t;
    t.setId(1000 + k);
    for (int i = 0; i < L[k]; ++i) { t.emplace_back(in[at], in[at + 1]); at += 2; }
    (k < K ? meas.msckf_tracks : meas.msckf_short_tracks).push_back(t);
  }
  for (int m = 0; m < n_matches; ++m) {
    const int trk = (int)in[at++], agent = (int)in[at++], Lm = (int)in[at++];
    auto rt = std::make_shared<Track>();
    rt->setId(5000 + m);
    for (int i = 0; i < Lm; ++i) { rt->emplace_back(in[at], in[at + 1]); at += 2; }
    meas.msckf_matches.emplace_back(agent, (uniqueId)(1000 + trk), rt->getId(), rt, others[agent]);
  }

  VioUpdater updater(0, N, 0, K, sigma_img, 0.1, 0.4, 1, ci_msckf_w);
  updater.setMultiUav(true);
  updater.setWindow(N, {});
  updater.setMeasurement(meas);
  Ekf ekf(updater);
  ekf.set(4, State(N, 0), nullptr, 0.02);
  ekf.setResident(resident);
  ekf.initializeFromState(s);
  ekf.processImu(1.0, 0, Vector3(0, 0, 0), Vector3(0, 0, 9.81));
  std::optional<State> post = ekf.processUpdateMeasurement();
  if (!post) { fprintf(stderr, "no update applied\n"); return 3; }
  const Matrix P = resident ? ekf.covarianceAt(-1) : post->cov_;

  FILE *f = fopen(argv[2], "wb");
  fwrite(P.data(), sizeof(double), (size_t)n * n, f);
  fwrite(post->p_array_.data(), sizeof(double), 3 * N, f);
  fwrite(post->q_array_.data(), sizeof(double), 4 * N, f);
  double dyn[16];
  post->getDynamicStates(dyn);
  fwrite(dyn, sizeof(double), 16, f);
  double nci = updater.ciEntriesOfLastUpdate();
  fwrite(&nci, sizeof(double), 1, f);
  for (int k = 0; k < K; ++k) { double v = updater.getMsckfInlierFlags()[k]; fwrite(&v, sizeof(double), 1, f); }
  fclose(f);
  printf("ok n=%d K=%d agents=%d matches=%d ci_entries=%d resident=%d\n", n, K, n_agents, n_matches, (int)nci, (int)resident);
  return 0;
}
