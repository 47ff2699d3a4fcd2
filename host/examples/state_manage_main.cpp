// Drives a sequence of StateManager::manage() calls through the mirrored API with the covariance operations on
// the GPU.  Flat little-endian double files, written/read by tests/test_gpu_state_manager.py.
//   in : N M S resident | cov[n*n] q_array[4N] p_array[3N] f_array[3M] | n_poses n_features filled anchors[M]
//        then S times: p[3] q[4 xyzw] q_ic[4 xyzw] p_ic[3] ndel del[ndel]
//   out: S times: cov[n*n] q_array[4N] p_array[3N] f_array[3M] n_poses n_features filled anchors[M]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "x/vio/state_manager.h"

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
  if (argc < 3) { fprintf(stderr, "usage: %s in.bin out.bin\n", argv[0]); return 2; }
  const std::vector<double> in = slurp(argv[1]);
  size_t at = 0;
  const int N = (int)in[at++], M = (int)in[at++], S = (int)in[at++];
  const bool resident = in[at++] != 0.0;
  const int n = kSizeCoreErr + 6 * N + 3 * M;
  xk_handle *xk = nullptr;
  if (xk_create(0, N, M, 4, &xk) != XK_OK) { fprintf(stderr, "xk_create failed\n"); return 1; }
  State s(N, M);
  for (int i = 0; i < n * n; ++i) s.cov_.data()[i] = in[at++];
  for (int i = 0; i < 4 * N; ++i) s.q_array_(i) = in[at++];
  for (int i = 0; i < 3 * N; ++i) s.p_array_(i) = in[at++];
  for (int i = 0; i < 3 * M; ++i) s.f_array_(i) = in[at++];
  StateManager sm(N, M, xk);
  {
    const int n_poses = (int)in[at++], n_features = (int)in[at++];
    const bool filled = in[at++] != 0.0;
    std::vector<int> anchors(M);
    for (int j = 0; j < M; ++j) anchors[j] = (int)in[at++];
    sm.restore(n_poses, n_features, anchors, filled);
  }
  if (resident && xk_upload_P(xk, s.cov_.data(), n, n) != XK_OK) return 1;
  FILE *fo = fopen(argv[2], "wb");
  if (!fo) { perror(argv[2]); return 2; }
  try {
    for (int step = 0; step < S; ++step) {
      for (int k = 0; k < 3; ++k) s.p_(k) = in[at++];
      const double qx = in[at], qy = in[at + 1], qz = in[at + 2], qw = in[at + 3]; at += 4;
      s.q_ = Quaternion(qw, qx, qy, qz);
      const double ix = in[at], iy = in[at + 1], iz = in[at + 2], iw = in[at + 3]; at += 4;
      s.q_ic_ = Quaternion(iw, ix, iy, iz);
      for (int k = 0; k < 3; ++k) s.p_ic_(k) = in[at++];
      const int ndel = (int)in[at++];
      std::vector<unsigned int> del;
      for (int k = 0; k < ndel; ++k) del.push_back((unsigned int)in[at++]);
      const auto t0 = std::chrono::steady_clock::now();
      sm.manage(s, del, resident);
      const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
      printf("step %d: manage %.1f us (%s)\n", step, us, resident ? "covariance resident on the device" : "upload + download");
      if (resident && xk_download_P(xk, s.cov_.data(), n, n) != XK_OK) return 1;   // only to report it
      fwrite(s.cov_.data(), sizeof(double), (size_t)n * n, fo);
      fwrite(s.q_array_.data(), sizeof(double), 4 * N, fo);
      fwrite(s.p_array_.data(), sizeof(double), 3 * N, fo);
      fwrite(s.f_array_.data(), sizeof(double), 3 * M, fo);
      std::vector<double> tail = {(double)sm.getNPoses(), (double)sm.getNFeatures(), sm.stateHasBeenFilledBefore() ? 1.0 : 0.0};
      for (int j = 0; j < M; ++j) tail.push_back((double)sm.getAnchorIdxs()[j]);
      fwrite(tail.data(), sizeof(double), tail.size(), fo);
    }
  } catch (const std::exception &e) {
    fprintf(stderr, "error: %s\n", e.what());
    return 1;
  }
  fclose(fo);
  xk_destroy(xk);
  return 0;
}
