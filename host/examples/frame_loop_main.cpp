// A filter frame through the mirrored reference API, over and over:
//   Ekf::processImu x (imu_per_frame)   state propagation on the host (+ the covariance transition of each step)
//   VioUpdater::setMeasurement          this frame's tracks
//   Ekf::processUpdateMeasurement       covariance brought to the camera time -> Updater::update: preUpdate =
//                                       StateManager::manage (window slide + pose augmentation) -> constructUpdate ->
//                                       applyUpdate -> State::correct -> postUpdate -> re-propagation of the later states
// with the covariance owned by the State (reference semantics, mode 0) or resident on the device (mode 1: steps composed
// into one device propagation, mode 2: one device propagation per IMU step).
//
// Every frame replays the same scenario: the window enters shifted by one slot with the newest pose as the CURRENT camera
// pose (stationary IMU), so that manage() slides it into exactly the window the tracks were generated for; the covariance
// is brought back by a device-side copy (xk_snapshot_P), not an upload.
//   in : N K frames imu_per_frame mode sigma_img | q[4N] p[3N] | L_k[K] | obs[2*sum L] | P[n*n]
//        optionally behind it (round 6: a tracker whose outlier rate MOVES from frame to frame -- the acceptance ratio the handle picks
//        the single launch's geometry by): nsets | obs[nsets][2*sum L] | set_of_frame[frames]  (set 0 = the obs above)
//   out: P_post[n*n] (tail covariance after the last frame) | p_array[3N] q_array[4N] | core16 | ms_per_frame[frames]
//        with measurement sets: | per frame {set, inliers, give-ups so far, schedule of the compression, rel ||P - P_first(set)||_F}
//                               | P_first[nsets + 1][n*n] (the posterior of every set's first frame)
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "x/ekf/ekf.h"
#include "x/vio/vio_updater.h"
#include "xk.h"

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
  const int N = (int)in[at++], K = (int)in[at++], frames = (int)in[at++], imu_per_frame = (int)in[at++], mode = (int)in[at++];
  const double sigma_img = in[at++];
  const int n = kSizeCoreErr + 6 * N;
  const double *q = &in[at]; at += 4 * N;
  const double *p = &in[at]; at += 3 * N;
  VioMeasurement meas;
  std::vector<int> L(K);
  for (int k = 0; k < K; ++k) L[k] = (int)in[at++];
  for (int k = 0; k < K; ++k) {
    Track t;
    t.setId(k);
    for (int i = 0; i < L[k]; ++i) { t.emplace_back(in[at], in[at + 1]); at += 2; }
    meas.msckf_tracks.push_back(t);
  }
  Matrix P0(n, n);
  for (size_t i = 0; i < (size_t)n * n; ++i) P0.data()[i] = in[at + i];
  at += (size_t)n * n;
  // optional: more measurement sets (same tracks, other observations) and which one every frame sees
  std::vector<VioMeasurement> sets(1, meas);
  std::vector<int> set_of_frame(frames, 0);
  if (at < in.size()) {
    const int nsets = (int)in[at++];
    for (int s = 0; s < nsets; ++s) {
      VioMeasurement m2;
      for (int k = 0; k < K; ++k) {
        Track t;
        t.setId(k);
        for (int i = 0; i < L[k]; ++i) { t.emplace_back(in[at], in[at + 1]); at += 2; }
        m2.msckf_tracks.push_back(t);
      }
      sets.push_back(m2);
    }
    for (int f = 0; f < frames; ++f) set_of_frame[f] = (int)in[at++];
  }
  const bool moving = sets.size() > 1;
  std::vector<std::vector<double>> P_first(sets.size());
  std::vector<double> per_frame;

  // the state BEFORE manage(): window shifted by one slot, core pose = newest camera pose, at rest
  State s0(N, 0);
  for (int i = 0; i < N; ++i) {
    const int src = i > 0 ? i - 1 : 0;
    for (int c = 0; c < 4; ++c) s0.q_array_(4 * i + c) = q[4 * src + c];
    for (int c = 0; c < 3; ++c) s0.p_array_(3 * i + c) = p[3 * src + c];
  }
  s0.q_ = Quaternion(q[4 * (N - 1) + 3], q[4 * (N - 1)], q[4 * (N - 1) + 1], q[4 * (N - 1) + 2]);
  s0.p_ = Vector3(p[3 * (N - 1)], p[3 * (N - 1) + 1], p[3 * (N - 1) + 2]);
  const Vector3 g(0, 0, -9.81);
  double r[9];
  s0.q_.toRotationMatrix(r);
  const Vector3 a_rest(-(r[0] * g(0) + r[3] * g(1) + r[6] * g(2)), -(r[1] * g(0) + r[4] * g(1) + r[7] * g(2)),
                       -(r[2] * g(0) + r[5] * g(1) + r[8] * g(2)));   // C(q)^T (-g): specific force at rest
  const Vector3 w_rest(0, 0, 0);
  const double t0 = 10.0, dt_imu = 0.005;

  const bool resident = mode != 0;
  VioUpdater updater(0, N, 0, K, sigma_img);
  updater.setManageWindow(true);
  Propagator prop(g, ImuNoise());
  Propagator::acknowledgeModelProcessNoise();   // this benchmark knowingly runs the mirror's clean q_d model (a drop-in keeps the reference's)
  prop.setEngine(updater.engine());
  Ekf ekf(updater);
  ekf.set(4 * imu_per_frame + 8, State(N, 0), &prop, 0.5 * dt_imu);
  ekf.setResident(resident);
  ekf.setComposeSteps(mode != 2);
  xk_handle *xk = updater.engine();
  if (resident) {
    if (xk_upload_P(xk, P0.data(), n, n) != XK_OK || xk_snapshot_P(xk, 0) != XK_OK) { fprintf(stderr, "%s\n", xk_last_error(xk)); return 3; }
  }
  std::vector<double> ms(frames);
  double t_imu = 0, t_set = 0, t_upd = 0, prof[4] = {0, 0, 0, 0};
  std::optional<State> post;
  unsigned int seq = 0;
  for (int f = 0; f < frames; ++f) {
    // ---- replay set-up (not timed): same prior, same window, same IMU
    State init = s0;
    init.setTime(t0);
    if (resident) { if (xk_snapshot_P(xk, 1) != XK_OK) return 3; init.cov_ = Matrix(); }   // no covariance: the handle's is the prior
    else init.cov_ = P0;
    updater.setWindow(N, {}, true);
    ekf.initializeFromState(init);
    ekf.processImu(t0, seq++, w_rest, a_rest);                         // first message: stand-by -> initialised
    VioMeasurement frame_meas = sets[set_of_frame[f]];                 // what the tracker would hand over this frame
    frame_meas.timestamp = t0 + imu_per_frame * dt_imu;
    const auto c0 = std::chrono::steady_clock::now();
    // ---- one frame
    for (int i = 1; i <= imu_per_frame; ++i) ekf.processImu(t0 + i * dt_imu, seq++, w_rest, a_rest);
    const auto ca = std::chrono::steady_clock::now();
    updater.setMeasurement(std::move(frame_meas));
    const auto cb = std::chrono::steady_clock::now();
    post = ekf.processUpdateMeasurement();
    const auto c1 = std::chrono::steady_clock::now();
    t_imu += std::chrono::duration<double, std::milli>(ca - c0).count();
    t_set += std::chrono::duration<double, std::milli>(cb - ca).count();
    t_upd += std::chrono::duration<double, std::milli>(c1 - cb).count();
    for (int i = 0; i < 4; ++i) prof[i] += updater.profileUs()[i];
    if (!post) { fprintf(stderr, "frame %d: no update applied\n", f); return 3; }
    ms[f] = std::chrono::duration<double, std::milli>(c1 - c0).count();
    if (moving) {                                                      // (not timed)
      const Matrix Pf = resident ? ekf.covarianceAt(-1) : post->cov_;
      const int sidx = set_of_frame[f];
      double dev = 0.0;
      if (P_first[sidx].empty()) P_first[sidx].assign(Pf.data(), Pf.data() + (size_t)n * n);
      else {
        double num = 0, den = 0;
        for (size_t i = 0; i < (size_t)n * n; ++i) { const double d = Pf.data()[i] - P_first[sidx][i]; num += d * d; den += P_first[sidx][i] * P_first[sidx][i]; }
        dev = std::sqrt(num / den);
      }
      int inl_f = 0, sched = 0, armed = 0, giveups = 0, reason = 0;
      for (int v : updater.getMsckfInlierFlags()) inl_f += v;
      xk_caqr_status(xk, &sched, &armed, &giveups, &reason);
      per_frame.insert(per_frame.end(), {(double)sidx, (double)inl_f, (double)giveups, (double)sched, dev});
    }
  }
  const Matrix P = resident ? ekf.covarianceAt(-1) : post->cov_;
  FILE *fo = fopen(argv[2], "wb");
  fwrite(P.data(), sizeof(double), (size_t)n * n, fo);
  fwrite(post->p_array_.data(), sizeof(double), 3 * N, fo);
  fwrite(post->q_array_.data(), sizeof(double), 4 * N, fo);
  double dyn[16];
  post->getDynamicStates(dyn);
  fwrite(dyn, sizeof(double), 16, fo);
  fwrite(ms.data(), sizeof(double), frames, fo);
  if (moving) {
    fwrite(per_frame.data(), sizeof(double), per_frame.size(), fo);
    const std::vector<double> none((size_t)n * n, 0.0);
    for (const auto &pf : P_first) fwrite(pf.empty() ? none.data() : pf.data(), sizeof(double), (size_t)n * n, fo);
  }
  fclose(fo);
  int inl = 0;
  for (int v : updater.getMsckfInlierFlags()) inl += v;
  printf("ok n=%d K=%d frames=%d imu_per_frame=%d mode=%d inliers=%d | per frame: imu %.4f ms, setMeasurement %.4f ms, update %.4f ms (host sections, us: manage %.1f construct %.1f apply %.1f post %.1f)\n",
         n, K, frames, imu_per_frame, mode, inl, t_imu / frames, t_set / frames, t_upd / frames, prof[0] / frames, prof[1] / frames,
         prof[2] / frames, prof[3] / frames);
  return 0;
}
