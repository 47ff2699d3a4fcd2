// eigen_variant.cpp -- TEST INFRASTRUCTURE (CPU baseline footnote of bench.py; never linked into the product).
//
// "if the GPU box happens to have Eigen3 (__has_include(<Eigen/Dense>)), additionally build a true-Eigen variant of
// a9 / a11 / a12 and report it".  The reference's hot loops are Eigen expressions; oracle/xk_oracle.c restates them with its own
// loops.  This file runs the SAME three steps through Eigen itself, written the way the reference writes them:
//   a9   MsckfUpdate::processOneTrack, msckf_update.cpp:423-463   Q = Hf.householderQr().householderQ(); A = Q.rightCols(2L-3);
//                                                                res0 = A^T res; jac0 = A^T jac; S = jac0 P jac0^T + var I;
//                                                                gamma = res0^T S.inverse() res0; inlier iff gamma < chi2
//   a11  VioUpdater::applyQRDecomposition, vio_updater.cpp:487-512  HouseholderQR of [h | res], R's top n + 1 rows
//   a12  Updater::applyUpdate, updater.cpp:117-141                S = H P H^T + R; K = P H^T S.inverse(); P = (I - K H) P; sym
// The per-track inputs (jac 2L x n, Hf 2L x 3, res 2L: what msckf_update.cpp:328-417 leaves) come from the C restatement
// (xo_track_jacobians), so that what is timed here is Eigen's linear algebra and nothing else.
//
// This image has no Eigen: the program then prints {"eigen": "absent"} -- which is what bench.py reports.  The Eigen branch has
// never been compiled by the builder (nothing to compile it against); bench.py treats a compile error as "present, did not build".
//
//   usage: eigen_variant <dump.bin> [reps]
//   dump (little-endian doubles): n, K, var_img, reps_hint, P[n*n] col-major, then per track: L, chi2_095(2L-3), nan_flag,
//   jac[2L*n] col-major, hf[2L*3] col-major, res[2L]; finally P_oracle[n*n], corr_oracle[n] (what xo_visual_update gives).
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#if __has_include(<Eigen/Dense>)
#include <Eigen/Dense>
using Mat = Eigen::MatrixXd;
using Vec = Eigen::VectorXd;

int main(int argc, char **argv) {
  if (argc < 2) { std::puts("{\"eigen\": \"present\", \"error\": \"no dump file\"}"); return 1; }
  FILE *f = std::fopen(argv[1], "rb");
  if (!f) { std::puts("{\"eigen\": \"present\", \"error\": \"cannot open the dump\"}"); return 1; }
  std::fseek(f, 0, SEEK_END);
  const long bytes = std::ftell(f);
  std::fseek(f, 0, SEEK_SET);
  std::vector<double> d((size_t)bytes / 8);
  if (std::fread(d.data(), 8, d.size(), f) != d.size()) { std::puts("{\"eigen\": \"present\", \"error\": \"short read\"}"); return 1; }
  std::fclose(f);
  size_t at = 0;
  const int n = (int)d[at++], K = (int)d[at++];
  const double var = d[at++];
  int reps = (int)d[at++];
  if (argc > 2) reps = std::atoi(argv[2]);
  const Mat P0 = Eigen::Map<const Mat>(d.data() + at, n, n);
  at += (size_t)n * n;
  struct Trk { int L; double chi; bool nan; Mat jac, hf; Vec res; };
  std::vector<Trk> trk(K);
  for (int k = 0; k < K; ++k) {
    Trk &t = trk[k];
    t.L = (int)d[at++]; t.chi = d[at++]; t.nan = d[at++] != 0.0;
    const int m = 2 * t.L;
    t.jac = Eigen::Map<const Mat>(d.data() + at, m, n); at += (size_t)m * n;
    t.hf = Eigen::Map<const Mat>(d.data() + at, m, 3); at += (size_t)m * 3;
    t.res = Eigen::Map<const Vec>(d.data() + at, m); at += (size_t)m;
  }
  const Mat P_ref = Eigen::Map<const Mat>(d.data() + at, n, n);
  at += (size_t)n * n;
  const Vec c_ref = Eigen::Map<const Vec>(d.data() + at, n);
  Mat P = P0;
  Vec corr = Vec::Zero(n);
  int inliers = 0;
  double best = 1e300;
  for (int rep = 0; rep < reps; ++rep) {
    const auto t0 = std::chrono::steady_clock::now();
    P = P0;
    // ---- a9: every track (MsckfUpdate ctor pre-sizes for all tracks, rejected ones leave zero rows: msckf_update.cpp:27-63)
    int rows_all = 0;
    for (const Trk &t : trk) rows_all += 2 * t.L - 3;
    Mat h = Mat::Zero(rows_all, n);
    Vec res = Vec::Zero(rows_all);
    int row = 0;
    inliers = 0;
    for (const Trk &t : trk) {
      if (t.nan) continue;
      const int m = 2 * t.L, dd = m - 3;
      Eigen::HouseholderQR<Mat> qr(t.hf);
      const Mat Q = qr.householderQ();                         // full 2L x 2L (msckf_update.cpp:423-424)
      const Mat A = Q.rightCols(dd);
      const Vec res0 = A.transpose() * t.res;
      const Mat jac0 = A.transpose() * t.jac;
      Mat S = jac0 * P0 * jac0.transpose();
      S.diagonal().array() += var;
      const double gamma = res0.dot(S.inverse() * res0);       // general inverse, as written (:457)
      if (gamma < t.chi) {
        h.middleRows(row, dd) = jac0;
        res.segment(row, dd) = res0;
        row += dd;
        ++inliers;
      }
    }
    // ---- a11 (rows of rejected tracks stay zero rows of h, as in the reference)
    Mat H = h;
    Vec z = res;
    if (h.rows() > h.cols() + 1) {
      Mat aug(h.rows(), n + 1);
      aug << h, res;
      Eigen::HouseholderQR<Mat> qr(aug);
      const Mat R = qr.matrixQR().topRows(n + 1).template triangularView<Eigen::Upper>();
      H = R.topLeftCorner(n, n);
      z = R.topRightCorner(n, 1);
    }
    // ---- a12 (correction_total = 0)
    const Mat S = H * P * H.transpose() + var * Mat::Identity(H.rows(), H.rows());
    const Mat Kg = P * H.transpose() * S.inverse();
    corr = Kg * z;
    P = (Mat::Identity(n, n) - Kg * H) * P;
    P = 0.5 * (P + P.transpose().eval());
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (dt < best) best = dt;
  }
  const double relP = (P - P_ref).norm() / P_ref.norm(), relc = (corr - c_ref).norm() / c_ref.norm();
  std::printf("{\"eigen\": \"present\", \"version\": \"%d.%d.%d\", \"ms_per_update\": %.4f, \"updates_per_s\": %.4f, \"reps\": %d, "
              "\"inliers\": %d, \"rel_dP_vs_c_restatement\": %.3e, \"rel_dcorr_vs_c_restatement\": %.3e}\n",
              EIGEN_WORLD_VERSION, EIGEN_MAJOR_VERSION, EIGEN_MINOR_VERSION, 1e3 * best, 1.0 / best, reps, inliers, relP, relc);
  return 0;
}
#else
int main() {
  std::puts("{\"eigen\": \"absent\", \"note\": \"<Eigen/Dense> is not on this box's include path: the true-Eigen variant of a9 / a11 / a12 was not built\"}");
  return 0;
}
#endif
