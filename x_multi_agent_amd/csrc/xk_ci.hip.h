// xk_ci.hip.h -- covariance-intersection kernels (gfx950).
//   MultiSlamUpdate::processOneMatch   src/x/vio/multi_slam_update.cpp:61-246
//   CovarianceIntersection::fuseCI     src/x/ekf/ci.cpp:94-127 (pairwise form)
// The 3 x n Jacobians of a SLAM-SLAM match have nine non-zero columns per
// agent, so h P h^T is a 9x9 gather from P rather than a dense product.
#pragma once
#include <hip/hip_runtime.h>

#include "xk_feature.hip.h"

struct XkSlamMatchArgs {
  // own side
  const double *q, *p, *feat, *P;
  int anchor, fid, n, npm;
  // other side
  const double *oq, *op, *ofeat, *oP;
  int oanchor, ofid, no, onpm;
  double var_l, w, chi;  // sigma_landmark^2, ci_slam_w, chi2_3(0.9)
  // outputs
  double *H;    // 3 x n column-major (ld 3)
  double *out;  // [0..2] res, [3..11] S (col-major 3x3), [12] gamma, [13] inlier, [14] w_result
  int *cols;    // [3] first state column of the three scaled diagonal blocks
};

__device__ inline void xk_match_side(const double *q, const double *p, const double *f, int a, int fid, int npm,
                                     double sign, double gpf[3], int cols[3], double blk[3][3][3]) {
  const double al = f[3 * fid], be = f[3 * fid + 1], rho = f[3 * fid + 2];
  double Ra[9];
  xk_quat_to_rot(q + 4 * a, Ra);
  for (int k = 0; k < 3; ++k) gpf[k] = (1.0 / rho) * (Ra[3 * k] * al + Ra[3 * k + 1] * be + Ra[3 * k + 2]) + p[3 * a + k];
  const double sk[9] = {0, -1.0, be, 1.0, 0, -al, -be, al, 0};
  const double mat[9] = {1, 0, -al / rho, 0, 1, -be / rho, 0, 0, -1.0 / rho};
  cols[0] = XK_CORE + 3 * a;
  cols[1] = cols[0] + 3 * npm;
  cols[2] = XK_CORE + (2 * npm + fid) * 3;
  for (int x = 0; x < 3; ++x)
    for (int y = 0; y < 3; ++y) {
      const double rs = Ra[3 * x] * sk[y] + Ra[3 * x + 1] * sk[3 + y] + Ra[3 * x + 2] * sk[6 + y];
      const double rm = Ra[3 * x] * mat[y] + Ra[3 * x + 1] * mat[3 + y] + Ra[3 * x + 2] * mat[6 + y];
      blk[0][x][y] = sign * (x == y ? 1.0 : 0.0);  // anchor position
      blk[1][x][y] = sign * (-(1.0 / rho) * rs);   // anchor attitude
      blk[2][x][y] = sign * ((1.0 / rho) * rm);    // inverse-depth feature
    }
}

__device__ inline void xk_hPht(const double *P, int n, const int cols[3], const double blk[3][3][3], double S[9]) {
  for (int i = 0; i < 9; ++i) S[i] = 0.0;
  for (int b1 = 0; b1 < 3; ++b1)
    for (int b2 = 0; b2 < 3; ++b2)
      for (int c1 = 0; c1 < 3; ++c1)
        for (int c2 = 0; c2 < 3; ++c2) {
          const double pv = P[(size_t)(cols[b1] + c1) + (size_t)(cols[b2] + c2) * n];
          for (int x = 0; x < 3; ++x)
            for (int y = 0; y < 3; ++y) S[x + 3 * y] += blk[b1][x][c1] * pv * blk[b2][y][c2];
        }
}

__global__ __launch_bounds__(64) void xk_slam_match(XkSlamMatchArgs a) {
  __shared__ double blk[3][3][3];
  __shared__ int cols[3];
  const int lane = threadIdx.x;
  if (lane == 0) {
    double gpf[3], ogpf[3], oblk[3][3][3], mb[3][3][3];
    int ocols[3], mc[3];
    xk_match_side(a.q, a.p, a.feat, a.anchor, a.fid, a.npm, +1.0, gpf, mc, mb);
    xk_match_side(a.oq, a.op, a.ofeat, a.oanchor, a.ofid, a.onpm, -1.0, ogpf, ocols, oblk);
    double Pa[9], Pb[9], S0[9], res[3];
    xk_hPht(a.P, a.n, mc, mb, Pa);
    xk_hPht(a.oP, a.no, ocols, oblk, Pb);
    for (int k = 0; k < 3; ++k) res[k] = -gpf[k] + ogpf[k];  // :131
    for (int i = 0; i < 9; ++i) S0[i] = Pa[i] + Pb[i];
    S0[0] += a.var_l; S0[4] += a.var_l; S0[8] += a.var_l;
    // gamma = res^T S0^-1 res via 3x3 adjugate
    const double c00 = S0[4] * S0[8] - S0[5] * S0[7], c01 = S0[5] * S0[6] - S0[3] * S0[8], c02 = S0[3] * S0[7] - S0[4] * S0[6];
    const double det = S0[0] * c00 + S0[1] * c01 + S0[2] * c02;
    double inv[9];
    inv[0] = c00 / det; inv[1] = (S0[2] * S0[7] - S0[1] * S0[8]) / det; inv[2] = (S0[1] * S0[5] - S0[2] * S0[4]) / det;
    inv[3] = c01 / det; inv[4] = (S0[0] * S0[8] - S0[2] * S0[6]) / det; inv[5] = (S0[2] * S0[3] - S0[0] * S0[5]) / det;
    inv[6] = c02 / det; inv[7] = (S0[1] * S0[6] - S0[0] * S0[7]) / det; inv[8] = (S0[0] * S0[4] - S0[1] * S0[3]) / det;
    double g = 0.0;
    for (int x = 0; x < 3; ++x)
      for (int y = 0; y < 3; ++y) g += res[x] * inv[x + 3 * y] * res[y];
    const bool inl = g < a.chi;  // :216-220
    // fuseCI pairwise (ci.cpp:120-122) + noise (:226)
    const double wr = 1.0 / (1.0 - a.w);
    for (int i = 0; i < 9; ++i) a.out[3 + i] = wr * Pa[i] + (1.0 / a.w) * Pb[i];
    a.out[3] += a.var_l; a.out[7] += a.var_l; a.out[11] += a.var_l;
    for (int k = 0; k < 3; ++k) a.out[k] = res[k];
    a.out[12] = g;
    a.out[13] = inl ? 1.0 : 0.0;
    a.out[14] = wr;
    for (int b = 0; b < 3; ++b) {
      cols[b] = mc[b];
      a.cols[b] = mc[b];
      for (int x = 0; x < 3; ++x)
        for (int y = 0; y < 3; ++y) blk[b][x][y] = mb[b][x][y];
    }
  }
  __syncthreads();
  for (int c = lane; c < a.n; c += 64) {
    double v[3] = {0, 0, 0};
    for (int b = 0; b < 3; ++b)
      if (c >= cols[b] && c < cols[b] + 3)
        for (int x = 0; x < 3; ++x) v[x] = blk[b][x][c - cols[b]];
    a.H[3 * (size_t)c] = v[0];
    a.H[3 * (size_t)c + 1] = v[1];
    a.H[3 * (size_t)c + 2] = v[2];
  }
}

// P_j = P with the listed 3x3 DIAGONAL blocks scaled by w (cross terms untouched, SURVEY Q7;
// msckf_update.cpp:256-267, multi_slam_update.cpp:229-239).
struct XkScaleArgs {
  const double *P;
  double *Pj;
  int n, nblk;
  const int *cols;   // device: first column of each block
  const double *w;   // device scalar (w_result)
};
__global__ void xk_scale_blocks(XkScaleArgs a) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)a.n * a.n) return;
  const int r = (int)(idx % a.n), c = (int)(idx / a.n);
  double v = a.P[idx];
  for (int b = 0; b < a.nblk; ++b) {
    const int c0 = a.cols[b];
    if (r >= c0 && r < c0 + 3 && c >= c0 && c < c0 + 3) v *= *a.w;
  }
  a.Pj[idx] = v;
}
