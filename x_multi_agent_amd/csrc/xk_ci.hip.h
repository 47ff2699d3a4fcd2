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
  // arith != 0: the blocks are the position and attitude blocks of window poses [p0, p0 + L) -- first columns
  // XK_CORE + 3 pos and XK_CORE + 3 N + 3 pos -- and the factor is wval: nothing to fetch, nothing to stage from the host
  int arith, p0, L, N;
  double wval;
};
__global__ void xk_scale_blocks(XkScaleArgs a) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)a.n * a.n) return;
  const int r = (int)(idx % a.n), c = (int)(idx / a.n);
  double v = a.P[idx];
  if (a.arith) {
    const int rb = (r - 15) / 3, cb = (c - 15) / 3;                 // 3 x 3 block coordinates behind the core states
    if (r >= 15 && c >= 15 && rb == cb) {
      const int pos = rb < a.N ? rb : rb - a.N;                     // position blocks, then attitude blocks
      if (rb < 2 * a.N && pos >= a.p0 && pos < a.p0 + a.L) v *= a.wval;
    }
  } else {
    for (int b = 0; b < a.nblk; ++b) {
      const int c0 = a.cols[b];
      if (r >= c0 && r < c0 + 3 && c >= c0 && c < c0 + 3) v *= *a.w;
    }
  }
  a.Pj[idx] = v;
}

// ----------------------------------------------------------------------------
// MSCKF-MSCKF CI block (msckf_update.cpp:96-279).
// ----------------------------------------------------------------------------
// Triangulation of ONE landmark from the concatenated observations of several agents
// (matched agents first, self last, msckf_update.cpp:113-149,160-165).  One wave.
struct XkTriMultiArgs {
  const double *q, *p, *obs;  // [Ltot][4], [Ltot][3], [Ltot][2]
  int Ltot;
  double *gpf;                // [3]
  int *iters;
  // batch mode (one block per track of ONE agent, Triangulation::triangulateGN on the last L poses of the
  // window): trk_off [K+1] non-null, q / p = the n_poses window lists, obs = the concatenated tracks
  const int *trk_off;
  int n_poses;
};

__global__ __launch_bounds__(64) void xk_triangulate_multi(XkTriMultiArgs a_in) {
  XkTriMultiArgs a = a_in;
  if (a_in.trk_off) {
    const int b = blockIdx.x, o = a_in.trk_off[b];
    a.Ltot = a_in.trk_off[b + 1] - o;
    a.q = a_in.q + 4 * (size_t)(a_in.n_poses - a.Ltot);
    a.p = a_in.p + 3 * (size_t)(a_in.n_poses - a.Ltot);
    a.obs = a_in.obs + 2 * (size_t)o;
    a.gpf = a_in.gpf + 3 * (size_t)b;
    a.iters = a_in.iters + b;
  }
  const int lane = threadIdx.x, L = a.Ltot;
  double Ra[9], R1[9];
  xk_quat_to_rot(a.q + 4 * (size_t)(L - 1), Ra);
  xk_quat_to_rot(a.q, R1);
  const double *pa = a.p + 3 * (size_t)(L - 1), *p1 = a.p;
  double alpha, beta, rho;
  {
    double A4[4][4], X[4], P1[3][4], P2[3][4];
    for (int r = 0; r < 3; ++r) {
      double t1 = 0, t2 = 0;
      for (int c = 0; c < 3; ++c) {
        P1[r][c] = R1[3 * c + r];
        P2[r][c] = Ra[3 * c + r];
        t1 -= R1[3 * c + r] * p1[c];
        t2 -= Ra[3 * c + r] * pa[c];
      }
      P1[r][3] = t1;
      P2[r][3] = t2;
    }
    const double o1x = a.obs[0], o1y = a.obs[1], o2x = a.obs[2 * (size_t)(L - 1)], o2y = a.obs[2 * (size_t)(L - 1) + 1];
    for (int c = 0; c < 4; ++c) {
      A4[0][c] = o1x * P1[2][c] - P1[0][c];
      A4[1][c] = o1y * P1[2][c] - P1[1][c];
      A4[2][c] = o2x * P2[2][c] - P2[0][c];
      A4[3][c] = o2y * P2[2][c] - P2[1][c];
    }
    xk_null4(A4, X);
    const double wx = X[0] / X[3], wy = X[1] / X[3], wz = X[2] / X[3];
    double pc[3];
    for (int r = 0; r < 3; ++r) pc[r] = P2[r][0] * wx + P2[r][1] * wy + P2[r][2] * wz + P2[r][3];
    alpha = pc[0] / pc[2];
    beta = pc[1] / pc[2];
    rho = 1.0 / pc[2];
  }
  double r_norm_last = 1000.0, r_norm = 100.0;
  int iter = 0;
  bool ok = true;
  while (r_norm_last - r_norm > 1e-5) {
    iter++;
    if (iter > 10) break;
    double acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = lane; i < L; i += 64) {
      double Ri[9], drot[3][3], dpos[3];
      xk_quat_to_rot(a.q + 4 * (size_t)i, Ri);
      const double *pi = a.p + 3 * (size_t)i;
      for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) drot[r][c] = Ri[r] * Ra[c] + Ri[3 + r] * Ra[3 + c] + Ri[6 + r] * Ra[6 + c];
        dpos[r] = (Ri[r] * pa[0] + Ri[3 + r] * pa[1] + Ri[6 + r] * pa[2]) - (Ri[r] * pi[0] + Ri[3 + r] * pi[1] + Ri[6 + r] * pi[2]);
      }
      xk_gn_accum(drot, dpos, a.obs[2 * (size_t)i], a.obs[2 * (size_t)i + 1], alpha, beta, rho, acc);
    }
    for (int c = 0; c < 10; ++c) acc[c] = xk_wave_sum(acc[c]);
    double dl[3];
    if (!xk_solve3(acc, acc + 6, dl)) { ok = false; break; }
    alpha -= dl[0];
    beta -= dl[1];
    rho -= dl[2];
    r_norm_last = r_norm;
    r_norm = sqrt(acc[9]);
  }
  if (lane == 0) {
    for (int r = 0; r < 3; ++r)
      a.gpf[r] = ok ? (1.0 / rho) * (Ra[3 * r] * alpha + Ra[3 * r + 1] * beta + Ra[3 * r + 2]) + pa[r] : nan("");
    *a.iters = iter;
  }
}

// Null-space projection of the stacked column-space rows (nullSpaceProjection, msckf_update.cpp:494-501)
// and the split into per-agent Jacobians (:211-223).  up[i] = [3*n_i | 9 | 3] from xk_msckf_feature.
#define XK_CI_MAXK 7
struct XkCiProjArgs {
  int k1;                               // agents incl. self (k + 1)
  const double *up[XK_CI_MAXK + 1];
  int n[XK_CI_MAXK + 1];
  double *H[XK_CI_MAXK + 1];            // out: (3k) x n_i column-major, ld = 3k
  double *res;                          // out: 3k
};

__global__ __launch_bounds__(256) void xk_ci_project(XkCiProjArgs a) {
  __shared__ double Q[24][25];   // full Q of the 3(k+1) x 3 stack
  __shared__ double V[24][3];    // the stack, then its reflectors (below the diagonal)
  __shared__ double rp[24], taus[3];
  const int mr = 3 * a.k1, m = mr - 3, tid = threadIdx.x;
  if (tid < mr) {                // row tid of the stacked [Hf_i | res_i]
    const int i = tid / 3, r = tid % 3;
    const double *uh = a.up[i] + 3 * (size_t)a.n[i];
    for (int c = 0; c < 3; ++c) V[tid][c] = uh[r + 3 * c];
    rp[tid] = uh[9 + r];
  }
  for (int e = tid; e < 24 * 24; e += 256) Q[e / 24][e % 24] = (e / 24 == e % 24) ? 1.0 : 0.0;
  __syncthreads();
  if (tid == 0) {                // Householder QR of the mr x 3 stack (Eigen convention), in registers
    double jf[24][3], tau[3];
#pragma unroll
    for (int r = 0; r < 24; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) jf[r][c] = (r < mr) ? V[r][c] : 0.0;
#pragma unroll
    for (int kk = 0; kk < 3; ++kk) {
      double tail = 0;
#pragma unroll
      for (int r = kk + 1; r < 24; ++r) tail += jf[r][kk] * jf[r][kk];   // rows >= mr are zero
      const double c0 = jf[kk][kk];
      double bet, sc;
      if (tail <= 2.2250738585072014e-308) { tau[kk] = 0; bet = c0; sc = 0; }
      else { bet = sqrt(c0 * c0 + tail); if (c0 >= 0) bet = -bet; tau[kk] = (bet - c0) / bet; sc = 1.0 / (c0 - bet); }
#pragma unroll
      for (int r = kk + 1; r < 24; ++r) jf[r][kk] *= sc;
      jf[kk][kk] = bet;
#pragma unroll
      for (int c2 = kk + 1; c2 < 3; ++c2) {
        double w = jf[kk][c2];
#pragma unroll
        for (int r = kk + 1; r < 24; ++r) w += jf[r][kk] * jf[r][c2];
        w *= tau[kk];
        jf[kk][c2] -= w;
#pragma unroll
        for (int r = kk + 1; r < 24; ++r) jf[r][c2] -= w * jf[r][kk];
      }
    }
#pragma unroll
    for (int r = 0; r < 24; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) V[r][c] = jf[r][c];
    for (int kk = 0; kk < 3; ++kk) taus[kk] = tau[kk];
  }
  __syncthreads();
  for (int kk = 2; kk >= 0; --kk) {  // Q = H0 H1 H2 applied to I, one thread per column
    const double tk = taus[kk];
    if (tk != 0.0 && tid >= kk && tid < mr) {
      const int j = tid;
      double w = Q[kk][j];
      for (int r = kk + 1; r < mr; ++r) w += V[r][kk] * Q[r][j];
      w *= tk;
      Q[kk][j] -= w;
      for (int r = kk + 1; r < mr; ++r) Q[r][j] -= w * V[r][kk];
    }
    __syncthreads();
  }
  if (blockIdx.x == 0 && tid < m) {  // res = A^T res_pf, A = Q[:, 3:]
    double s = 0;
    for (int r = 0; r < mr; ++r) s += Q[r][3 + tid] * rp[r];
    a.res[tid] = s;
  }
  // one workgroup per agent: every one forms the (tiny) Q itself and writes its own agent's Jacobian (eight agents one after
  // the other, each behind its own round of load latency, were 16 of this kernel's 26 us)
  {                                    // H_i = A[3i:3i+3, :]^T * up_jac_i
    const int i = blockIdx.x;
    const double *uj = a.up[i];
    for (int col = tid; col < a.n[i]; col += 256) {
      const double u0 = uj[3 * (size_t)col], u1 = uj[3 * (size_t)col + 1], u2 = uj[3 * (size_t)col + 2];
      for (int c = 0; c < m; ++c)
        a.H[i][c + (size_t)m * col] = Q[3 * i][3 + c] * u0 + Q[3 * i + 1][3 + c] * u1 + Q[3 * i + 2][3 + c] * u2;
    }
  }
}

// gamma = res^T S^-1 res for a small SPD S (m <= 24), by Cholesky in one thread (msckf_update.cpp:243-244).
__global__ void xk_small_gamma(const double *S /*m x m col-major*/, const double *res, int m, double *gamma) {
  if (threadIdx.x || blockIdx.x) return;
  double L[24][24], y[24];
  bool bad = false;
  for (int i = 0; i < m; ++i)
    for (int j = 0; j <= i; ++j) {
      double s = S[i + (size_t)m * j];
      for (int t = 0; t < j; ++t) s -= L[i][t] * L[j][t];
      if (i == j) { if (!(s > 0)) bad = true; L[i][i] = sqrt(s); }
      else L[i][j] = s / L[j][j];
    }
  double g = 0;
  for (int i = 0; i < m; ++i) {
    double s = res[i];
    for (int t = 0; t < i; ++t) s -= L[i][t] * y[t];
    y[i] = s / L[i][i];
    g += y[i] * y[i];
  }
  *gamma = bad ? INFINITY : g;
}

// ----------------------------------------------------------------------------
// Device-resident CI round (the received SimpleState payloads never leave HBM): the same
// MSCKF-MSCKF CI block as above, with every per-agent stage batched over the agents.
// ----------------------------------------------------------------------------
// Concatenated pose / observation lists for the joint triangulation (matched agents first, self
// last, msckf_update.cpp:113-149): agent i contributes the last L_i poses of its window.
struct XkCiGatherArgs {
  int k1;
  const double *q[XK_CI_MAXK + 1], *p[XK_CI_MAXK + 1], *obs[XK_CI_MAXK + 1];
  int np[XK_CI_MAXK + 1], L[XK_CI_MAXK + 1];
  double *dq, *dp, *dobs;
};
// one workgroup per agent (eight agents one after the other on one wave were eight rounds of load latency: 13 us)
__global__ __launch_bounds__(64) void xk_ci_gather(XkCiGatherArgs a) {
  const int i = blockIdx.x;
  int at = 0;
  for (int r = 0; r < i; ++r) at += a.L[r];
  const int L = a.L[i], p0 = a.np[i] - L;
  for (int t = threadIdx.x; t < L; t += 64) {
    for (int c = 0; c < 4; ++c) a.dq[4 * (size_t)(at + t) + c] = a.q[i][4 * (size_t)(p0 + t) + c];
    for (int c = 0; c < 3; ++c) a.dp[3 * (size_t)(at + t) + c] = a.p[i][3 * (size_t)(p0 + t) + c];
    for (int c = 0; c < 2; ++c) a.dobs[2 * (size_t)(at + t) + c] = a.obs[i][2 * (size_t)t + c];
  }
}

// S_i = H_i P_i H_i^T for every agent at once: block (i, cb) owns 32 columns of W = H_i P_i and their
// contribution W[:, cols] H_i[:, cols]^T to S_i (partials are summed in a fixed order by xk_ci_combine).
// H_i is m x n_i (column-major, ld = m, m = 3k <= 21), P_i the agent's n_i x n_i covariance (symmetric:
// row j is read as column j, coalesced).  Thread (c, g) accumulates rows 3g..3g+2 of column c.
#define XK_CI_CHUNK 32
#define XK_CI_MAXCHUNK 16   // n <= 512
struct XkCiHphArgs {
  int m;
  const double *H[XK_CI_MAXK + 1], *P[XK_CI_MAXK + 1];
  int n[XK_CI_MAXK + 1];
  double *S;   // out: [k1][XK_CI_MAXCHUNK][24 * 24] column-major m x m partials
};
__global__ __launch_bounds__(256) void xk_ci_hph(XkCiHphArgs a) {
  extern __shared__ __attribute__((aligned(16))) double hsm[];
  const int i = blockIdx.x, cb = blockIdx.y, m = a.m, n = a.n[i], tid = threadIdx.x;
  double *Hs = hsm;                       // m x n, ld = m
  double *Ws = hsm + (size_t)m * n;       // 24 x 32 chunk of W, ld = 25
  double *out = a.S + ((size_t)i * XK_CI_MAXCHUNK + cb) * 576;
  const int c0 = cb * XK_CI_CHUNK;
  if (c0 >= n) {                          // (grid.y is sized for the widest agent)
    for (int e = tid; e < m * m; e += 256) out[e] = 0.0;
    return;
  }
  const double *H = a.H[i], *P = a.P[i];
  for (int e = tid; e < m * n; e += 256) Hs[e] = H[e];
  __syncthreads();
  const int cl = tid & 31, g = tid >> 5, c = c0 + cl;   // 8 row groups of 3
  double a0 = 0.0, a1 = 0.0, a2 = 0.0;
  if (c < n && 3 * g < m) {
    for (int j0 = 0; j0 < n; j0 += 16) {   // 16 covariance entries in flight per thread
      double pj[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) pj[u] = (j0 + u < n) ? P[(size_t)c + (size_t)(j0 + u) * n] : 0.0;   // = P(j, c)
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int j = (j0 + u < n) ? j0 + u : 0;
        const double *hj = Hs + (size_t)j * m + 3 * g;
        a0 = fma(hj[0], pj[u], a0);
        a1 = fma(hj[1], pj[u], a1);
        a2 = fma(hj[2], pj[u], a2);
      }
    }
  }
  Ws[(3 * g) * 33 + cl] = a0; Ws[(3 * g + 1) * 33 + cl] = a1; Ws[(3 * g + 2) * 33 + cl] = a2;
  __syncthreads();
  // partial S = W[:, chunk] H[:, chunk]^T
  const int nc = min(XK_CI_CHUNK, n - c0);
  for (int e = tid; e < m * m; e += 256) {
    const int r = e % m, s2 = e / m;
    double acc = 0.0;
    for (int cc = 0; cc < nc; ++cc) acc = fma(Ws[r * 33 + cc], Hs[s2 + (size_t)(c0 + cc) * m], acc);
    out[e] = acc;
  }
}

// S_gate = sum_i S_i + sigma^2 I (msckf_update.cpp:217-237), S_ci = S_0 / w0 + sum_{i>0} S_i / w + sigma^2 I
// (ci.cpp:78-85 + msckf_update.cpp:255), gamma = res^T S_gate^-1 res (:243-244).
struct XkCiCombineArgs {
  int k1, m;
  const double *S;      // [k1][XK_CI_MAXCHUNK][576] partials from xk_ci_hph
  int nchunk;
  double w0, w, var_img;
  const double *res;    // m
  double *S_gate, *S_ci;  // m x m column-major
  double *gamma;
  // optional: the two gate results of the track written straight into pinned host memory -- host_gate[0] = own chi-square
  // verdict (*own_inlier), host_gate[1] = joint gamma -- and then host_seq into *host_marker (system-scope release): a host that
  // polls the marker has the decisions without a copy or the runtime's completion signal
  const int *own_inlier;
  double *host_gate;
  unsigned long long *host_marker;
  unsigned long long host_seq;
};
// 512 threads: one entry of the m x m matrices per thread for the sum over agents and chunks (k1 * nchunk
// dependent-free loads each; on 64 threads this loop was 100 of the kernel's 110 us), then the small Cholesky.
__global__ __launch_bounds__(512) void xk_ci_combine(XkCiCombineArgs a) {
  const int m = a.m;
  for (int e = threadIdx.x; e < m * m; e += 512) {
    double g = 0.0, c = 0.0;
    for (int i = 0; i < a.k1; ++i) {
      // the chunk partials of agent i: all loads in flight at once, added in chunk order (a load per addition, one after the
      // other, was most of this kernel's 38 us)
      double part[XK_CI_MAXCHUNK];
#pragma unroll
      for (int cb = 0; cb < XK_CI_MAXCHUNK; ++cb) part[cb] = (cb < a.nchunk) ? a.S[((size_t)i * XK_CI_MAXCHUNK + cb) * 576 + e] : 0.0;
      double v = 0.0;
#pragma unroll
      for (int cb = 0; cb < XK_CI_MAXCHUNK; ++cb)
        if (cb < a.nchunk) v += part[cb];
      g += v;
      c += v * (i == 0 ? 1.0 / a.w0 : 1.0 / a.w);
    }
    if (e % m == e / m) { g += a.var_img; c += a.var_img; }
    a.S_gate[e] = g;
    a.S_ci[e] = c;
  }
  __syncthreads();
  // gamma = res^T S_gate^-1 res: right-looking Cholesky in LDS, lane i owns row i; then a column-oriented forward substitution.
  // m <= 24: ONE wave does it, with wave barriers (twenty-one steps of two 512-thread barriers each were most of this kernel)
  __shared__ double A[24][25], y[24], invd[24];
  const int t = threadIdx.x;
  if (t >= 64) return;
  auto wsync = [] { __builtin_amdgcn_s_waitcnt(0xc07f); __builtin_amdgcn_wave_barrier(); };
  for (int e = t; e < m * m; e += 64) A[e % m][e / m] = a.S_gate[e];
  if (t < m) y[t] = a.res[t];
  wsync();
  bool bad = false;
  for (int k = 0; k < m; ++k) {
    const double p = A[k][k];
    if (!(p > 0)) bad = true;
    // 1 / sqrt(p) from the hardware seed and two Newton steps, as every other pivot chain here (the IEEE sqrt and division
    // expansions were half of a step's 1400 clocks)
    double inv = __builtin_amdgcn_rsq(p);
    inv = inv * fma(-0.5 * p * inv, inv, 1.5);
    inv = inv * fma(-0.5 * p * inv, inv, 1.5);
    if (t == k) invd[k] = inv;
    double lik = 0.0;
    if (t > k && t < m) { lik = A[t][k] * inv; A[t][k] = lik; }
    wsync();
    // trailing update A(r, j) -= L(r, k) L(j, k), k < j <= r, dealt over the 64 lanes (one lane per ROW walked its row with two
    // dependent LDS reads per entry: 20 entries x 21 steps was most of the chain)
    for (int r0 = k + 1; r0 < m; r0 += 8)          // 8 x 8 patches of the trailing triangle (no index divisions in the chain)
      for (int j0 = k + 1; j0 <= r0 + 7 && j0 < m; j0 += 8) {
        const int r = r0 + (t >> 3), j = j0 + (t & 7);
        if (r < m && j <= r) A[r][j] -= A[r][k] * A[j][k];
      }
    if (t == k) A[k][k] = p * inv;   // sqrt(p)
    wsync();
  }
  double g = 0.0;
  for (int i = 0; i < m; ++i) {
    const double yi = y[i] * invd[i];   // uniform (1 / L(i,i) from the factorisation)
    g += yi * yi;
    wsync();
    if (t > i && t < m) y[t] -= A[t][i] * yi;
    wsync();
  }
  if (t == 0) {
    const double gam = bad ? INFINITY : g;
    *a.gamma = gam;
    if (a.host_marker) {
      __hip_atomic_store(a.host_gate, (double)*a.own_inlier, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(a.host_gate + 1, gam, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(a.host_marker, a.host_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}
