// xk_slaminit.hip.h -- MSCKF-SLAM measurement rows and the column-space rows that initialise a persistent
// feature (gfx950).  SURVEY 8(f) rank 3:
//   MsckfSlamUpdate::processOneTrack   src/x/vio/msckf_slam_update.cpp:64-267
// A handful of tracks per frame go through this (at most the free feature slots), so the kernel favours
// plain dense algebra over the tricks of xk_msckf_feature: the projected rows are written straight into the
// track's tile, W = H0 P goes through a global scratch, and the gate reuses the 2-D LDS Cholesky.
#pragma once
#include "xk_feature.hip.h"

struct XkSlamInitArgs {
  const double *q, *p;           // window [n_poses][4 xyzw], [n_poses][3]
  int n_poses, n_poses_max;
  const int *trk_off;            // [K2+1]
  const double *obs;             // [sum L][2]
  const double *gpf;             // [K2][3] triangulated landmark (world frame), NaN = triangulation failed
  const double *P;               // n x n column-major
  int n, na;                     // error states, active columns (n - 15); the residual sits in tile column na
  double var_img;
  const double *chi95;
  double *A;                     // tiles of these tracks [K2][DB][C1P] row-major
  int DB, C1P;
  double *W;                     // scratch [K2][DB][na]
  int *tile_rows, *inlier;       // [K2]
  double *gamma;                 // [K2]
  double *H1;                    // [K2][3][n]   U^T h          (row-major per track)
  double *H2;                    // [K2][9]      U^T Hf = R     (column-major 3x3)
  double *r1;                    // [K2][3]      U^T res
  double *features;              // [K2][3]      (alpha, beta, rho) in the last pose
};

static inline size_t xk_slaminit_lds_bytes(int n_poses) {
  const int L = n_poses, m2 = 2 * L, ldm = m2 + 1;
  return sizeof(double) * (size_t)(12 * L + 24 * L + 3 * m2 + m2 + 3 * m2 + (size_t)(m2 + 1) * ldm + 64);
}

__global__ __launch_bounds__(XK_FEAT_THREADS) void xk_msckf_slam_init(XkSlamInitArgs a) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int tid = threadIdx.x, k = blockIdx.x, np = a.n_poses;
  const int off = a.trk_off[k], L = a.trk_off[k + 1] - off, m2 = 2 * L, d = m2 - 3, p0 = np - L;
  const int ldm = 2 * np + 1;
  double *rot = sm;                    // [np][9] camera->world
  double *pos = rot + 9 * np;          // [np][3]
  double *blk = pos + 3 * np;          // [L][4][6]: J_pos, J_att (own pose), J_anchor_pos, J_anchor_att, each 2x3 row-major
  double *hf = blk + 24 * np;          // [3][2L] Hf columns, then the reflectors
  double *res = hf + 3 * 2 * np;       // [2L]
  double *yv = res + 2 * np;           // [3][2L] scratch
  double *Mm = yv + 3 * 2 * np;        // [(2L+1)][ldm] gate matrix at offset (3,3), residual in row 2L
  double *scal = Mm + (size_t)(2 * np + 1) * ldm;   // 0..2 tau, 3 g01, 4 g02, 5 g12, 9 valid, 10 bad, 11 inlier, 12 gamma, 16..24 R
  const double chi_gate = a.chi95[d];
  for (int i = tid; i < np; i += XK_FEAT_THREADS) {
    xk_quat_to_rot(a.q + 4 * i, rot + 9 * i);
    for (int c = 0; c < 3; ++c) pos[3 * i + c] = a.p[3 * i + c];
  }
  if (tid == 0) { scal[9] = 1.0; scal[10] = 0.0; }
  __syncthreads();
  const double gx = a.gpf[3 * k], gy = a.gpf[3 * k + 1], gz = a.gpf[3 * k + 2];
  // inverse-depth parameters in the anchor (= last) pose (:82-99)
  const double *Rn = rot + 9 * (np - 1), *pn = pos + 3 * (np - 1);
  double al, be, rho;
  {
    const double dx = gx - pn[0], dy = gy - pn[1], dz = gz - pn[2];
    const double cx = Rn[0] * dx + Rn[3] * dy + Rn[6] * dz, cy = Rn[1] * dx + Rn[4] * dy + Rn[7] * dz;
    const double cz = Rn[2] * dx + Rn[5] * dy + Rn[8] * dz;
    al = cx / cz; be = cy / cz; rho = 1.0 / cz;
  }
  // ---- per-observation residual and Jacobian blocks (:102-201)
  for (int i = tid; i < L; i += XK_FEAT_THREADS) {
    const double *R = rot + 9 * (p0 + i), *pp = pos + 3 * (p0 + i);
    const double dx = gx - pp[0], dy = gy - pp[1], dz = gz - pp[2];
    const double c[3] = {R[0] * dx + R[3] * dy + R[6] * dz, R[1] * dx + R[4] * dy + R[7] * dz, R[2] * dx + R[5] * dy + R[8] * dz};
    if (!(c[0] == c[0] && c[1] == c[1] && c[2] == c[2])) scal[9] = 0.0;
    res[2 * i] = a.obs[2 * (size_t)(off + i)] - c[0] / c[2];
    res[2 * i + 1] = a.obs[2 * (size_t)(off + i) + 1] - c[1] / c[2];
    double *B = blk + 24 * i;
    if (i == L - 1) {                                       // special case (:133-142)
      for (int e = 0; e < 24; ++e) B[e] = 0.0;
      hf[2 * i] = 1.0; hf[m2 + 2 * i] = 0.0; hf[2 * m2 + 2 * i] = 0.0;
      hf[2 * i + 1] = 0.0; hf[m2 + 2 * i + 1] = 1.0; hf[2 * m2 + 2 * i + 1] = 0.0;
      continue;
    }
    const double Ji[2][3] = {{1.0 / c[2], 0.0, -c[0] / (c[2] * c[2])}, {0.0, 1.0 / c[2], -c[1] / (c[2] * c[2])}};
    double JRt[2][3], M3[3][3];                             // Ji R_i^T ; R_i^T R_n
    for (int r = 0; r < 2; ++r)
      for (int cc = 0; cc < 3; ++cc) JRt[r][cc] = Ji[r][0] * R[3 * cc] + Ji[r][1] * R[3 * cc + 1] + Ji[r][2] * R[3 * cc + 2];
    for (int r = 0; r < 3; ++r)
      for (int cc = 0; cc < 3; ++cc) M3[r][cc] = R[r] * Rn[cc] + R[3 + r] * Rn[3 + cc] + R[6 + r] * Rn[6 + cc];
    double JM[2][3];                                        // Ji R_i^T R_n
    for (int r = 0; r < 2; ++r)
      for (int cc = 0; cc < 3; ++cc) JM[r][cc] = Ji[r][0] * M3[0][cc] + Ji[r][1] * M3[1][cc] + Ji[r][2] * M3[2][cc];
    for (int r = 0; r < 2; ++r) {
      for (int cc = 0; cc < 3; ++cc) {
        B[3 * r + cc] = -JRt[r][cc];                        // J_position (:163-164)
        B[12 + 3 * r + cc] = JRt[r][cc];                    // J_anchor_pos = -J_position (:173)
      }
      // J_attitude = Ji Skew(c) (:156-160)
      B[6 + 3 * r] = Ji[r][1] * c[2] - Ji[r][2] * c[1];
      B[6 + 3 * r + 1] = -Ji[r][0] * c[2] + Ji[r][2] * c[0];
      B[6 + 3 * r + 2] = Ji[r][0] * c[1] - Ji[r][1] * c[0];
      // J_anchor_att = -1/rho Ji R_i^T R_n Skew(alpha, beta, 1) (:167-170)
      B[18 + 3 * r] = -(1.0 / rho) * (JM[r][1] * 1.0 - JM[r][2] * be);
      B[18 + 3 * r + 1] = -(1.0 / rho) * (-JM[r][0] * 1.0 + JM[r][2] * al);
      B[18 + 3 * r + 2] = -(1.0 / rho) * (JM[r][0] * be - JM[r][1] * al);
      // Hf = 1/rho Ji R_i^T R_n [[1,0,-al/rho],[0,1,-be/rho],[0,0,-1/rho]] (:176-183)
      hf[2 * i + r] = (1.0 / rho) * JM[r][0];
      hf[m2 + 2 * i + r] = (1.0 / rho) * JM[r][1];
      hf[2 * m2 + 2 * i + r] = (1.0 / rho) * (-JM[r][0] * al / rho - JM[r][1] * be / rho - JM[r][2] / rho);
    }
  }
  __syncthreads();
  // ---- Householder QR of Hf (2L x 3), Eigen convention (:206)
  if (tid < 64) {
    const int lane = tid;
    for (int kk = 0; kk < 3; ++kk) {
      double *col = hf + kk * m2;
      double tail = 0.0;
      for (int r = lane; r < m2; r += 64)
        if (r > kk) tail += col[r] * col[r];
      tail = xk_wave_sum(tail);
      const double c0 = col[kk];
      double tau, bet, sc;
      if (tail <= 2.2250738585072014e-308) { tau = 0.0; bet = c0; sc = 0.0; }
      else { bet = sqrt(c0 * c0 + tail); if (c0 >= 0) bet = -bet; tau = (bet - c0) / bet; sc = 1.0 / (c0 - bet); }
      for (int r = lane; r < m2; r += 64)
        if (r > kk) col[r] *= sc;
      if (lane == 0) scal[16 + 4 * kk] = bet;
      for (int c2 = kk + 1; c2 < 3; ++c2) {
        double *cc = hf + c2 * m2;
        double w = 0.0;
        for (int r = lane; r < m2; r += 64)
          if (r > kk) w += col[r] * cc[r];
        w = tau * (xk_wave_sum(w) + cc[kk]);
        for (int r = lane; r < m2; r += 64)
          if (r > kk) cc[r] -= w * col[r];
        if (lane == 0) { cc[kk] -= w; scal[16 + kk + 3 * c2] = cc[kk]; }
      }
      if (lane == 0) scal[kk] = tau;
    }
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) {   // explicit reflectors: v_k[k] = 1, v_k[r<k] = 0; R below the diagonal is zero
      hf[0] = 1.0;
      hf[m2] = 0.0; hf[m2 + 1] = 1.0;
      hf[2 * m2] = 0.0; hf[2 * m2 + 1] = 0.0; hf[2 * m2 + 2] = 1.0;
      scal[17] = 0.0; scal[18] = 0.0; scal[21] = 0.0;   // (1,0), (2,0), (2,1)
    }
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    double g01 = 0, g02 = 0, g12 = 0;
    for (int r = lane; r < m2; r += 64) {
      g01 += hf[r] * hf[m2 + r];
      g02 += hf[r] * hf[2 * m2 + r];
      g12 += hf[m2 + r] * hf[2 * m2 + r];
    }
    g01 = xk_wave_sum(g01); g02 = xk_wave_sum(g02); g12 = xk_wave_sum(g12);
    if (lane == 0) { scal[3] = g01; scal[4] = g02; scal[5] = g12; }
  }
  __syncthreads();
  const double tau0 = scal[0], tau1 = scal[1], tau2 = scal[2], g01 = scal[3], g02 = scal[4], g12 = scal[5];
  const double *V0 = hf, *V1 = hf + m2, *V2 = hf + 2 * m2;
  // y = Q^T x = H2 H1 H0 x in compact-WY form: y = x - V w,
  //   w0 = tau0 a0, w1 = tau1 (a1 - w0 g01), w2 = tau2 (a2 - w0 g02 - w1 g12),  a_k = v_k^T x
  // ---- Q^T [h | res] column by column: rows 0..2 -> H1 / r1, rows 3.. -> the tile (:210-233)
  double *tile = a.A + (size_t)k * a.DB * a.C1P;
  const int N3 = 3 * a.n_poses_max, anchor = np - 1;
  for (int c = tid; c < a.C1P; c += XK_FEAT_THREADS) {
    // column c of h over the active columns (state column 15 + c); c == na is the residual
    int kind = 0, pose = -1, comp = 0;           // 0: zero column, 1: residual, 2: pose column
    bool att = false;
    if (c == a.na) kind = 1;
    else if (c < N3) { kind = 2; pose = c / 3; comp = c % 3; }
    else if (c < 2 * N3) { kind = 2; pose = (c - N3) / 3; comp = (c - N3) % 3; att = true; }
    auto xval = [&](int r) -> double {
      if (kind == 1) return res[r];
      const int i = r >> 1, rr = r & 1;
      double v = 0.0;
      if (pose == p0 + i) v += blk[24 * i + (att ? 6 : 0) + 3 * rr + comp];          // own pose block
      if (pose == anchor) v += blk[24 * i + (att ? 18 : 12) + 3 * rr + comp];        // anchor block
      return v;
    };
    const bool live = kind == 1 || (kind == 2 && pose >= p0 && pose < np);
    double a0 = 0.0, a1 = 0.0, a2 = 0.0;
    if (live)
      for (int r = 0; r < m2; ++r) { const double x = xval(r); a0 += V0[r] * x; a1 += V1[r] * x; a2 += V2[r] * x; }
    const double w0 = tau0 * a0, w1 = tau1 * (a1 - w0 * g01), w2 = tau2 * (a2 - w0 * g02 - w1 * g12);
    for (int r = 0; r < m2; ++r) {
      const double y = live ? xval(r) - w0 * V0[r] - w1 * V1[r] - w2 * V2[r] : 0.0;
      if (r >= 3) tile[(size_t)(r - 3) * a.C1P + c] = (c <= a.na) ? y : 0.0;
      else if (kind == 1) a.r1[3 * (size_t)k + r] = y;
      else if (c < a.na) a.H1[((size_t)k * 3 + r) * a.n + XK_CORE + c] = y;
    }
  }
  for (int e = tid; e < 3 * XK_CORE; e += XK_FEAT_THREADS) a.H1[((size_t)k * 3 + e / XK_CORE) * a.n + e % XK_CORE] = 0.0;
  if (tid < 9) a.H2[9 * (size_t)k + tid] = scal[16 + tid];
  if (tid == 0) { a.features[3 * (size_t)k] = al; a.features[3 * (size_t)k + 1] = be; a.features[3 * (size_t)k + 2] = rho; }
  __threadfence_block();
  __syncthreads();
  // ---- S = H0 P H0^T + sigma^2 I through W = H0 P (:243), then the chi-square gate (:244-249)
  double *W = a.W + (size_t)k * a.DB * a.na;
  for (int e = tid; e < d * a.na; e += XK_FEAT_THREADS) {
    const int i = e / a.na, c = e % a.na;
    const double *hrow = tile + (size_t)i * a.C1P;
    const double *pc = a.P + (size_t)(XK_CORE + c) * a.n + XK_CORE;     // column c of the active block
    double s0 = 0.0, s1 = 0.0;
    int j = 0;
    for (; j + 1 < a.na; j += 2) { s0 = fma(hrow[j], pc[j], s0); s1 = fma(hrow[j + 1], pc[j + 1], s1); }
    if (j < a.na) s0 = fma(hrow[j], pc[j], s0);
    W[(size_t)i * a.na + c] = s0 + s1;
  }
  __threadfence_block();
  __syncthreads();
  for (int e = tid; e < d * d; e += XK_FEAT_THREADS) {
    const int i = e / d, j = e % d;
    if (j > i) continue;
    const double *wi = W + (size_t)i * a.na, *hj = tile + (size_t)j * a.C1P;
    double s0 = 0.0, s1 = 0.0;
    int c = 0;
    for (; c + 1 < a.na; c += 2) { s0 = fma(wi[c], hj[c], s0); s1 = fma(wi[c + 1], hj[c + 1], s1); }
    if (c < a.na) s0 = fma(wi[c], hj[c], s0);
    Mm[(size_t)(3 + i) * ldm + 3 + j] = s0 + s1 + (i == j ? a.var_img : 0.0);
  }
  for (int j = tid; j < d; j += XK_FEAT_THREADS) Mm[(size_t)(3 + d) * ldm + 3 + j] = tile[(size_t)j * a.C1P + a.na];
  __syncthreads();
  xk_chol_gate<8, false>(Mm, ldm, d, tid, scal);
  __syncthreads();
  if (tid == 0) {
    const double g = scal[12];
    const bool valid = scal[9] != 0.0 && gx == gx, bad = scal[10] != 0.0;
    const bool inl = valid && !bad && (g < chi_gate);
    a.gamma[k] = (valid && !bad) ? g : (valid ? INFINITY : nan(""));
    a.inlier[k] = inl ? 1 : 0;
    a.tile_rows[k] = inl ? d : 0;
  }
}
