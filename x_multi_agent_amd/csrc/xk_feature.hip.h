// xk_feature.hip.h -- per-feature MSCKF build and SLAM-row kernels (gfx950).
//
// One workgroup per MSCKF track (a5..a9 of SURVEY 8a):
//   triangulation   src/x/vision/triangulation.cpp:48-206
//   Jacobians + observability constraint   src/x/vio/msckf_update.cpp:328-417
//   left-nullspace projection               msckf_update.cpp:423-432
//   chi-square gate                         msckf_update.cpp:452-463
// Output per inlier track: the d = 2L-3 projected rows [H0 | res0] over the
// ACTIVE columns (state columns 15.., the 15 core columns are identically
// zero, msckf_update.cpp:412-416) as one row-major tile for the QR compression.
//
// What is exploited that the reference does not: J is block-sparse (one 2x3
// position block and one 2x3 attitude block per observation), so
//   J P J^T      is built from 6x6 blocks of P (never the dense d x n product),
//   A^T(.)A      is applied as three Householder reflectors on both sides,
//   A^T J        is formed column by column from the reflectors (WY form).
// All arithmetic is IEEE double; the results differ from the reference only
// by summation order (basis choice of A is immaterial).
#pragma once
#include <hip/hip_runtime.h>

#include "xk_chol16.hip.h"

#define XK_CORE 15
typedef double xk_f2 __attribute__((ext_vector_type(2)));
#define XK_FEAT_THREADS 256
#ifdef XK_FEAT_PROBE
#define XK_STAMP(i) do { __syncthreads(); if (a.dbg && threadIdx.x == 0) a.dbg[12 * (size_t)blockIdx.x + i] = clock64(); } while (0)
// per-workgroup record dbg[12k ..]: 8 phase stamps (clock64), wall-clock start/end, placement (XCC_ID, HW_ID)
#define XK_WG_BEGIN() const long long xk_w0 = wall_clock64()
#define XK_WG_END() do { if (a.dbg && threadIdx.x == 0) { long long *e = a.dbg + 12 * (size_t)blockIdx.x + 8; e[0] = xk_w0; e[1] = wall_clock64(); \
    unsigned xcc, hw; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc)); asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw)); \
    e[2] = ((long long)xcc << 32) | hw; } } while (0)
#else
#define XK_STAMP(i)
#define XK_WG_BEGIN()
#define XK_WG_END()
#endif

struct XkFeatArgs {
  const double *q;   // [n_poses][4] xyzw
  const double *p;   // [n_poses][3]
  int n_poses, n_poses_max;
  const int *trk_off;  // [K+1]
  const double *obs;   // [sum L][2]
  int K;
  const double *P;  // n x n column-major, ld = n
  int n;
  double var_img;
  const double *chi95;  // chi-square 0.95 quantile, indexed by dof
  double *A;            // tiles [ntiles][DB][C1P] row-major
  int DB, C1P, na;      // na active columns, residual in column na
  // FACTOR RECORDS instead of the tile (Hc != nullptr; 64-row slots only): what the rows 3.. of Q^T [J | res] are made of --
  // per row {v0, v1, v2, r'} (XK_HC_VR doubles), per column {w0, w1, w2, x0, x1, r0} (xk_h0_entry) -- hs doubles per track, a tenth
  // of its tile.  xk_caqr_pipe's tile workgroups and the first pass of the multi-launch schedule (xk_caqr_tile, panel 0) form the
  // entries from them.
  double *Hc;
  int hs, hcvr;         // doubles per record / of its per-row part (xk_hc_vr(DB))
  int *tile_rows;       // rows of tile k that hold data (0 = skip)
  int *inlier;
  double *gamma;
  int *inlier_h;     // optional mirrors of the two in PINNED HOST memory (device-visible address): the host reads the gate
  double *gamma_h;   // results in place after its next synchronisation instead of queueing copies behind an event
  double *gpf;       // [K][3] triangulated landmark, world frame
  int *gn_iters;     // [K]
  // multi-agent MSCKF (msckf_update.cpp:201-203,439-443): use a landmark triangulated elsewhere and
  // also emit the 3 column-space rows  A_up^T [jac | Hf | res]  (up_out: [3*n | 9 | 3] doubles, col-major)
  const double *gpf_in;
  double *up_out;
  // batch mode (CI round): block b works on its OWN window / prior / track, described by batch[b]
  const struct XkFeatBatch *batch;
  long long *dbg;   // probe builds only: phase stamps of block 0
};

// One agent's view of a shared track: window, prior covariance, observations (all device pointers).
struct XkFeatBatch {
  const double *q, *p, *obs, *P;
  int n_poses, n_poses_max, n, L;
  double *up_out;
};

// Entry (row r, column c) of Q^T [J | res], Q = H0 H1 H2 = I - V T V^T, from the track's factor record: column c of [J | res]
// has two non-zero entries x0, x1 in rows r0, r0 + 1 (the observation of c's pose), w = T^T V^T J[:, c]; r0 = -1: the residual
// column (the entry is r'[r]), r0 = -2: a column the track does not touch.  (msckf_update.cpp:423-432)
#define XK_HC_ROWS 68               // rows a record holds next to 64-row slots (2 L <= 66); DB + 4 in general (xk_hc_vr)
#define XK_HC_VR (4 * XK_HC_ROWS)   // doubles of its per-row part {v0, v1, v2, r'} there; the per-column part follows,
#define XK_HC_WC 6                  // doubles per column: {w0, w1, w2, x0, x1, r0}
static inline int xk_hc_vr(int DB) { return 4 * (DB + 4); }
static inline int xk_hc_stride(int DB, int C1P) { return xk_hc_vr(DB) + XK_HC_WC * C1P; }
__device__ __forceinline__ double xk_h0_entry(double w0, double w1, double w2, double x0, double x1, int r0, double v0, double v1, double v2,
                                              double rres, int r) {
  double v = -w0 * v0 - w1 * v1 - w2 * v2;
  if (r == r0) v += x0;
  if (r == r0 + 1) v += x1;
  return (r0 == -1) ? rres : (r0 == -2 ? 0.0 : v);
}

__device__ __forceinline__ void xk_quat_to_rot(const double *q, double *r /*row-major 3x3*/) {
  // q.normalized().toRotationMatrix(), camera -> world (msckf_update.cpp:339)
  const double nn = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  const double x = q[0] / nn, y = q[1] / nn, z = q[2] / nn, w = q[3] / nn;
  const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
  const double twx = tx * w, twy = ty * w, twz = tz * w;
  const double txx = tx * x, txy = ty * x, txz = tz * x;
  const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
  r[0] = 1 - (tyy + tzz); r[1] = txy - twz;       r[2] = txz + twy;
  r[3] = txy + twz;       r[4] = 1 - (txx + tzz); r[5] = tyz - twx;
  r[6] = txz - twy;       r[7] = tyz + twx;       r[8] = 1 - (txx + tyy);
}

template <int CTRL>
__device__ __forceinline__ double xk_dpp_f64(double x) {
  const long long q = __builtin_bit_cast(long long, x);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)(q & 0xffffffffLL), CTRL, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(q >> 32), CTRL, 0xf, 0xf, true);
  return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}
// All-lanes sum over the wave: four DPP stages inside each 16-lane row (quad xor 1, quad xor 2,
// half-row mirror, row mirror), then two cross-row lane swaps.  Every lane gets the same value.
__device__ __forceinline__ double xk_wave_sum(double v) {
  v += xk_dpp_f64<0xB1>(v);    // quad_perm [1,0,3,2]
  v += xk_dpp_f64<0x4E>(v);    // quad_perm [2,3,0,1]
  v += xk_dpp_f64<0x141>(v);   // row_half_mirror
  v += xk_dpp_f64<0x140>(v);   // row_mirror
  // across the four rows: v_permlane16_swap (rows 2i <-> 2i+1) and v_permlane32_swap (halves) on two copies of
  // the value -- a + b is the pair sum in both partners; VALU only, no trip through the LDS crossbar
  {
    const long long q = __builtin_bit_cast(long long, v);
    const unsigned lo = (unsigned)q, hi = (unsigned)(q >> 32);
    const auto r0 = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    const auto r1 = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    v = __builtin_bit_cast(double, ((long long)r1[0] << 32) | (unsigned int)r0[0]) +
        __builtin_bit_cast(double, ((long long)r1[1] << 32) | (unsigned int)r0[1]);
  }
  {
    const long long q = __builtin_bit_cast(long long, v);
    const unsigned lo = (unsigned)q, hi = (unsigned)(q >> 32);
    const auto r0 = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    const auto r1 = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    v = __builtin_bit_cast(double, ((long long)r1[0] << 32) | (unsigned int)r0[0]) +
        __builtin_bit_cast(double, ((long long)r1[1] << 32) | (unsigned int)r0[1]);
  }
  return v;
}

// Smallest right singular vector of a 4x4 (row-major) by one-sided Jacobi;
// stands in for cv::SVD inside cv::triangulatePoints (triangulation.cpp:93).
__device__ inline void xk_null4(double a[4][4], double x[4]) {
  double v[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) v[i][j] = (i == j) ? 1.0 : 0.0;
  // Reciprocals and inverse square roots from the hardware seeds + two Newton steps: the IEEE division /
  // sqrt expansions were most of a rotation's ~180 instructions, and a wave issues one per 4 clocks.
  auto rsq = [](double z) { double y = __builtin_amdgcn_rsq(z); y = y * fma(-0.5 * z * y, y, 1.5); return y * fma(-0.5 * z * y, y, 1.5); };
  auto rcp = [](double z) { double y = __builtin_amdgcn_rcp(z); y = fma(y, fma(-z, y, 1.0), y); return fma(y, fma(-z, y, 1.0), y); };
  for (int sweep = 0; sweep < 30; ++sweep) {
    bool rotated = false;
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int q = p + 1; q < 4; ++q) {
        double al = 0, be = 0, ga = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          al += a[i][p] * a[i][p];
          be += a[i][q] * a[i][q];
          ga += a[i][p] * a[i][q];
        }
        // columns already orthogonal to working precision (|ga| <= 1e-15 |a_p||a_q|) are left alone:
        // rotating on rounding noise never converges (a handful of tracks used to burn all 30 sweeps)
        if (ga * ga > 1e-30 * (al * be) && fabs(ga) > 1e-300) {
          rotated = true;
          const double zeta = (be - al) * rcp(2.0 * ga);
          const double s1 = fma(zeta, zeta, 1.0);
          const double t = (zeta >= 0 ? 1.0 : -1.0) * rcp(fabs(zeta) + s1 * rsq(s1));
          const double c = rsq(fma(t, t, 1.0)), s = c * t;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const double ap = a[i][p], aq = a[i][q];
            a[i][p] = c * ap - s * aq;
            a[i][q] = s * ap + c * aq;
            const double vp = v[i][p], vq = v[i][q];
            v[i][p] = c * vp - s * vq;
            v[i][q] = s * vp + c * vq;
          }
        }
      }
    if (!rotated) break;
  }
  double bn = 1e300;
  x[0] = x[1] = x[2] = x[3] = 0.0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    double nn = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) nn += a[i][j] * a[i][j];
    if (nn < bn) {
      bn = nn;
#pragma unroll
      for (int i = 0; i < 4; ++i) x[i] = v[i][j];
    }
  }
}

// Solve the symmetric 3x3 system G d = b by LU with partial pivoting
// (Eigen's dynamic .inverse() at triangulation.cpp:193-194 is PartialPivLU).
__device__ inline bool xk_solve3(const double G[6] /*00 01 02 11 12 22*/, const double b[3], double d[3]) {
  double m[3][4] = {{G[0], G[1], G[2], b[0]}, {G[1], G[3], G[4], b[1]}, {G[2], G[4], G[5], b[2]}};
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    int piv = k;
    double mx = fabs(m[k][k]);
    for (int i = k + 1; i < 3; ++i)
      if (fabs(m[i][k]) > mx) { mx = fabs(m[i][k]); piv = i; }
    if (!(mx > 0.0)) return false;
    if (piv != k)
      for (int j = 0; j < 4; ++j) { double t = m[k][j]; m[k][j] = m[piv][j]; m[piv][j] = t; }
    for (int i = k + 1; i < 3; ++i) {
      const double f = m[i][k] / m[k][k];
      for (int j = k; j < 4; ++j) m[i][j] -= f * m[k][j];
    }
  }
  d[2] = m[2][3] / m[2][2];
  d[1] = (m[1][3] - m[1][2] * d[2]) / m[1][1];
  d[0] = (m[0][3] - m[0][1] * d[1] - m[0][2] * d[2]) / m[0][0];
  return true;
}

// One observation's contribution to the Gauss-Newton normal equations
// (triangulation.cpp:158-190).
__device__ __forceinline__ void xk_gn_accum(const double (&dr)[3][3], const double (&dp)[3], double ox,
                                            double oy, double alpha, double beta, double rho,
                                            double (&acc)[10]) {
  const double hx = dr[0][0] * alpha + dr[0][1] * beta + dr[0][2] + rho * dp[0];
  const double hy = dr[1][0] * alpha + dr[1][1] * beta + dr[1][2] + rho * dp[1];
  const double hz = dr[2][0] * alpha + dr[2][1] * beta + dr[2][2] + rho * dp[2];
  const double rx = ox - hx / hz, ry = oy - hy / hz;
  const double j1a = -1.0 / hz, j1c = hx / (hz * hz), j1d = hy / (hz * hz);
  double jr0[3], jr1[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const double j00 = (c < 2) ? dr[0][c] : dp[0];
    const double j01 = (c < 2) ? dr[1][c] : dp[1];
    const double j02 = (c < 2) ? dr[2][c] : dp[2];
    jr0[c] = j1a * j00 + j1c * j02;
    jr1[c] = j1a * j01 + j1d * j02;
  }
  acc[0] += jr0[0] * jr0[0] + jr1[0] * jr1[0];
  acc[1] += jr0[0] * jr0[1] + jr1[0] * jr1[1];
  acc[2] += jr0[0] * jr0[2] + jr1[0] * jr1[2];
  acc[3] += jr0[1] * jr0[1] + jr1[1] * jr1[1];
  acc[4] += jr0[1] * jr0[2] + jr1[1] * jr1[2];
  acc[5] += jr0[2] * jr0[2] + jr1[2] * jr1[2];
  acc[6] += jr0[0] * rx + jr1[0] * ry;
  acc[7] += jr0[1] * rx + jr1[1] * ry;
  acc[8] += jr0[2] * rx + jr1[2] * ry;
  acc[9] += rx * rx + ry * ry;
}

// Cholesky-based gate on S = M[3:,3:] (d x d, lower triangle) with the residual stored as row d:
// right-looking and un-normalised, S(i,j) -= S(i,k) S(j,k) / S(k,k) for i >= j > k, which touches
// only columns > k -- column k and the pivot stay valid through the step, so ONE barrier per step
// suffices; gamma = sum_k S(d,k)^2 / S(k,k).  Threads form a fixed 16 x 16 grid; each owns an
// the elements (i,j) = (ti, tj) mod 16 (no index divisions in the loop).
// Gate matrix storage: full rows of ldm doubles (windows <= 33 poses: the row sweeps of the two-sided update want them) or the
// PACKED lower triangle, element (i, j), i >= j, at i (i + 1) / 2 + j -- half the LDS, so that two workgroups still share a CU
// at windows of 34..64 poses (81 KB -> 41 KB for the matrix at L = 50).
template <bool PACKED>
__device__ __forceinline__ size_t xk_gm(int i, int j, int ldm) {
  if constexpr (PACKED) return (i >= j) ? (size_t)i * (i + 1) / 2 + j : (size_t)j * (j + 1) / 2 + i;
  else return (size_t)i * ldm + j;
}
template <int NT, bool PACKED>
__device__ __forceinline__ void xk_chol_gate(double *Mm, int ldm, int d, int tid, double *scal) {
#define XK_S(i, j) Mm[xk_gm<PACKED>(3 + (i), 3 + (j), ldm)]
  const int ti = tid >> 4, tj = tid & 15;
  double g = 0.0;
  bool bad = false;
  for (int kk = 0; kk < d; ++kk) {
    const double piv = XK_S(kk, kk);
    if (!(piv > 0.0)) { bad = true; break; }   // uniform: every thread reads the same pivot
    double rp = __builtin_amdgcn_rcp(piv);
    rp = fma(rp, fma(-piv, rp, 1.0), rp);
    rp = fma(rp, fma(-piv, rp, 1.0), rp);
    if (tid == 0) { const double y = XK_S(d, kk); g = fma(y * y, rp, g); }
    for (int i = kk + 1 + ((ti - (kk + 1)) & 15); i <= d; i += 16) {
      const double sik = XK_S(i, kk) * rp;
      for (int j = kk + 1 + ((tj - (kk + 1)) & 15); j <= i && j < d; j += 16)
        XK_S(i, j) = fma(-sik, XK_S(j, kk), XK_S(i, j));
    }
    __syncthreads();
  }
  if (tid == 0) {
    if (bad) scal[10] = 1.0;
    scal[12] = g;
  }
#undef XK_S
}

// The same gate for d <= 63 as a blocked Cholesky on ONE wave, 16 x 16 tiles in the MFMA C/D layout, all of
// them (<= 10 of S plus <= 4 of the residual, which rides along as a right-hand-side column) in this wave's
// registers: per block step the diagonal tile is factored and inverted by the DPP row-broadcast pivot chain
// (xk_chol16_bcast, ~3.1 k clocks for 16 pivots; the row-per-lane version with the column broadcast through LDS that
// this replaces took ~12 k), row j becomes
// X_jk = L_jj^-1 S_jk on the matrix cores, and the trailing tiles take S_ik -= X_ji^T X_jk straight from
// registers (a C/D-layout register is both operands of X^T X).  gamma = |L^-1 r|^2 = the squares of the
// residual column of X.  work: 256 + 16*17 doubles of LDS (diagonal tile row-major, L_jj^-1).
template <bool PACKED>
__device__ __forceinline__ void xk_chol_gate_blocked(const double *Mm, int ldm, int d, int lane, double *scal, double *work) {
#define XK_S(i, j) Mm[xk_gm<PACKED>(3 + (i), 3 + (j), ldm)]      // lower triangle valid; row d = the residual
  constexpr int NB = 4;
  const int nb = (d + 15) >> 4, li = lane & 15, lk = lane >> 4;
  double *dbuf = work, *Ls = work + 256;
  xk_d4 T[NB][NB], R[NB];                                     // T[i][k], i <= k
#pragma unroll
  for (int i = 0; i < NB; ++i) {
#pragma unroll
    for (int k = i; k < NB; ++k) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int r = 16 * i + lk + 4 * q, c = 16 * k + li;  // element (r, c), r <= c off the diagonal tile
        double v = (r == c) ? 1.0 : 0.0;                      // identity padding past d
        if (i < nb && k < nb && r < d && c < d) v = (r >= c) ? XK_S(r, c) : XK_S(c, r);
        T[i][k][q] = v;
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int r = 16 * i + lk + 4 * q;
      R[i][q] = (li == 0 && r < d) ? XK_S(d, r) : 0.0;        // residual in column 0 of its tile
    }
  }
  bool bad = false;
  double g = 0.0;
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    if (j < nb) {                                             // uniform
#pragma unroll
      for (int q = 0; q < 4; ++q) dbuf[64 * q + lane] = T[j][j][q];   // C/D layout == row-major 16 x 16
      __builtin_amdgcn_s_waitcnt(0xc07f);
      __builtin_amdgcn_wave_barrier();
      if (xk_chol16_bcast(dbuf, Ls, lane)) bad = true;
      __builtin_amdgcn_s_waitcnt(0xc07f);
      __builtin_amdgcn_wave_barrier();
      double lv[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) lv[q] = Ls[li * 17 + 4 * q + lk];
      // row j
#pragma unroll
      for (int k = j + 1; k < NB; ++k) {
        if (k < nb) {
          xk_d4 x = {0, 0, 0, 0};
#pragma unroll
          for (int q = 0; q < 4; ++q) x = __builtin_amdgcn_mfma_f64_16x16x4f64(lv[q], T[j][k][q], x, 0, 0, 0);
          T[j][k] = x;
        }
      }
      {
        xk_d4 x = {0, 0, 0, 0};
#pragma unroll
        for (int q = 0; q < 4; ++q) x = __builtin_amdgcn_mfma_f64_16x16x4f64(lv[q], R[j][q], x, 0, 0, 0);
        R[j] = x;
#pragma unroll
        for (int q = 0; q < 4; ++q) g = fma(x[q], x[q], g);   // only column 0 (lanes li == 0) is non-zero
      }
      // trailing tiles
#pragma unroll
      for (int i = j + 1; i < NB; ++i) {
        if (i < nb) {
#pragma unroll
          for (int k = i; k < NB; ++k) {
            if (k < nb) {
#pragma unroll
              for (int q = 0; q < 4; ++q) T[i][k] = __builtin_amdgcn_mfma_f64_16x16x4f64(-T[j][i][q], T[j][k][q], T[i][k], 0, 0, 0);
            }
          }
#pragma unroll
          for (int q = 0; q < 4; ++q) R[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(-T[j][i][q], R[j][q], R[i], 0, 0, 0);
        }
      }
    }
  }
  g = xk_wave_sum(g);
  if (lane == 0) {
    scal[12] = g;
    if (bad) scal[10] = 1.0;
  }
#undef XK_S
}

// The blocked gate for 64 <= d <= 125 (windows of 34..64 poses) on the FOUR waves of the workgroup: 16 x 16 tiles as above,
// tile column k -- tiles (i, k), i <= k -- in the registers of wave k % 4; the residual rides along as tile column nb.
// Per block step j: the owner of column j factors and inverts the diagonal tile (xk_chol16_bcast), barrier, every wave turns
// its tiles of row j into X_jk = L_jj^-1 S_jk and leaves them in LDS (C/D layout, 2 KB each -- in the gate matrix's own
// storage, which is dead once the tiles are in registers), barrier, every wave updates its columns, S_ik -= X_ji^T X_jk, with
// X_ji read back in the layout that makes a C/D register the A operand of X^T X.  ~8 block steps of (16-pivot chain + two
// barriers + <= 9 tile updates per wave) instead of d barrier-separated rank-1 steps: 119 -> 25 us at d = 97.
// NC = tile columns per wave: 2 covers nb <= 7 (d <= 111, residual column included), 3 the rest
template <bool PACKED, int NC>
__device__ __forceinline__ void xk_chol_gate_blocked4(double *Mm, int ldm, int d, int tid, double *scal, double *work) {
#define XK_S(i, j) Mm[xk_gm<PACKED>(3 + (i), 3 + (j), ldm)]      // lower triangle valid; row d = the residual
  constexpr int NB = 8;
  const int nb = (d + 15) >> 4, lane = tid & 63, w = tid >> 6, li = lane & 15, lk = lane >> 4;
  double *dbuf = work, *Ls = work + 256;
  xk_d4 T[NC][NB];                                             // T[c][i] = tile (i, k), k = w + 4 c
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const int k = w + 4 * c;
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      if (c == 0 && i >= 4) continue;                         // (column k <= 3 has no tile below row 3)
      xk_d4 t = {0, 0, 0, 0};
      if (i <= k && i < nb && k <= nb) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int r = 16 * i + lk + 4 * q, cc = 16 * k + li;
          double v;
          if (k < nb) {
            v = (r == cc) ? 1.0 : 0.0;                        // identity padding past d
            if (r < d && cc < d) v = (r >= cc) ? XK_S(r, cc) : XK_S(cc, r);
          } else {
            v = (li == 0 && r < d) ? XK_S(d, r) : 0.0;        // residual in column 0 of its tiles
          }
          t[q] = v;
        }
      }
      T[c][i] = t;
    }
  }
  __syncthreads();                                            // every tile is in registers: Mm becomes the exchange area
  double *Xs = Mm + (((size_t)Mm >> 3) & 1);                  // [NB][256], 16-byte aligned
  bool bad = false;
  double g = 0.0;
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    if (j < nb) {                                             // uniform
      const int wj = j & 3, cj = j >> 2;
      if (w == wj) {
#pragma unroll
        for (int q = 0; q < 4; ++q) dbuf[64 * q + lane] = T[cj][j][q];   // C/D layout == row-major 16 x 16
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
        if (xk_chol16_bcast(dbuf, Ls, lane)) bad = true;
      }
      __syncthreads();
      double lv[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) lv[q] = Ls[li * 17 + 4 * q + lk];
      // row j of this wave's columns
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        if (c == 0 && j >= 4) continue;
        const int k = w + 4 * c;
        if (k > j && k <= nb) {
          xk_d4 x = {0, 0, 0, 0};
#pragma unroll
          for (int q = 0; q < 4; ++q) x = __builtin_amdgcn_mfma_f64_16x16x4f64(lv[q], T[c][j][q], x, 0, 0, 0);
          T[c][j] = x;
          if (k < nb) {
#pragma unroll
            for (int q = 0; q < 4; ++q) Xs[256 * k + 64 * q + lane] = x[q];
          } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) g = fma(x[q], x[q], g);   // only column 0 (lanes li == 0) is non-zero
          }
        }
      }
      __syncthreads();
      // trailing tiles of this wave's columns
#pragma unroll
      for (int i = j + 1; i < NB; ++i) {
        if (i < nb && i <= w + 4 * (NC - 1)) {
          double xa[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) xa[q] = -Xs[256 * i + 64 * q + lane];
#pragma unroll
          for (int c = 0; c < NC; ++c) {
            if (c == 0 && (i >= 4 || j >= 4)) continue;
            const int k = w + 4 * c;
            if (i <= k && k <= nb) {
#pragma unroll
              for (int q = 0; q < 4; ++q) T[c][i] = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[q], T[c][j][q], T[c][i], 0, 0, 0);
            }
          }
        }
      }
    }
  }
  if (bad && lane == 0) scal[10] = 1.0;
  if (w == (nb & 3)) {                                        // the owner of the residual column
    g = xk_wave_sum(g);
    if (lane == 0) scal[12] = g;
  }
#undef XK_S
}

// LDS size in bytes for n_poses window poses.
static inline bool xk_feature_packed(int n_poses) { return n_poses > 33; }
static inline size_t xk_feature_lds_bytes(int n_poses, bool packed = false) {
  const int L = n_poses, m2 = 2 * L, ldm = m2 + 1;
  const size_t gate = packed ? (size_t)(m2 + 1) * (m2 + 2) / 2 + 1 : (size_t)(m2 + 1) * ldm;
  return sizeof(double) * (size_t)(9 * L + 3 * L + 6 * L + 6 * L + m2 + 3 * m2 + gate + 32 + 6 * (size_t)m2 + 64 + 2 + 256 + 272);
}

template <bool PACKED>
__device__ __forceinline__ void xk_msckf_feature_body(XkFeatArgs a_in) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  XkFeatArgs a = a_in;
  if (a_in.batch) {
    const XkFeatBatch bb = a_in.batch[blockIdx.x];
    a.q = bb.q; a.p = bb.p; a.obs = bb.obs; a.P = bb.P;
    a.n_poses = bb.n_poses; a.n_poses_max = bb.n_poses_max; a.n = bb.n; a.na = bb.n - XK_CORE; a.up_out = bb.up_out;
  }
  XK_WG_BEGIN();
  const int tid = threadIdx.x, k = blockIdx.x;
  const int np = a.n_poses, Lmax = np;
  const int ldm = 2 * Lmax + 1;
  double *rot = sm;              // [np][9] camera->world, row-major
  double *pos = rot + 9 * np;    // [np][3]
  double *Jp = pos + 3 * np;     // [L][2][3] post-OC position blocks
  double *Ja = Jp + 6 * Lmax;    // [L][2][3] post-OC attitude blocks
  double *res = Ja + 6 * Lmax;   // [2L]
  double *V = res + 2 * Lmax;    // [3][2L] Hf, then the three reflectors
  double *Mm = V + 6 * Lmax;     // [(2L+1)][ldm] gate matrix (+ appended residual row)
  double *scal = Mm + (PACKED ? (((size_t)(2 * Lmax + 1) * (2 * Lmax + 2) / 2 + 1) & ~(size_t)1) : (size_t)(2 * Lmax + 1) * ldm);  // 32 scalars (16..24: R factor of Hf)
  // scal: 0..2 tau, 3 g01, 4 g02, 5 g12, 6..8 gpf, 9 valid, 10 bad, 11 inlier, 12 gamma

  const int off = a_in.batch ? 0 : a.trk_off[k], L = a_in.batch ? a_in.batch[k].L : a.trk_off[k + 1] - off;
  const int m2 = 2 * L, d = m2 - 3, p0 = np - L;
  const double chi_gate = a.chi95[d];   // fetched now: its HBM latency would otherwise sit between the gate and the tile write

  for (int i = tid; i < np; i += XK_FEAT_THREADS) {
    xk_quat_to_rot(a.q + 4 * i, rot + 9 * i);
    pos[3 * i] = a.p[3 * i];
    pos[3 * i + 1] = a.p[3 * i + 1];
    pos[3 * i + 2] = a.p[3 * i + 2];
  }
  if (tid == 0) { scal[9] = 1.0; scal[10] = 0.0; }
  __syncthreads();

  XK_STAMP(0);
  // ---- triangulation: DLT + Gauss-Newton, wave 0 (triangulation.cpp:102-206)
  if (a.gpf_in) {
    if (tid < 3) scal[6 + tid] = a.gpf_in[tid];
    if (tid == 0) a.gn_iters[k] = 0;
  } else if (tid < 64) {
    const int lane = tid;
    const double *Ra = rot + 9 * (p0 + L - 1), *pa = pos + 3 * (p0 + L - 1);
    const double *R1 = rot + 9 * p0, *p1 = pos + 3 * p0;
    double alpha, beta, rho;
    {
      // projection matrices [R^T | -R^T p] of first and last pose (:208-216)
      double A4[4][4], X[4];
      double P1[3][4], P2[3][4];
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        double t1 = 0, t2 = 0;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          P1[r][c] = R1[3 * c + r];
          P2[r][c] = Ra[3 * c + r];
          t1 -= R1[3 * c + r] * p1[c];
          t2 -= Ra[3 * c + r] * pa[c];
        }
        P1[r][3] = t1;
        P2[r][3] = t2;
      }
      const double o1x = a.obs[2 * (size_t)off], o1y = a.obs[2 * (size_t)off + 1];
      const double o2x = a.obs[2 * (size_t)(off + L - 1)], o2y = a.obs[2 * (size_t)(off + L - 1) + 1];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        A4[0][c] = o1x * P1[2][c] - P1[0][c];
        A4[1][c] = o1y * P1[2][c] - P1[1][c];
        A4[2][c] = o2x * P2[2][c] - P2[0][c];
        A4[3][c] = o2y * P2[2][c] - P2[1][c];
      }
      xk_null4(A4, X);
      const double wx = X[0] / X[3], wy = X[1] / X[3], wz = X[2] / X[3];
      double pc[3];
#pragma unroll
      for (int r = 0; r < 3; ++r) pc[r] = P2[r][0] * wx + P2[r][1] * wy + P2[r][2] * wz + P2[r][3];
      alpha = pc[0] / pc[2];
      beta = pc[1] / pc[2];
      rho = 1.0 / pc[2];
    }
    // per-lane observation geometry (lanes >= L contribute zeros)
    double drot[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}}, dpos[3] = {0, 0, 0}, ox = 0, oy = 0;
    // tracks longer than 64 are handled by a second observation per lane
    double drot2[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}}, dpos2[3] = {0, 0, 0}, ox2 = 0, oy2 = 0;
    const bool act = lane < L, act2 = lane + 64 < L;
    if (act) {
      const double *Ri = rot + 9 * (p0 + lane), *pi = pos + 3 * (p0 + lane);
#pragma unroll
      for (int r = 0; r < 3; ++r) {
#pragma unroll
        for (int c = 0; c < 3; ++c)
          drot[r][c] = Ri[r] * Ra[c] + Ri[3 + r] * Ra[3 + c] + Ri[6 + r] * Ra[6 + c];  // rot_i rot_a^T
        dpos[r] = (Ri[r] * pa[0] + Ri[3 + r] * pa[1] + Ri[6 + r] * pa[2]) -
                  (Ri[r] * pi[0] + Ri[3 + r] * pi[1] + Ri[6 + r] * pi[2]);
      }
      ox = a.obs[2 * (size_t)(off + lane)];
      oy = a.obs[2 * (size_t)(off + lane) + 1];
    }
    if (act2) {
      const double *Ri = rot + 9 * (p0 + lane + 64), *pi = pos + 3 * (p0 + lane + 64);
#pragma unroll
      for (int r = 0; r < 3; ++r) {
#pragma unroll
        for (int c = 0; c < 3; ++c)
          drot2[r][c] = Ri[r] * Ra[c] + Ri[3 + r] * Ra[3 + c] + Ri[6 + r] * Ra[6 + c];
        dpos2[r] = (Ri[r] * pa[0] + Ri[3 + r] * pa[1] + Ri[6 + r] * pa[2]) -
                   (Ri[r] * pi[0] + Ri[3 + r] * pi[1] + Ri[6 + r] * pi[2]);
      }
      ox2 = a.obs[2 * (size_t)(off + lane + 64)];
      oy2 = a.obs[2 * (size_t)(off + lane + 64) + 1];
    }
    double r_norm_last = 1000.0, r_norm = 100.0;
    int iter = 0;
    bool ok = true;
    while (r_norm_last - r_norm > 1e-5) {  // term, vio_updater.cpp:290 / msckf_update.h:93-96
      iter++;
      if (iter > 10) break;
      double acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};  // JtJ(00 01 02 11 12 22) Jtr(3) rr
      if (act) xk_gn_accum(drot, dpos, ox, oy, alpha, beta, rho, acc);
      if (act2) xk_gn_accum(drot2, dpos2, ox2, oy2, alpha, beta, rho, acc);
#pragma unroll
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
      // getGlobalFeaturePosition, msckf_update.cpp:283-304
      for (int r = 0; r < 3; ++r)
        scal[6 + r] = (1.0 / rho) * (Ra[3 * r] * alpha + Ra[3 * r + 1] * beta + Ra[3 * r + 2]) + pa[r];
      if (!ok) scal[6] = nan("");
      a.gn_iters[k] = iter;
    }
  }
  __syncthreads();
  const double gx = scal[6], gy = scal[7], gz = scal[8];
  if (tid < 3) a.gpf[3 * (size_t)k + tid] = scal[6 + tid];

  XK_STAMP(1);
  // ---- per-observation Jacobians (msckf_update.cpp:328-417)
  for (int i = tid; i < L; i += XK_FEAT_THREADS) {
    const double *R = rot + 9 * (p0 + i), *pp = pos + 3 * (p0 + i);
    const double dx = gx - pp[0], dy = gy - pp[1], dz = gz - pp[2];
    const double cx = R[0] * dx + R[3] * dy + R[6] * dz;
    const double cy = R[1] * dx + R[4] * dy + R[7] * dz;
    const double cz = R[2] * dx + R[5] * dy + R[8] * dz;
    if (!(cx == cx && cy == cy && cz == cz)) scal[9] = 0.0;  // :349-357
    const double ox = a.obs[2 * (size_t)(off + i)], oy = a.obs[2 * (size_t)(off + i) + 1];
    res[2 * i] = ox - cx / cz;
    res[2 * i + 1] = oy - cy / cz;
    const double Ji[2][3] = {{1.0 / cz, 0.0, -cx / (cz * cz)}, {0.0, 1.0 / cz, -cy / (cz * cz)}};
    double jp[2][3], ja[2][3];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
#pragma unroll
      for (int c = 0; c < 3; ++c)  // -Ji R^T
        jp[r][c] = -(Ji[r][0] * R[3 * c] + Ji[r][1] * R[3 * c + 1] + Ji[r][2] * R[3 * c + 2]);
      // Ji * Skew(c)
      ja[r][0] = Ji[r][1] * cz - Ji[r][2] * cy;
      ja[r][1] = -Ji[r][0] * cz + Ji[r][2] * cx;
      ja[r][2] = Ji[r][0] * cy - Ji[r][1] * cx;
    }
    // observability constraint, g = (0,0,-9.81) (:393-406)
    {
      const double g = -9.81;
      double u[3] = {R[2] * g, R[5] * g, R[8] * g};  // R g
      double uu = u[0] * u[0] + u[1] * u[1] + u[2] * u[2];
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const double t = (jp[r][0] * u[0] + jp[r][1] * u[1] + jp[r][2] * u[2]) * (1.0 / uu);
        jp[r][0] -= t * u[0];
        jp[r][1] -= t * u[1];
        jp[r][2] -= t * u[2];
      }
      // Skew(G_p_f - G_p_C) g
      u[0] = dy * g;
      u[1] = -dx * g;
      u[2] = 0.0;
      uu = u[0] * u[0] + u[1] * u[1];
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const double t = (ja[r][0] * u[0] + ja[r][1] * u[1]) * (1.0 / uu);
        ja[r][0] -= t * u[0];
        ja[r][1] -= t * u[1];
      }
    }
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        Jp[6 * i + 3 * r + c] = jp[r][c];
        Ja[6 * i + 3 * r + c] = ja[r][c];
        V[c * m2 + 2 * i + r] = -jp[r][c];  // Hf block, :409
      }
  }
  __syncthreads();

  XK_STAMP(2);
  // ---- wave 0: Householder QR of Hf (2L x 3) -> three reflectors (:423), then the residual r' = Q^T res;
  //      waves 1-3 meanwhile: the gate matrix (independent of the reflectors)
  if (tid < 64) {
    const int lane = tid;
    for (int kk = 0; kk < 3; ++kk) {
      double *col = V + kk * m2;
      double tail = 0.0;
      for (int r = lane; r < m2; r += 64)
        if (r > kk) tail += col[r] * col[r];
      tail = xk_wave_sum(tail);
      const double c0 = col[kk];
      double tau, bet, sc;
      if (tail <= 2.2250738585072014e-308) { tau = 0.0; bet = c0; sc = 0.0; }
      else {
        bet = sqrt(c0 * c0 + tail);
        if (c0 >= 0) bet = -bet;
        tau = (bet - c0) / bet;
        sc = 1.0 / (c0 - bet);
      }
      for (int r = lane; r < m2; r += 64)
        if (r > kk) col[r] *= sc;
      if (lane == 0) scal[16 + 4 * kk] = bet;  // R(kk,kk) of Hf = (A_up^T Hf)(kk,kk)
      for (int c2 = kk + 1; c2 < 3; ++c2) {
        double *cc = V + c2 * m2;
        double w = 0.0;
        for (int r = lane; r < m2; r += 64)
          if (r > kk) w += col[r] * cc[r];
        w = tau * (xk_wave_sum(w) + cc[kk]);
        for (int r = lane; r < m2; r += 64)
          if (r > kk) cc[r] -= w * col[r];
        if (lane == 0) { cc[kk] -= w; scal[16 + kk + 3 * c2] = cc[kk]; }  // R(kk,c2), column-major 3x3 at scal[16..24]
      }
      if (lane == 0) scal[kk] = tau;
    }
    // make the reflectors explicit: v_k[k] = 1, v_k[r<k] = 0
    if (lane == 0) {
      V[0] = 1.0;
      V[m2] = 0.0; V[m2 + 1] = 1.0;
      V[2 * m2] = 0.0; V[2 * m2 + 1] = 0.0; V[2 * m2 + 2] = 1.0;
    }
    // Gram terms for the WY application
    double g01 = 0, g02 = 0, g12 = 0;
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    for (int r = lane; r < m2; r += 64) {
      const double v0 = (r == 0) ? 1.0 : V[r];
      const double v1 = (r < 1) ? 0.0 : (r == 1 ? 1.0 : V[m2 + r]);
      const double v2 = (r < 2) ? 0.0 : (r == 2 ? 1.0 : V[2 * m2 + r]);
      g01 += v0 * v1;
      g02 += v0 * v2;
      g12 += v1 * v2;
    }
    g01 = xk_wave_sum(g01);
    g02 = xk_wave_sum(g02);
    g12 = xk_wave_sum(g12);
    if (lane == 0) { scal[3] = g01; scal[4] = g02; scal[5] = g12; }
    // residual r' = Q^T res
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    for (int kk = 0; kk < 3; ++kk) {
      const double tk = scal[kk];
      double w = 0.0;
      for (int r = lane; r < m2; r += 64)
        if (r >= kk) w += V[kk * m2 + r] * res[r];
      w = tk * xk_wave_sum(w);
      for (int r = lane; r < m2; r += 64)
        if (r >= kk) res[r] -= w * V[kk * m2 + r];
      __builtin_amdgcn_s_waitcnt(0);
      __builtin_amdgcn_wave_barrier();
    }
  } else {
    // ---- gate matrix M = J P J^T + sigma^2 I from 6x6 blocks of P (:452-457)
    {
      const int n = a.n;
      const double *P = a.P;
      // Pairs (ia <= ib) are dealt so that the lanes of a group share ib and walk ia: P is column-major, so a
      // load then touches 3-double runs 24 bytes apart in ONE column (a dozen cache lines per wave) instead of 64
      // scattered lines.  Column ib = c carries c + 1 pairs and column L-1-c carries L - c: together L + 1 slots,
      // one group of G lanes per such couple of columns.
      const int G = (L + 1 <= 32) ? 32 : (L + 1 <= 64) ? 64 : 128, ncouple = (L + 1) / 2;
      for (int idx = tid - 64; idx < ncouple * G; idx += XK_FEAT_THREADS - 64) {
        const int cpl = idx / G, u = idx - cpl * G, ib2 = L - 1 - cpl;
        if (u > L || (u > cpl && ib2 == cpl)) continue;          // past the slots / the middle column of an odd L
        const int ia = (u <= cpl) ? u : u - cpl - 1, ib = (u <= cpl) ? cpl : ib2;
        const int cpa = XK_CORE + 3 * (p0 + ia), caa = cpa + 3 * a.n_poses_max;
        const int cpb = XK_CORE + 3 * (p0 + ib), cab = cpb + 3 * a.n_poses_max;
        const double *jpa = Jp + 6 * ia, *jaa = Ja + 6 * ia, *jpb = Jp + 6 * ib, *jab = Ja + 6 * ib;
        double t1[2][3], t2[2][3];  // rows of J_a times P[.., pos_b cols] / P[.., att_b cols]
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const double *pc1 = P + (size_t)(cpb + c) * n, *pc2 = P + (size_t)(cab + c) * n;
          const double p00 = pc1[cpa], p01 = pc1[cpa + 1], p02 = pc1[cpa + 2];
          const double p10 = pc1[caa], p11 = pc1[caa + 1], p12 = pc1[caa + 2];
          const double q00 = pc2[cpa], q01 = pc2[cpa + 1], q02 = pc2[cpa + 2];
          const double q10 = pc2[caa], q11 = pc2[caa + 1], q12 = pc2[caa + 2];
#pragma unroll
          for (int r = 0; r < 2; ++r) {
            t1[r][c] = jpa[3 * r] * p00 + jpa[3 * r + 1] * p01 + jpa[3 * r + 2] * p02 + jaa[3 * r] * p10 +
                       jaa[3 * r + 1] * p11 + jaa[3 * r + 2] * p12;
            t2[r][c] = jpa[3 * r] * q00 + jpa[3 * r + 1] * q01 + jpa[3 * r + 2] * q02 + jaa[3 * r] * q10 +
                       jaa[3 * r + 1] * q11 + jaa[3 * r + 2] * q12;
          }
        }
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
          for (int s = 0; s < 2; ++s) {
            double v = t1[r][0] * jpb[3 * s] + t1[r][1] * jpb[3 * s + 1] + t1[r][2] * jpb[3 * s + 2] +
                       t2[r][0] * jab[3 * s] + t2[r][1] * jab[3 * s + 1] + t2[r][2] * jab[3 * s + 2];
            if (ia == ib && r == s) v += a.var_img;
            if (ia == ib && r > s) continue;  // keep the diagonal block symmetric: use upper entry
            Mm[xk_gm<PACKED>(2 * ia + r, 2 * ib + s, ldm)] = v;
            if constexpr (!PACKED) Mm[(size_t)(2 * ib + s) * ldm + 2 * ia + r] = v;
          }
      }
    }
  }
  __syncthreads();
  const double tau0 = scal[0], tau1 = scal[1], tau2 = scal[2];
  const double g01 = scal[3], g02 = scal[4], g12 = scal[5];

  XK_STAMP(3);
  XK_STAMP(4);
  // ---- M <- Q^T M Q with Q = H0 H1 H2 = I - V T V^T (compact WY):
  //        Q^T M Q = M - V Z^T - Z V^T,   Z = Y T - V B / 2,  Y = M V,  B = T^T (V^T Y) T
  // one matvec sweep + one rank-6 sweep over the lower triangle instead of six reflector sweeps.
  // Y and Z (3 x 2L each) and the Cholesky broadcast column live behind the scalar block.
  double *Yv = scal + 32;          // [3][m2]
  double *Zv = Yv + 3 * 2 * Lmax;  // [3][m2]
  {
    // Y = M V : 4 lanes per row, each sweeps a quarter of the columns
    const int row = tid >> 2, qd = tid & 3;
    double y0 = 0.0, y1 = 0.0, y2 = 0.0;
    if (row < m2) {
      for (int j = qd; j < m2; j += 4) {
        const double mv = Mm[xk_gm<PACKED>(row, j, ldm)];
        y0 = fma(mv, V[j], y0);
        y1 = fma(mv, V[m2 + j], y1);
        y2 = fma(mv, V[2 * m2 + j], y2);
      }
    }
    y0 += __shfl_xor(y0, 1, 64); y0 += __shfl_xor(y0, 2, 64);
    y1 += __shfl_xor(y1, 1, 64); y1 += __shfl_xor(y1, 2, 64);
    y2 += __shfl_xor(y2, 1, 64); y2 += __shfl_xor(y2, 2, 64);
    if (row < m2 && qd == 0) { Yv[row] = y0; Yv[m2 + row] = y1; Yv[2 * m2 + row] = y2; }
    // rows 64.. (tracks longer than 32): second sweep
    for (int row2 = row + 64; row2 < m2; row2 += 64) {
      double z0 = 0.0, z1 = 0.0, z2 = 0.0;
      for (int j = qd; j < m2; j += 4) {
        const double mv = Mm[xk_gm<PACKED>(row2, j, ldm)];
        z0 = fma(mv, V[j], z0);
        z1 = fma(mv, V[m2 + j], z1);
        z2 = fma(mv, V[2 * m2 + j], z2);
      }
      z0 += __shfl_xor(z0, 1, 64); z0 += __shfl_xor(z0, 2, 64);
      z1 += __shfl_xor(z1, 1, 64); z1 += __shfl_xor(z1, 2, 64);
      z2 += __shfl_xor(z2, 1, 64); z2 += __shfl_xor(z2, 2, 64);
      if (qd == 0) { Yv[row2] = z0; Yv[m2 + row2] = z1; Yv[2 * m2 + row2] = z2; }
    }
  }
  __syncthreads();
  if (tid < 64) {
    // W = V^T Y (3x3), T, B = T^T W T, then Z = Y T - V B / 2
    double wacc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int r = tid; r < m2; r += 64)
#pragma unroll
      for (int x = 0; x < 3; ++x)
#pragma unroll
        for (int y = 0; y < 3; ++y) wacc[3 * x + y] = fma(V[x * m2 + r], Yv[y * m2 + r], wacc[3 * x + y]);
#pragma unroll
    for (int c = 0; c < 9; ++c) wacc[c] = xk_wave_sum(wacc[c]);
    const double T00 = tau0, T11 = tau1, T22 = tau2;
    const double T01 = -tau1 * T00 * g01;
    const double T02 = -tau2 * (T00 * g02 + T01 * g12);
    const double T12 = -tau2 * (T11 * g12);
    const double Tm[3][3] = {{T00, T01, T02}, {0.0, T11, T12}, {0.0, 0.0, T22}};
    double WT[3][3], Bm[3][3];
#pragma unroll
    for (int x = 0; x < 3; ++x)
#pragma unroll
      for (int y = 0; y < 3; ++y) WT[x][y] = wacc[3 * x] * Tm[0][y] + wacc[3 * x + 1] * Tm[1][y] + wacc[3 * x + 2] * Tm[2][y];
#pragma unroll
    for (int x = 0; x < 3; ++x)
#pragma unroll
      for (int y = 0; y < 3; ++y) Bm[x][y] = Tm[0][x] * WT[0][y] + Tm[1][x] * WT[1][y] + Tm[2][x] * WT[2][y];
    for (int r = tid; r < m2; r += 64) {
      const double yv0 = Yv[r], yv1 = Yv[m2 + r], yv2 = Yv[2 * m2 + r];
      const double v0 = V[r], v1 = V[m2 + r], v2 = V[2 * m2 + r];
#pragma unroll
      for (int y = 0; y < 3; ++y)
        Zv[y * m2 + r] = (yv0 * Tm[0][y] + yv1 * Tm[1][y] + yv2 * Tm[2][y]) - 0.5 * (v0 * Bm[0][y] + v1 * Bm[1][y] + v2 * Bm[2][y]);
    }
  }
  __syncthreads();
  // rank-6 update of the lower triangle (rows >= 3 are all the Cholesky reads)
  // (four threads per row, each every fourth column up to the diagonal: no index division, the row's six factors read once)
  for (int i = 3 + (tid >> 2); i < m2; i += XK_FEAT_THREADS / 4) {
    const double vi0 = V[i], vi1 = V[m2 + i], vi2 = V[2 * m2 + i], zi0 = Zv[i], zi1 = Zv[m2 + i], zi2 = Zv[2 * m2 + i];
    for (int j = tid & 3; j <= i; j += 4)
      Mm[xk_gm<PACKED>(i, j, ldm)] -= vi0 * Zv[j] + zi0 * V[j] + vi1 * Zv[m2 + j] + zi1 * V[m2 + j] + vi2 * Zv[2 * m2 + j] + zi2 * V[2 * m2 + j];
  }
  __syncthreads();
  XK_STAMP(5);
  // ---- tile write: rows 3.. of Q^T [J | res] over the active columns (:431-432,468-479), by the threads
  //      t0, t0 + 1, .. of a group of `nthr`
  auto tile_write = [&](int t0, int nthr) {
    double *tile = a.A + (size_t)k * a.DB * a.C1P;
    const int N3 = 3 * a.n_poses_max;
    for (int c = t0; c < a.C1P; c += nthr) {
      if (c == a.na) {
        for (int r = 3; r < m2; ++r) tile[(size_t)(r - 3) * a.C1P + c] = res[r];
        continue;
      }
      int i = -1, comp = 0;
      const double *blk = nullptr;
      if (c < N3) { i = c / 3 - p0; comp = c % 3; blk = Jp; }
      else if (c < 2 * N3) { i = (c - N3) / 3 - p0; comp = (c - N3) % 3; blk = Ja; }
      if (c >= a.na || i < 0 || i >= L) {
        for (int r = 3; r < m2; ++r) tile[(size_t)(r - 3) * a.C1P + c] = 0.0;
        continue;
      }
      const double x0 = blk[6 * i + comp], x1 = blk[6 * i + 3 + comp];
      const int r0 = 2 * i;
      const double a0 = V[r0] * x0 + V[r0 + 1] * x1;
      const double a1 = V[m2 + r0] * x0 + V[m2 + r0 + 1] * x1;
      const double a2 = V[2 * m2 + r0] * x0 + V[2 * m2 + r0 + 1] * x1;
      const double w0 = tau0 * a0;
      const double w1 = tau1 * (a1 - w0 * g01);
      const double w2 = tau2 * (a2 - w0 * g02 - w1 * g12);
      for (int r = 3; r < m2; ++r) tile[(size_t)(r - 3) * a.C1P + c] = xk_h0_entry(w0, w1, w2, x0, x1, r0, V[r], V[m2 + r], V[2 * m2 + r], 0.0, r);
    }
  };
  // ---- or the factor record of the track (a.Hc): the same numbers, not multiplied out
  auto record_write = [&](int t0, int nthr) {
    double *rec = a.Hc + (size_t)k * a.hs;
    for (int r = t0; r < m2; r += nthr) {
      xk_d2 p0v = {V[r], V[m2 + r]}, p1v = {V[2 * m2 + r], res[r]};
      reinterpret_cast<xk_d2 *>(rec + 4 * r)[0] = p0v;
      reinterpret_cast<xk_d2 *>(rec + 4 * r)[1] = p1v;
    }
    const int N3 = 3 * a.n_poses_max;
    for (int c = t0; c < a.C1P; c += nthr) {
      double w0 = 0.0, w1 = 0.0, w2 = 0.0, x0 = 0.0, x1 = 0.0;
      int r0 = -2;
      if (c == a.na) r0 = -1;
      else {
        int i = -1, comp = 0;
        const double *blk = nullptr;
        if (c < N3) { i = c / 3 - p0; comp = c % 3; blk = Jp; }
        else if (c < 2 * N3) { i = (c - N3) / 3 - p0; comp = (c - N3) % 3; blk = Ja; }
        if (c < a.na && i >= 0 && i < L) {
          x0 = blk[6 * i + comp]; x1 = blk[6 * i + 3 + comp];
          r0 = 2 * i;
          const double a0 = V[r0] * x0 + V[r0 + 1] * x1;
          const double a1 = V[m2 + r0] * x0 + V[m2 + r0 + 1] * x1;
          const double a2 = V[2 * m2 + r0] * x0 + V[2 * m2 + r0 + 1] * x1;
          w0 = tau0 * a0;
          w1 = tau1 * (a1 - w0 * g01);
          w2 = tau2 * (a2 - w0 * g02 - w1 * g12);
        }
      }
      xk_d2 *wc = reinterpret_cast<xk_d2 *>(rec + a.hcvr + XK_HC_WC * c);
      xk_d2 q0 = {w0, w1}, q1 = {w2, x0}, q2 = {x1, (double)r0};
      wc[0] = q0; wc[1] = q1; wc[2] = q2;
    }
  };
  // ---- Cholesky of S = M[3:,3:] (d x d) with the residual as an extra row d (:457-458):
  //      y = L^-1 r0 falls out of the factorisation, gamma = |y|^2
  // Right-looking, un-normalised: S(i,j) -= S(i,k) S(j,k) / S(k,k) for i >= j > k touches only
  // columns > k, so column k and the pivot stay valid through the step and ONE barrier per step
  // suffices; gamma accumulates S(d,k)^2 / S(k,k).  Threads form a fixed 16 x 16 grid over the
  // matrix (no index arithmetic in the loop).
  for (int j = tid; j < d; j += XK_FEAT_THREADS) Mm[xk_gm<PACKED>(m2, 3 + j, ldm)] = res[3 + j];
  __syncthreads();
  if (d < 64) {
    // (the work area is read two doubles at a time: 16-byte aligned)
    double *work = scal + 32 + 12 * Lmax + 64;
    work += ((size_t)work >> 3) & 1;
    if (tid < 64) xk_chol_gate_blocked<PACKED>(Mm, ldm, d, tid, scal, work);
    else if (a.Hc) record_write(tid - 64, XK_FEAT_THREADS - 64);
    else if (a.A) tile_write(tid - 64, XK_FEAT_THREADS - 64);   // the other three waves write the tile meanwhile: a
                                                                 // rejected track's tile is masked by tile_rows = 0
  } else if (PACKED) {
    double *work = scal + 32 + 12 * Lmax + 64;
    work += ((size_t)work >> 3) & 1;
    if (d <= 111) xk_chol_gate_blocked4<PACKED, 2>(Mm, ldm, d, tid, scal, work);
    else xk_chol_gate_blocked4<PACKED, 3>(Mm, ldm, d, tid, scal, work);   // (34 spilled VGPRs, in this branch only)
  } else {
    xk_chol_gate<8, PACKED>(Mm, ldm, d, tid, scal);
  }
  __syncthreads();
  if (tid == 0) {
    const double g = scal[12];
    const bool valid = scal[9] != 0.0 && gx == gx;
    const bool bad = scal[10] != 0.0;
    const double gam = (valid && !bad) ? g : (valid ? INFINITY : nan(""));
    const bool inl = valid && !bad && (g < chi_gate);  // :459-463
    scal[11] = inl ? 1.0 : 0.0;
    scal[13] = gam;
  }
  XK_STAMP(6);
  __syncthreads();
  if (tid == 64) {   // after the barrier (it would wait for these stores), overlapped with the tile write
    const bool inl = scal[11] != 0.0;
    a.gamma[k] = scal[13];
    a.inlier[k] = inl ? 1 : 0;
    a.tile_rows[k] = inl ? d : 0;
    if (a.inlier_h) { a.gamma_h[k] = scal[13]; a.inlier_h[k] = inl ? 1 : 0; }
  }
  if (a.up_out) {
    // rows 0..2 of Q^T [J | Hf | res]   (msckf_update.cpp:439-443)
    double *uj = a.up_out, *uh = a.up_out + 3 * (size_t)a.n, *ur = uh + 9;
    const int N3u = 3 * a.n_poses_max;
    for (int sc_ = tid; sc_ < a.n; sc_ += XK_FEAT_THREADS) {
      double o0 = 0.0, o1 = 0.0, o2 = 0.0;
      const int c = sc_ - XK_CORE;
      int i = -1, comp = 0;
      const double *blk = nullptr;
      if (c >= 0 && c < N3u) { i = c / 3 - p0; comp = c % 3; blk = Jp; }
      else if (c >= N3u && c < 2 * N3u) { i = (c - N3u) / 3 - p0; comp = (c - N3u) % 3; blk = Ja; }
      if (blk && i >= 0 && i < L) {
        const double x0 = blk[6 * i + comp], x1 = blk[6 * i + 3 + comp];
        const int r0 = 2 * i;
        const double a0 = V[r0] * x0 + V[r0 + 1] * x1;
        const double a1 = V[m2 + r0] * x0 + V[m2 + r0 + 1] * x1;
        const double a2 = V[2 * m2 + r0] * x0 + V[2 * m2 + r0 + 1] * x1;
        const double w0 = tau0 * a0, w1 = tau1 * (a1 - w0 * g01), w2 = tau2 * (a2 - w0 * g02 - w1 * g12);
        double o[3];
        for (int r = 0; r < 3; ++r) {
          double v = -w0 * V[r] - w1 * V[m2 + r] - w2 * V[2 * m2 + r];
          if (r == r0) v += x0;
          if (r == r0 + 1) v += x1;
          o[r] = v;
        }
        o0 = o[0]; o1 = o[1]; o2 = o[2];
      }
      uj[3 * (size_t)sc_] = o0; uj[3 * (size_t)sc_ + 1] = o1; uj[3 * (size_t)sc_ + 2] = o2;
    }
    if (tid < 9) {
      const int r = tid % 3, c = tid / 3;
      uh[tid] = (r <= c) ? scal[16 + r + 3 * c] : 0.0;
    }
    if (tid < 3) ur[tid] = res[tid];
  }
  if (scal[11] == 0.0 || (!a.A && !a.Hc)) { XK_WG_END(); return; }

  if (d >= 64) {                                   // (shorter windows wrote the tile next to the single-wave gate)
    if (a.Hc) record_write(tid, XK_FEAT_THREADS);
    else tile_write(tid, XK_FEAT_THREADS);
  }
  XK_STAMP(7);
  XK_WG_END();
}

__global__ __launch_bounds__(XK_FEAT_THREADS) __attribute__((amdgpu_waves_per_eu(2))) void xk_msckf_feature(XkFeatArgs a_in) {
  xk_msckf_feature_body<false>(a_in);
}
// windows of 34..64 poses: the gate matrix as a packed triangle (two workgroups per CU instead of one)
__global__ __launch_bounds__(XK_FEAT_THREADS) __attribute__((amdgpu_waves_per_eu(2))) void xk_msckf_feature_packed(XkFeatArgs a_in) {
  xk_msckf_feature_body<true>(a_in);
}

// ----------------------------------------------------------------------------
// SLAM rows (src/x/vio/slam_update.cpp:49-214): one 64-thread workgroup per
// persistent feature; two rows at slot 2j of the SLAM tiles (zero if gated out).
// ----------------------------------------------------------------------------
struct XkSlamArgs {
  const double *q, *p;
  int n_poses, n_poses_max;
  const double *feat;      // [3M]
  const int *anchor_idxs;  // [M]
  const int *track_sizes;  // [M]
  const double *z_last;    // [M][2]
  int M;
  const double *P;
  int n;
  double var_img;
  const double *chi90;  // chi-square 0.9 quantile by dof
  int chi_len;
  double *A;  // tile base of the FIRST slam tile
  int DB, C1P, na;
  int *inlier;
  double *gamma;
};

__device__ __forceinline__ void xk_slam_rows_body(const XkSlamArgs &a, const int j) {
  __shared__ double hv[2][15];  // values of the up-to-15 nonzero columns
  __shared__ int hc[15];        // their ACTIVE column indices (-1 unused)
  __shared__ double rs[2];
  __shared__ int ok;
  const int lane = threadIdx.x;
  if (lane == 0) {
    for (int c = 0; c < 15; ++c) { hc[c] = -1; hv[0][c] = hv[1][c] = 0.0; }
    const double al = a.feat[3 * j], be = a.feat[3 * j + 1], rho = a.feat[3 * j + 2];
    const int an = a.anchor_idxs[j], pos = a.n_poses - 1, N3 = 3 * a.n_poses_max;
    double Ra[9], Rn[9];
    xk_quat_to_rot(a.q + 4 * an, Ra);
    xk_quat_to_rot(a.q + 4 * pos, Rn);
    double gp[3], dl[3], c[3];
    for (int r = 0; r < 3; ++r) gp[r] = 1.0 / rho * (Ra[3 * r] * al + Ra[3 * r + 1] * be + Ra[3 * r + 2]) + a.p[3 * an + r];
    for (int r = 0; r < 3; ++r) dl[r] = gp[r] - a.p[3 * pos + r];
    for (int r = 0; r < 3; ++r) c[r] = Rn[r] * dl[0] + Rn[3 + r] * dl[1] + Rn[6 + r] * dl[2];
    rs[0] = a.z_last[2 * j] - c[0] / c[2];
    rs[1] = a.z_last[2 * j + 1] - c[1] / c[2];
    const int fcol = 2 * N3 + 3 * j;  // active index of the feature columns
    if (an == pos) {  // :120-131
      hc[12] = fcol; hc[13] = fcol + 1; hc[14] = fcol + 2;
      hv[0][12] = 1.0; hv[1][13] = 1.0;
    } else {
      const double Ji[2][3] = {{1.0 / c[2], 0.0, -c[0] / (c[2] * c[2])}, {0.0, 1.0 / c[2], -c[1] / (c[2] * c[2])}};
      double RtRa[9], RS[9], RM[9];
      for (int x = 0; x < 3; ++x)
        for (int y = 0; y < 3; ++y) RtRa[3 * x + y] = Rn[x] * Ra[y] + Rn[3 + x] * Ra[3 + y] + Rn[6 + x] * Ra[6 + y];
      const double sk[9] = {0, -1.0, be, 1.0, 0, -al, -be, al, 0};  // Skew(alpha,beta,1)
      const double mat[9] = {1, 0, -al / rho, 0, 1, -be / rho, 0, 0, -1.0 / rho};
      for (int x = 0; x < 3; ++x)
        for (int y = 0; y < 3; ++y) {
          RS[3 * x + y] = RtRa[3 * x] * sk[y] + RtRa[3 * x + 1] * sk[3 + y] + RtRa[3 * x + 2] * sk[6 + y];
          RM[3 * x + y] = RtRa[3 * x] * mat[y] + RtRa[3 * x + 1] * mat[3 + y] + RtRa[3 * x + 2] * mat[6 + y];
        }
      for (int y = 0; y < 3; ++y) {
        hc[y] = 3 * pos + y;           // J_position
        hc[3 + y] = N3 + 3 * pos + y;  // J_attitude
        hc[6 + y] = 3 * an + y;        // J_anchor_pos
        hc[9 + y] = N3 + 3 * an + y;   // J_anchor_att
        hc[12 + y] = fcol + y;         // Hf
      }
      for (int x = 0; x < 2; ++x) {
        const double sc[3][3] = {{0, -c[2], c[1]}, {c[2], 0, -c[0]}, {-c[1], c[0], 0}};
        for (int y = 0; y < 3; ++y) {
          const double jpos = -(Ji[x][0] * Rn[3 * y] + Ji[x][1] * Rn[3 * y + 1] + Ji[x][2] * Rn[3 * y + 2]);
          hv[x][y] = jpos;
          hv[x][3 + y] = Ji[x][0] * sc[0][y] + Ji[x][1] * sc[1][y] + Ji[x][2] * sc[2][y];
          hv[x][6 + y] = -jpos;
          hv[x][9 + y] = -1.0 / rho * (Ji[x][0] * RS[y] + Ji[x][1] * RS[3 + y] + Ji[x][2] * RS[6 + y]);
          hv[x][12 + y] = 1.0 / rho * (Ji[x][0] * RM[y] + Ji[x][1] * RM[3 + y] + Ji[x][2] * RM[6 + y]);
        }
      }
    }
    // gate (:191-199): S = h P h^T + sigma^2 I over the nonzero columns
    double S[2][2] = {{a.var_img, 0}, {0, a.var_img}};
    for (int c1 = 0; c1 < 15; ++c1) {
      if (hc[c1] < 0) continue;
      for (int c2 = 0; c2 < 15; ++c2) {
        if (hc[c2] < 0) continue;
        const double pv = a.P[(size_t)(XK_CORE + hc[c1]) + (size_t)(XK_CORE + hc[c2]) * a.n];
        for (int x = 0; x < 2; ++x)
          for (int y = 0; y < 2; ++y) S[x][y] += hv[x][c1] * pv * hv[y][c2];
      }
    }
    const double det = S[0][0] * S[1][1] - S[0][1] * S[1][0];
    const double i00 = S[1][1] / det, i01 = -S[0][1] / det, i10 = -S[1][0] / det, i11 = S[0][0] / det;
    const double g = rs[0] * (i00 * rs[0] + i01 * rs[1]) + rs[1] * (i10 * rs[0] + i11 * rs[1]);
    const int dof = 2 * a.track_sizes[j];
    const double chi = (dof < a.chi_len) ? a.chi90[dof] : INFINITY;
    ok = (g < chi) ? 1 : 0;
    a.gamma[j] = g;
    a.inlier[j] = ok;
  }
  __syncthreads();
  const int row = 2 * j;
  double *r0 = a.A + ((size_t)(row / a.DB) * a.DB + (row % a.DB)) * a.C1P;
  double *r1 = a.A + ((size_t)((row + 1) / a.DB) * a.DB + ((row + 1) % a.DB)) * a.C1P;
  for (int c = lane; c < a.C1P; c += (int)blockDim.x) {
    double v0 = 0.0, v1 = 0.0;
    if (ok) {
      if (c == a.na) { v0 = rs[0]; v1 = rs[1]; }
      else
        for (int t = 0; t < 15; ++t)
          if (hc[t] == c) { v0 = hv[0][t]; v1 = hv[1][t]; }  // later blocks overwrite earlier ones
    }
    r0[c] = v0;
    r1[c] = v1;
  }
}

__global__ __launch_bounds__(64) void xk_slam_rows(XkSlamArgs a) { xk_slam_rows_body(a, (int)blockIdx.x); }

// MSCKF tracks and SLAM features of one update in ONE launch (blocks [0, K) = tracks, [K, K + M) = features): the two row
// kinds are independent (vio_updater.cpp:279-346 builds them one after the other), and as two launches the 64-thread SLAM
// kernel -- one serial lane per feature -- was 37 us behind the per-track kernel at BASELINE config 2.
__global__ __launch_bounds__(XK_FEAT_THREADS) __attribute__((amdgpu_waves_per_eu(2))) void xk_build_rows(XkFeatArgs fa, XkSlamArgs sa) {
  if ((int)blockIdx.x < fa.K) xk_msckf_feature_body<false>(fa);
  else xk_slam_rows_body(sa, (int)blockIdx.x - fa.K);
}
