/*
 * xk_oracle.c -- plain-C restatement of the xVIO EKF-update hot path, "as the
 * reference writes it" (dense per-feature A^T*jac, dense jac0*P*jac0^T, full
 * Householder QR of [H|res], general inverse of S, P=(I-KH)P; P=(P+P^T)/2).
 *
 * TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this library; the product
 * (x_multi_agent_amd/) never links or calls it.
 *
 * PARITY UNPINNED: the reference (jpl-x/x_multi_agent @ v1) has no tests, no
 * golden vectors and cannot be compiled here (Eigen3 / OpenCV / Boost absent,
 * no network).  This file and oracle/ref_np.py are two independent
 * restatements of the same reference lines; tests/test_oracle_*.py check that
 * they agree and that the invariants of SURVEY.md 8(c) hold.
 *
 * Third-party arithmetic restated from its published definition:
 *   Eigen3 (unpinned, CMakeLists.txt:181)  HouseholderQR / householderQ(),
 *       PartialPivLU inverse, Quaternion::toRotationMatrix, AngleAxis
 *   OpenCV>=3.3.1 (CMakeLists.txt:101)     cv::triangulatePoints = null
 *       vector of the 4x4 DLT system (here: one-sided Jacobi SVD)
 *   Boost.Math>=1.71 (CMakeLists.txt:117)  quantile(chi_squared(k), p) =
 *       2 * P^{-1}(k/2, p)  (inverse regularised incomplete gamma)
 *
 * One deliberate kindness to the baseline: the measurement
 * noise matrix is carried as a diagonal vector, not the dense rows x rows
 * matrix the reference materialises at vio_updater.cpp:417 (4.2 GB at the
 * headline configuration).  Result-identical.
 *
 * All matrices are column-major doubles with explicit leading dimension.
 * file:line citations are relative to /root/reference.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define XO_OK 0
#define XO_EINVAL 1
#define XO_ESINGULAR 2
#define XO_ENOMEM 3

#define K_CORE 15 /* kSizeCoreErr, include/x/common/types.h:45 */

static const double GRAV[3] = {0.0, 0.0, -9.81}; /* msckf_update.cpp:393 */

/* ------------------------------------------------------------------------ */
/* dense kernels                                                             */
/* ------------------------------------------------------------------------ */

/* C[m x n] (+)= alpha * op(A) * op(B); op selected by ta/tb (0 = N, 1 = T).
 * Column-major.  Loop order keeps the innermost loop contiguous so gcc
 * vectorises it; k is unrolled by 4 to cut C traffic. */
static void gemm(int ta, int tb, int m, int n, int k, double alpha, const double *A, int lda,
                 const double *B, int ldb, double beta, double *C, int ldc) {
  for (int j = 0; j < n; ++j) {
    double *c = C + (size_t)j * ldc;
    if (beta == 0.0)
      for (int i = 0; i < m; ++i) c[i] = 0.0;
    else if (beta != 1.0)
      for (int i = 0; i < m; ++i) c[i] *= beta;
  }
  if (!ta) {
    /* C[:,j] += A[:,l] * b(l,j) */
    const int MB = 512;
    for (int i0 = 0; i0 < m; i0 += MB) {
      int mb = (m - i0 < MB) ? m - i0 : MB;
      for (int j = 0; j < n; ++j) {
        double *c = C + (size_t)j * ldc + i0;
        int l = 0;
        for (; l + 4 <= k; l += 4) {
          double b0 = alpha * (tb ? B[j + (size_t)l * ldb] : B[l + (size_t)j * ldb]);
          double b1 = alpha * (tb ? B[j + (size_t)(l + 1) * ldb] : B[l + 1 + (size_t)j * ldb]);
          double b2 = alpha * (tb ? B[j + (size_t)(l + 2) * ldb] : B[l + 2 + (size_t)j * ldb]);
          double b3 = alpha * (tb ? B[j + (size_t)(l + 3) * ldb] : B[l + 3 + (size_t)j * ldb]);
          const double *a0 = A + (size_t)l * lda + i0, *a1 = a0 + lda, *a2 = a1 + lda, *a3 = a2 + lda;
          for (int i = 0; i < mb; ++i) c[i] += a0[i] * b0 + a1[i] * b1 + a2[i] * b2 + a3[i] * b3;
        }
        for (; l < k; ++l) {
          double b0 = alpha * (tb ? B[j + (size_t)l * ldb] : B[l + (size_t)j * ldb]);
          const double *a0 = A + (size_t)l * lda + i0;
          for (int i = 0; i < mb; ++i) c[i] += a0[i] * b0;
        }
      }
    }
  } else {
    /* C(i,j) += sum_l A(l,i) * op(B)(l,j): dot products over contiguous A columns */
    for (int j = 0; j < n; ++j) {
      for (int i = 0; i < m; ++i) {
        const double *a = A + (size_t)i * lda;
        double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
        int l = 0;
        if (!tb) {
          const double *b = B + (size_t)j * ldb;
          for (; l + 4 <= k; l += 4) {
            s0 += a[l] * b[l];
            s1 += a[l + 1] * b[l + 1];
            s2 += a[l + 2] * b[l + 2];
            s3 += a[l + 3] * b[l + 3];
          }
          for (; l < k; ++l) s0 += a[l] * b[l];
        } else {
          for (; l < k; ++l) s0 += a[l] * B[j + (size_t)l * ldb];
        }
        C[i + (size_t)j * ldc] += alpha * ((s0 + s1) + (s2 + s3));
      }
    }
  }
}

/* Inverse by LU with partial pivoting (Eigen PartialPivLU::inverse()).
 * A (n x n, lda) is overwritten; Ainv (n x n, ld n) receives the inverse. */
static int lu_inverse(int n, double *A, int lda, double *Ainv) {
  int *piv = (int *)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
  if (!piv) return XO_ENOMEM;
  for (int k = 0; k < n; ++k) {
    int p = k;
    double mx = fabs(A[k + (size_t)k * lda]);
    for (int i = k + 1; i < n; ++i) {
      double v = fabs(A[i + (size_t)k * lda]);
      if (v > mx) { mx = v; p = i; }
    }
    piv[k] = p;
    if (mx == 0.0 || mx != mx) { free(piv); return XO_ESINGULAR; }
    if (p != k)
      for (int j = 0; j < n; ++j) {
        double t = A[k + (size_t)j * lda];
        A[k + (size_t)j * lda] = A[p + (size_t)j * lda];
        A[p + (size_t)j * lda] = t;
      }
    double inv = 1.0 / A[k + (size_t)k * lda];
    for (int i = k + 1; i < n; ++i) A[i + (size_t)k * lda] *= inv;
    for (int j = k + 1; j < n; ++j) {
      double akj = A[k + (size_t)j * lda];
      double *cj = A + (size_t)j * lda;
      const double *ck = A + (size_t)k * lda;
      for (int i = k + 1; i < n; ++i) cj[i] -= ck[i] * akj;
    }
  }
  /* solve A X = I column by column: P A = L U */
  for (int j = 0; j < n; ++j) {
    double *x = Ainv + (size_t)j * n;
    for (int i = 0; i < n; ++i) x[i] = (i == j) ? 1.0 : 0.0;
    for (int k = 0; k < n; ++k)
      if (piv[k] != k) { double t = x[k]; x[k] = x[piv[k]]; x[piv[k]] = t; }
    for (int k = 0; k < n; ++k) {
      double xk = x[k];
      if (xk != 0.0) {
        const double *ck = A + (size_t)k * lda;
        for (int i = k + 1; i < n; ++i) x[i] -= ck[i] * xk;
      }
    }
    for (int k = n - 1; k >= 0; --k) {
      x[k] /= A[k + (size_t)k * lda];
      double xk = x[k];
      const double *ck = A + (size_t)k * lda;
      for (int i = 0; i < k; ++i) x[i] -= ck[i] * xk;
    }
  }
  free(piv);
  return XO_OK;
}

/* Householder reflector for x (length m), Eigen makeHouseholder convention:
 * H = I - tau [1;v][1;v]^T, H x = beta e0.  tail==0 -> tau = 0, beta = x0.
 * x[1..] overwritten with v (essential part); returns tau, *beta. */
static double make_householder(double *x, int m, double *beta) {
  double tail = 0.0;
  for (int i = 1; i < m; ++i) tail += x[i] * x[i];
  double c0 = x[0];
  if (tail <= 2.2250738585072014e-308) {
    *beta = c0;
    for (int i = 1; i < m; ++i) x[i] = 0.0;
    return 0.0;
  }
  double b = sqrt(c0 * c0 + tail);
  if (c0 >= 0) b = -b;
  for (int i = 1; i < m; ++i) x[i] /= (c0 - b);
  *beta = b;
  return (b - c0) / b;
}

/* In-place unblocked Householder QR of A (m x n, lda).  On exit the upper
 * triangle holds R, the essential parts of the reflectors sit below the
 * diagonal, tau[0..min(m,n)) the coefficients. */
static void qr_unblocked(double *A, int m, int n, int lda, double *tau) {
  int kmax = m < n ? m : n;
  for (int k = 0; k < kmax; ++k) {
    double beta;
    double *col = A + k + (size_t)k * lda;
    tau[k] = make_householder(col, m - k, &beta);
    col[0] = beta;
    if (tau[k] != 0.0)
      for (int j = k + 1; j < n; ++j) {
        double *cj = A + k + (size_t)j * lda;
        double w = cj[0];
        for (int i = 1; i < m - k; ++i) w += col[i] * cj[i];
        w *= tau[k];
        cj[0] -= w;
        for (int i = 1; i < m - k; ++i) cj[i] -= col[i] * w;
      }
  }
}

/* Form the full m x m Q from qr_unblocked's output (householderQ() assigned
 * to a Matrix, msckf_update.cpp:423). */
static void form_q(const double *A, int m, int n, int lda, const double *tau, double *Q) {
  int kmax = m < n ? m : n;
  for (size_t i = 0; i < (size_t)m * m; ++i) Q[i] = 0.0;
  for (int i = 0; i < m; ++i) Q[i + (size_t)i * m] = 1.0;
  for (int k = kmax - 1; k >= 0; --k) {
    if (tau[k] == 0.0) continue;
    const double *v = A + k + (size_t)k * lda; /* v[0] implicit 1 */
    for (int j = k; j < m; ++j) {
      double *qj = Q + k + (size_t)j * m;
      double w = qj[0];
      for (int i = 1; i < m - k; ++i) w += v[i] * qj[i];
      w *= tau[k];
      qj[0] -= w;
      for (int i = 1; i < m - k; ++i) qj[i] -= v[i] * w;
    }
  }
}

/* Blocked Householder QR (R only), compact-WY trailing updates through gemm,
 * as Eigen's HouseholderQR does for large inputs (block size 48). */
static int qr_blocked_r(double *A, int m, int n, int lda) {
  const int NB = 48;
  int kmax = m < n ? m : n;
  double *tau = (double *)malloc(sizeof(double) * (size_t)(kmax > 0 ? kmax : 1));
  double *T = (double *)malloc(sizeof(double) * NB * NB);
  double *W = (double *)malloc(sizeof(double) * NB * (size_t)(n > 0 ? n : 1));
  double *W2 = (double *)malloc(sizeof(double) * NB * (size_t)(n > 0 ? n : 1));
  if (!tau || !T || !W || !W2) { free(tau); free(T); free(W); free(W2); return XO_ENOMEM; }
  for (int k0 = 0; k0 < kmax; k0 += NB) {
    int nb = (kmax - k0 < NB) ? kmax - k0 : NB;
    int mp = m - k0;
    double *Ap = A + k0 + (size_t)k0 * lda;
    qr_unblocked(Ap, mp, nb, lda, tau + k0);
    int nt = n - k0 - nb;
    if (nt <= 0) continue;
    /* V = unit-lower part of panel (mp x nb).  Build T (upper, nb x nb):
     * T(j,j)=tau_j, T(0:j,j) = -tau_j * T(0:j,0:j) * V(:,0:j)^T v_j */
    for (int j = 0; j < nb; ++j) {
      for (int i = 0; i < nb; ++i) T[i + j * NB] = 0.0;
      double tj = tau[k0 + j];
      T[j + j * NB] = tj;
      if (j == 0 || tj == 0.0) continue;
      double z[48];
      for (int i = 0; i < j; ++i) {
        /* v_i^T v_j with implicit ones */
        const double *vi = Ap + (size_t)i * lda, *vj = Ap + (size_t)j * lda;
        double s = vi[j]; /* v_i[j] * v_j[j](=1) */
        for (int r = j + 1; r < mp; ++r) s += vi[r] * vj[r];
        z[i] = s;
      }
      for (int i = 0; i < j; ++i) {
        double s = 0.0;
        for (int l = i; l < j; ++l) s += T[i + l * NB] * z[l];
        T[i + j * NB] = -tj * s;
      }
    }
    /* W = V^T * C  (nb x nt), C = A[k0:, k0+nb:] */
    double *Cm = A + k0 + (size_t)(k0 + nb) * lda;
    /* temporarily patch the panel so V is explicit unit-lower: copy to Vx */
    double *Vx = (double *)malloc(sizeof(double) * (size_t)mp * nb);
    if (!Vx) { free(tau); free(T); free(W); free(W2); return XO_ENOMEM; }
    for (int j = 0; j < nb; ++j)
      for (int i = 0; i < mp; ++i)
        Vx[i + (size_t)j * mp] = (i < j) ? 0.0 : (i == j ? 1.0 : Ap[i + (size_t)j * lda]);
    gemm(1, 0, nb, nt, mp, 1.0, Vx, mp, Cm, lda, 0.0, W, nb);
    /* W2 = T^T * W */
    gemm(1, 0, nb, nt, nb, 1.0, T, NB, W, nb, 0.0, W2, nb);
    /* C -= V * W2 */
    gemm(0, 0, mp, nt, nb, -1.0, Vx, mp, W2, nb, 1.0, Cm, lda);
    free(Vx);
  }
  free(tau); free(T); free(W); free(W2);
  return XO_OK;
}

/* ------------------------------------------------------------------------ */
/* chi-square quantile: boost::math::quantile(chi_squared(k), p)            */
/* ------------------------------------------------------------------------ */
static double gamma_p(double a, double x) { /* regularised lower incomplete gamma */
  if (x <= 0.0) return 0.0;
  double gln = lgamma(a);
  if (x < a + 1.0) {
    double ap = a, sum = 1.0 / a, del = sum;
    for (int n = 0; n < 2000; ++n) {
      ap += 1.0;
      del *= x / ap;
      sum += del;
      if (fabs(del) < fabs(sum) * 1e-17) break;
    }
    return sum * exp(-x + a * log(x) - gln);
  }
  double b = x + 1.0 - a, c = 1.0 / 1e-300, d = 1.0 / b, h = d;
  for (int i = 1; i < 2000; ++i) {
    double an = -i * (i - a);
    b += 2.0;
    d = an * d + b;
    if (fabs(d) < 1e-300) d = 1e-300;
    c = b + an / c;
    if (fabs(c) < 1e-300) c = 1e-300;
    d = 1.0 / d;
    double del = d * c;
    h *= del;
    if (fabs(del - 1.0) < 1e-16) break;
  }
  return 1.0 - exp(-x + a * log(x) - gln) * h;
}

int xo_chi2inv(double p, int dof, double *out) {
  if (!(p > 0.0 && p < 1.0) || dof < 1) return XO_EINVAL;
  double a = 0.5 * dof;
  /* Wilson-Hilferty start, then safeguarded Newton on P(a, x/2) = p */
  double z;
  { /* inverse normal via bisection on erfc -- cold path, accuracy irrelevant */
    double lo = -10, hi = 10;
    for (int i = 0; i < 200; ++i) {
      double mid = 0.5 * (lo + hi);
      if (0.5 * erfc(-mid / sqrt(2.0)) < p) lo = mid; else hi = mid;
    }
    z = 0.5 * (lo + hi);
  }
  double t = 1.0 - 2.0 / (9.0 * dof) + z * sqrt(2.0 / (9.0 * dof));
  double x = dof * t * t * t;
  if (!(x > 0.0)) x = 1e-3;
  double lo = 0.0, hi = INFINITY;
  for (int it = 0; it < 200; ++it) {
    double f = gamma_p(a, 0.5 * x) - p;
    if (f > 0) hi = x; else lo = x;
    /* pdf of chi2 */
    double pdf = exp((a - 1.0) * log(0.5 * x) - 0.5 * x - lgamma(a)) * 0.5;
    double xn = x - f / pdf;
    if (!(xn > lo && xn < hi)) xn = isinf(hi) ? 2.0 * x : 0.5 * (lo + hi);
    if (fabs(xn - x) <= 4e-16 * fabs(xn)) { x = xn; break; }
    x = xn;
  }
  *out = x;
  return XO_OK;
}

/* ------------------------------------------------------------------------ */
/* small helpers                                                             */
/* ------------------------------------------------------------------------ */
/* q.normalized().toRotationMatrix(), q stored x,y,z,w; R row-major r[3][3];
 * maps camera -> world (msckf_update.cpp:339-340). */
static void quat_to_rot(const double *q, double r[3][3]) {
  double nn = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  double x = q[0] / nn, y = q[1] / nn, z = q[2] / nn, w = q[3] / nn;
  double tx = 2 * x, ty = 2 * y, tz = 2 * z;
  double twx = tx * w, twy = ty * w, twz = tz * w;
  double txx = tx * x, txy = ty * x, txz = tz * x;
  double tyy = ty * y, tyz = tz * y, tzz = tz * z;
  r[0][0] = 1 - (tyy + tzz); r[0][1] = txy - twz;       r[0][2] = txz + twy;
  r[1][0] = txy + twz;       r[1][1] = 1 - (txx + tzz); r[1][2] = tyz - twx;
  r[2][0] = txz - twy;       r[2][1] = tyz + twx;       r[2][2] = 1 - (txx + tyy);
}

static void skew3(const double v[3], double s[3][3]) { /* tools.h:57-65 */
  s[0][0] = 0;     s[0][1] = -v[2]; s[0][2] = v[1];
  s[1][0] = v[2];  s[1][1] = 0;     s[1][2] = -v[0];
  s[2][0] = -v[1]; s[2][1] = v[0];  s[2][2] = 0;
}

/* null vector (smallest right singular vector) of a 4x4 row-major matrix by
 * one-sided Jacobi (Hestenes) SVD -- stands in for cv::SVD inside
 * cv::triangulatePoints. */
static void null4(const double a_in[4][4], double x[4]) {
  double a[4][4], v[4][4];
  memcpy(a, a_in, sizeof(a));
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) v[i][j] = (i == j);
  for (int sweep = 0; sweep < 60; ++sweep) {
    double off = 0.0;
    for (int p = 0; p < 3; ++p)
      for (int q = p + 1; q < 4; ++q) {
        double al = 0, be = 0, ga = 0;
        for (int i = 0; i < 4; ++i) {
          al += a[i][p] * a[i][p];
          be += a[i][q] * a[i][q];
          ga += a[i][p] * a[i][q];
        }
        if (ga == 0.0) continue;
        double lim = sqrt(al * be);
        if (fabs(ga) <= 1e-300 || fabs(ga) <= 1e-17 * lim) continue;
        off = fmax(off, fabs(ga) / (lim > 0 ? lim : 1.0));
        double zeta = (be - al) / (2.0 * ga);
        double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
        double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
        for (int i = 0; i < 4; ++i) {
          double ap = a[i][p], aq = a[i][q];
          a[i][p] = c * ap - s * aq;
          a[i][q] = s * ap + c * aq;
          double vp = v[i][p], vq = v[i][q];
          v[i][p] = c * vp - s * vq;
          v[i][q] = s * vp + c * vq;
        }
      }
    if (off < 1e-16) break;
  }
  int best = 0;
  double bn = INFINITY;
  for (int j = 0; j < 4; ++j) {
    double nn = 0;
    for (int i = 0; i < 4; ++i) nn += a[i][j] * a[i][j];
    if (nn < bn) { bn = nn; best = j; }
  }
  for (int i = 0; i < 4; ++i) x[i] = v[i][best];
}

/* ------------------------------------------------------------------------ */
/* Triangulation (src/x/vision/triangulation.cpp)                           */
/* ------------------------------------------------------------------------ */
/* ctor :48-79 + triangulateGN :102-206.  q [L*4] xyzw, p [L*3], obs [L*2].
 * Output ivd = (alpha,beta,rho) anchored in the last pose. */
int xo_triangulate_gn(const double *q, const double *p, const double *obs, int L, int max_iter,
                      double term, double *ivd, int *iters) {
  if (L < 2) return XO_EINVAL;
  double(*rot)[3][3] = malloc(sizeof(double[3][3]) * (size_t)L); /* world->camera */
  double *Jm = malloc(sizeof(double) * 6 * (size_t)L), *rv = malloc(sizeof(double) * 2 * (size_t)L);
  if (!rot || !Jm || !rv) { free(rot); free(Jm); free(rv); return XO_ENOMEM; }
  for (int i = 0; i < L; ++i) {
    double r[3][3];
    quat_to_rot(q + 4 * i, r);
    for (int a = 0; a < 3; ++a)
      for (int b = 0; b < 3; ++b) rot[i][a][b] = r[b][a]; /* transpose, :70 */
  }
  /* projection matrices [R | -R p] of first and last pose, :208-216 */
  int i1 = 0, i2 = L - 1;
  double P1[3][4], P2[3][4];
  for (int a = 0; a < 3; ++a) {
    double t1 = 0, t2 = 0;
    for (int b = 0; b < 3; ++b) {
      P1[a][b] = rot[i1][a][b];
      P2[a][b] = rot[i2][a][b];
      t1 -= rot[i1][a][b] * p[3 * i1 + b];
      t2 -= rot[i2][a][b] * p[3 * i2 + b];
    }
    P1[a][3] = t1;
    P2[a][3] = t2;
  }
  double A[4][4], X[4];
  for (int c = 0; c < 4; ++c) {
    A[0][c] = obs[2 * i1] * P1[2][c] - P1[0][c];
    A[1][c] = obs[2 * i1 + 1] * P1[2][c] - P1[1][c];
    A[2][c] = obs[2 * i2] * P2[2][c] - P2[0][c];
    A[3][c] = obs[2 * i2 + 1] * P2[2][c] - P2[1][c];
  }
  null4(A, X);
  double pw[4] = {X[0] / X[3], X[1] / X[3], X[2] / X[3], 1.0};
  double pc2[3];
  for (int a = 0; a < 3; ++a) pc2[a] = P2[a][0] * pw[0] + P2[a][1] * pw[1] + P2[a][2] * pw[2] + P2[a][3];
  double alpha = pc2[0] / pc2[2], beta = pc2[1] / pc2[2], rho = 1.0 / pc2[2];
  const double *pa = p + 3 * i2;
  double r_norm_last = 1000.0, r_norm = 100.0;
  int iter = 0;
  while (r_norm_last - r_norm > term) { /* :149 */
    iter++;
    if (iter > max_iter) break;
    for (int i = i1; i <= i2; ++i) {
      double drot[3][3], dpos[3];
      for (int a = 0; a < 3; ++a) {
        for (int b = 0; b < 3; ++b) {
          double s = 0;
          for (int c = 0; c < 3; ++c) s += rot[i][a][c] * rot[i2][b][c]; /* rot * rot_a^T */
          drot[a][b] = s;
        }
        double sa = 0, sp = 0;
        for (int c = 0; c < 3; ++c) { sa += rot[i][a][c] * pa[c]; sp += rot[i][a][c] * p[3 * i + c]; }
        dpos[a] = sa - sp;
      }
      double h[3];
      for (int a = 0; a < 3; ++a) h[a] = drot[a][0] * alpha + drot[a][1] * beta + drot[a][2] + rho * dpos[a];
      int k = i - i1;
      rv[2 * k] = obs[2 * k] - h[0] / h[2];
      rv[2 * k + 1] = obs[2 * k + 1] - h[1] / h[2];
      double j1[2][3] = {{-1.0 / h[2], 0.0, h[0] / (h[2] * h[2])}, {0.0, -1.0 / h[2], h[1] / (h[2] * h[2])}};
      for (int a = 0; a < 2; ++a)
        for (int b = 0; b < 3; ++b) {
          double s = 0;
          for (int c = 0; c < 3; ++c) s += j1[a][c] * (b < 2 ? drot[c][b] : dpos[c]);
          Jm[(2 * k + a) * 3 + b] = s; /* row-major 2L x 3 */
        }
    }
    double JtJ[9] = {0}, Jtr[3] = {0}, inv[9], rn = 0;
    for (int r = 0; r < 2 * L; ++r) {
      for (int a = 0; a < 3; ++a) {
        for (int b = 0; b < 3; ++b) JtJ[a + 3 * b] += Jm[r * 3 + a] * Jm[r * 3 + b];
        Jtr[a] += Jm[r * 3 + a] * rv[r];
      }
      rn += rv[r] * rv[r];
    }
    if (lu_inverse(3, JtJ, 3, inv) != XO_OK) { free(rot); free(Jm); free(rv); return XO_ESINGULAR; }
    double d[3];
    for (int a = 0; a < 3; ++a) d[a] = inv[a] * Jtr[0] + inv[a + 3] * Jtr[1] + inv[a + 6] * Jtr[2];
    alpha -= d[0];
    beta -= d[1];
    rho -= d[2];
    r_norm_last = r_norm;
    r_norm = sqrt(rn);
  }
  ivd[0] = alpha; ivd[1] = beta; ivd[2] = rho;
  if (iters) *iters = iter;
  free(rot); free(Jm); free(rv);
  return XO_OK;
}

/* msckf_update.cpp:283-304 */
static void global_feature_position(const double ivd[3], const double *q_last, const double *p_last,
                                    double out[3]) {
  double r[3][3];
  quat_to_rot(q_last, r);
  for (int a = 0; a < 3; ++a)
    out[a] = (1.0 / ivd[2]) * (r[a][0] * ivd[0] + r[a][1] * ivd[1] + r[a][2]) + p_last[a];
}

/* ------------------------------------------------------------------------ */
/* MsckfUpdate::processOneTrack (msckf_update.cpp:306-492)                   */
/* ------------------------------------------------------------------------ */
typedef struct {
  int valid, inlier;
  double gamma;
  double *jac0; /* (2L-3) x n, ld = 2L-3 */
  double *res0; /* 2L-3 */
  double up_hf[9], up_res[3]; /* A_up^T Hf (col-major 3x3), A_up^T res */
  double *up_jac;             /* 3 x n, ld 3 (A_up^T jac) */
} track_out_t;

static void track_out_free(track_out_t *t) { free(t->jac0); free(t->res0); free(t->up_jac); }

/* msckf_update.cpp:328-417: per observation the residual, the 2 x 3 position / attitude Jacobians with the observability
 * constraint, scattered into jac (m x n, ld m), hf (m x 3), res (m), m = 2 L.  Returns 1 if a camera-frame point is NaN
 * (the reference drops such a track, :349-357), 0 otherwise.  jac / hf / res must be zero on entry. */
static int track_jacobians(const double *obs, int L, int n, const double *C_q_G, const double *G_p_C, int n_poses,
                           int n_poses_max, const double gpf[3], double *jac, double *hf, double *res) {
  (void)n;
  const int m = 2 * L;
  for (int i = 0; i < L; ++i) { /* :328-417 */
    int pos = n_poses - L + i;
    double R[3][3], c[3], dlt[3];
    quat_to_rot(C_q_G + 4 * pos, R);
    for (int a = 0; a < 3; ++a) dlt[a] = gpf[a] - G_p_C[3 * pos + a];
    for (int a = 0; a < 3; ++a) c[a] = R[0][a] * dlt[0] + R[1][a] * dlt[1] + R[2][a] * dlt[2];
    if (!(c[0] == c[0] && c[1] == c[1] && c[2] == c[2])) return 1; /* :349-357 */
    res[2 * i] = obs[2 * i] - c[0] / c[2];
    res[2 * i + 1] = obs[2 * i + 1] - c[1] / c[2];
    double Ji[2][3] = {{1.0 / c[2], 0.0, -c[0] / (c[2] * c[2])}, {0.0, 1.0 / c[2], -c[1] / (c[2] * c[2])}};
    double Jp[2][3], Ja[2][3], S[3][3];
    for (int a = 0; a < 2; ++a)
      for (int b = 0; b < 3; ++b) {
        double s = 0;
        for (int k = 0; k < 3; ++k) s += Ji[a][k] * R[b][k]; /* Ji * R^T */
        Jp[a][b] = -s;
      }
    skew3(c, S);
    for (int a = 0; a < 2; ++a)
      for (int b = 0; b < 3; ++b) Ja[a][b] = Ji[a][0] * S[0][b] + Ji[a][1] * S[1][b] + Ji[a][2] * S[2][b];
    /* observability constraint :393-406 */
    double u[3], uu, t[2];
    for (int a = 0; a < 3; ++a) u[a] = R[a][0] * GRAV[0] + R[a][1] * GRAV[1] + R[a][2] * GRAV[2];
    uu = u[0] * u[0] + u[1] * u[1] + u[2] * u[2];
    for (int a = 0; a < 2; ++a) t[a] = Jp[a][0] * u[0] + Jp[a][1] * u[1] + Jp[a][2] * u[2];
    for (int a = 0; a < 2; ++a)
      for (int b = 0; b < 3; ++b) Jp[a][b] = Jp[a][b] - t[a] * (1.0 / uu) * u[b];
    skew3(dlt, S);
    for (int a = 0; a < 3; ++a) u[a] = S[a][0] * GRAV[0] + S[a][1] * GRAV[1] + S[a][2] * GRAV[2];
    uu = u[0] * u[0] + u[1] * u[1] + u[2] * u[2];
    for (int a = 0; a < 2; ++a) t[a] = Ja[a][0] * u[0] + Ja[a][1] * u[1] + Ja[a][2] * u[2];
    for (int a = 0; a < 2; ++a)
      for (int b = 0; b < 3; ++b) Ja[a][b] = Ja[a][b] - t[a] * (1.0 / uu) * u[b];
    int cp = K_CORE + 3 * pos, ca = cp + 3 * n_poses_max;
    for (int a = 0; a < 2; ++a)
      for (int b = 0; b < 3; ++b) {
        hf[(2 * i + a) + (size_t)b * m] = -Jp[a][b]; /* :409 */
        jac[(2 * i + a) + (size_t)(cp + b) * m] = Jp[a][b];
        jac[(2 * i + a) + (size_t)(ca + b) * m] = Ja[a][b];
      }
  }
  return 0;
}

/* The same for a caller outside this file (oracle/eigen_variant.cpp is handed these matrices: the per-track inputs of the
 * linear algebra the reference does with Eigen, msckf_update.cpp:423-463).  gpf = the track's triangulated landmark
 * (xo_msckf_update's `feats`).  Column-major jac (2L x n), hf (2L x 3), res (2L); *nan_track = 1 if the track is dropped. */
int xo_track_jacobians(const double *obs, int L, int n, const double *C_q_G, const double *G_p_C, int n_poses,
                       int n_poses_max, const double *gpf, double *jac, double *hf, double *res, int *nan_track) {
  if (!obs || L < 2 || !jac || !hf || !res) return XO_EINVAL;
  memset(jac, 0, sizeof(double) * (size_t)2 * L * n);
  memset(hf, 0, sizeof(double) * (size_t)2 * L * 3);
  memset(res, 0, sizeof(double) * (size_t)2 * L);
  const int bad = track_jacobians(obs, L, n, C_q_G, G_p_C, n_poses, n_poses_max, gpf, jac, hf, res);
  if (nan_track) *nan_track = bad;
  return XO_OK;
}

static int process_one_track(const double *obs, int L, const double *P, int n, const double *C_q_G,
                             const double *G_p_C, int n_poses, int n_poses_max, double var_img,
                             const double gpf[3], int want_up, track_out_t *o) {
  memset(o, 0, sizeof(*o));
  int m = 2 * L, d = m - 3;
  double *jac = calloc((size_t)m * n, sizeof(double));
  double *hf = calloc((size_t)m * 3, sizeof(double));
  double *res = calloc((size_t)m, sizeof(double));
  double *Q = malloc(sizeof(double) * (size_t)m * m);
  if (!jac || !hf || !res || !Q) { free(jac); free(hf); free(res); free(Q); return XO_ENOMEM; }
  if (track_jacobians(obs, L, n, C_q_G, G_p_C, n_poses, n_poses_max, gpf, jac, hf, res)) { /* :349-357 */
    free(jac); free(hf); free(res); free(Q);
    o->valid = 0; o->gamma = NAN;
    return XO_OK;
  }
  /* nullspace :423-432 */
  double hfqr[3 * 512], tau[3];
  double *hq = (m <= 512) ? hfqr : malloc(sizeof(double) * 3 * (size_t)m);
  memcpy(hq, hf, sizeof(double) * 3 * (size_t)m);
  qr_unblocked(hq, m, 3, m, tau);
  form_q(hq, m, 3, m, tau, Q);
  const double *A = Q + (size_t)3 * m; /* Q[:,3:] */
  o->res0 = malloc(sizeof(double) * (size_t)d);
  o->jac0 = malloc(sizeof(double) * (size_t)d * n);
  gemm(1, 0, d, 1, m, 1.0, A, m, res, m, 0.0, o->res0, d);
  gemm(1, 0, d, n, m, 1.0, A, m, jac, m, 0.0, o->jac0, d);
  if (want_up) { /* :439-443 */
    o->up_jac = malloc(sizeof(double) * 3 * (size_t)n);
    gemm(1, 0, 3, n, m, 1.0, Q, m, jac, m, 0.0, o->up_jac, 3);
    gemm(1, 0, 3, 3, m, 1.0, Q, m, hf, m, 0.0, o->up_hf, 3);
    gemm(1, 0, 3, 1, m, 1.0, Q, m, res, m, 0.0, o->up_res, 3);
  }
  /* gate :452-463 */
  double *JP = malloc(sizeof(double) * (size_t)d * n);
  double *S = malloc(sizeof(double) * (size_t)d * d), *Si = malloc(sizeof(double) * (size_t)d * d);
  gemm(0, 0, d, n, n, 1.0, o->jac0, d, P, n, 0.0, JP, d);
  gemm(0, 1, d, d, n, 1.0, JP, d, o->jac0, d, 0.0, S, d);
  for (int i = 0; i < d; ++i) S[i + (size_t)i * d] += var_img;
  int rc = lu_inverse(d, S, d, Si);
  double g = 0.0;
  if (rc == XO_OK) {
    for (int j = 0; j < d; ++j) {
      double s = 0;
      for (int i = 0; i < d; ++i) s += o->res0[i] * Si[i + (size_t)j * d];
      g += s * o->res0[j];
    }
  } else g = INFINITY;
  double chi;
  xo_chi2inv(0.95, 2 * L - 3, &chi);
  o->valid = 1;
  o->gamma = g;
  o->inlier = g < chi;
  free(JP); free(S); free(Si);
  if (hq != hfqr) free(hq);
  free(jac); free(hf); free(res); free(Q);
  return XO_OK;
}

/* MsckfUpdate ctor, single agent: msckf_update.cpp:27-63, :65-173.
 * trk_off [K+1] prefix offsets into obs_xy [sum L, 2]; a length-L track sees
 * the last L of the n_poses window poses.  jac is rows x n (ld rows) with
 * rows = 2*n_obs - 3K pre-sized for ALL tracks; rejected tracks leave
 * trailing zero rows with cov_diag 1 (Q1). */
int xo_msckf_update(const double *C_q_G, const double *G_p_C, int n_poses, const int *trk_off,
                    const double *obs_xy, int K, const double *P, int n, int n_poses_max,
                    double sigma_img, double *jac, double *res, double *cov_diag, int rows,
                    int *inlier, double *gamma, double *feats, int *gn_iters, int *rows_used) {
  int n_obs = trk_off[K] - trk_off[0];
  if (rows != 2 * n_obs - 3 * K) return XO_EINVAL;
  double var_img = sigma_img * sigma_img;
  for (size_t i = 0; i < (size_t)rows * n; ++i) jac[i] = 0.0;
  for (int i = 0; i < rows; ++i) { res[i] = 0.0; cov_diag[i] = 1.0; }
  int row_h = 0;
  for (int k = 0; k < K; ++k) {
    int L = trk_off[k + 1] - trk_off[k];
    const double *obs = obs_xy + 2 * (size_t)trk_off[k];
    const double *ql = C_q_G + 4 * (size_t)(n_poses - L), *pl = G_p_C + 3 * (size_t)(n_poses - L);
    double ivd[3], gpf[3];
    int it = 0;
    int rc = xo_triangulate_gn(ql, pl, obs, L, 10, 1e-5, ivd, &it); /* vio_updater.cpp:289-290 */
    if (rc != XO_OK) { ivd[0] = ivd[1] = ivd[2] = NAN; }
    global_feature_position(ivd, ql + 4 * (L - 1), pl + 3 * (L - 1), gpf);
    if (feats) { feats[3 * k] = gpf[0]; feats[3 * k + 1] = gpf[1]; feats[3 * k + 2] = gpf[2]; }
    if (gn_iters) gn_iters[k] = it;
    track_out_t o;
    rc = process_one_track(obs, L, P, n, C_q_G, G_p_C, n_poses, n_poses_max, var_img, gpf, 0, &o);
    if (rc != XO_OK) return rc;
    if (gamma) gamma[k] = o.gamma;
    if (inlier) inlier[k] = 0;
    if (o.valid && o.inlier) {
      int d = 2 * L - 3;
      for (int j = 0; j < n; ++j) memcpy(jac + row_h + (size_t)j * rows, o.jac0 + (size_t)j * d, sizeof(double) * d);
      memcpy(res + row_h, o.res0, sizeof(double) * d);
      for (int i = 0; i < d; ++i) cov_diag[row_h + i] = var_img;
      row_h += d;
      if (inlier) inlier[k] = 1;
    }
    track_out_free(&o);
  }
  if (rows_used) *rows_used = row_h;
  return XO_OK;
}

/* ------------------------------------------------------------------------ */
/* SlamUpdate (src/x/vio/slam_update.cpp:25-214)                             */
/* ------------------------------------------------------------------------ */
static void mat33_mul(const double a[3][3], const double b[3][3], double c[3][3]) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) c[i][j] = a[i][0] * b[0][j] + a[i][1] * b[1][j] + a[i][2] * b[2][j];
}

int xo_slam_update(const double *C_q_G, const double *G_p_C, int n_poses, const double *feat,
                   const int *anchor_idxs, const int *track_sizes, const double *z_last, int M,
                   const double *P, int n, int n_poses_max, double sigma_img, double *jac,
                   double *res, double *cov_diag, int *inlier, double *gamma, int *rows_used) {
  int rows = 2 * M;
  double var_img = sigma_img * sigma_img;
  for (size_t i = 0; i < (size_t)rows * n; ++i) jac[i] = 0.0;
  for (int i = 0; i < rows; ++i) { res[i] = 0.0; cov_diag[i] = 1.0; }
  double *h = malloc(sizeof(double) * 2 * (size_t)n), *hp = malloc(sizeof(double) * 2 * (size_t)n);
  if (!h || !hp) { free(h); free(hp); return XO_ENOMEM; }
  int row_h = 0;
  for (int j = 0; j < M; ++j) {
    for (int i = 0; i < 2 * n; ++i) h[i] = 0.0;
    double al = feat[3 * j], be = feat[3 * j + 1], rho = feat[3 * j + 2];
    int a = anchor_idxs[j];
    double Ra[3][3], Rn[3][3], gpf[3], d[3], c[3];
    quat_to_rot(C_q_G + 4 * a, Ra);
    for (int k = 0; k < 3; ++k) gpf[k] = 1.0 / rho * (Ra[k][0] * al + Ra[k][1] * be + Ra[k][2]) + G_p_C[3 * a + k];
    int pos = n_poses - 1;
    quat_to_rot(C_q_G + 4 * pos, Rn);
    for (int k = 0; k < 3; ++k) d[k] = gpf[k] - G_p_C[3 * pos + k];
    for (int k = 0; k < 3; ++k) c[k] = Rn[0][k] * d[0] + Rn[1][k] * d[1] + Rn[2][k] * d[2];
    double r2[2] = {z_last[2 * j] - c[0] / c[2], z_last[2 * j + 1] - c[1] / c[2]};
    int fcol = K_CORE + (2 * n_poses_max + j) * 3;
    if (a == pos) { /* :120-131 */
      h[0 + 2 * (size_t)fcol] = 1.0;
      h[1 + 2 * (size_t)(fcol + 1)] = 1.0;
    } else {
      double Ji[2][3] = {{1.0 / c[2], 0.0, -c[0] / (c[2] * c[2])}, {0.0, 1.0 / c[2], -c[1] / (c[2] * c[2])}};
      double S[3][3], RnT[3][3], RtRa[3][3], M1[3][3], M2[3][3];
      skew3(c, S);
      for (int x = 0; x < 3; ++x)
        for (int y = 0; y < 3; ++y) RnT[x][y] = Rn[y][x];
      mat33_mul(RnT, Ra, RtRa);
      double sk[3] = {al, be, 1.0};
      skew3(sk, M1);
      double RS[3][3];
      mat33_mul(RtRa, M1, RS);
      double mat[3][3] = {{1, 0, -al / rho}, {0, 1, -be / rho}, {0, 0, -1.0 / rho}};
      mat33_mul(RtRa, mat, M2);
      int cp = K_CORE + 3 * pos, cpa = cp + 3 * n_poses_max;
      int ap = K_CORE + 3 * a, aa = ap + 3 * n_poses_max;
      for (int x = 0; x < 2; ++x)
        for (int y = 0; y < 3; ++y) {
          double jatt = Ji[x][0] * S[0][y] + Ji[x][1] * S[1][y] + Ji[x][2] * S[2][y];
          double jpos = -(Ji[x][0] * RnT[0][y] + Ji[x][1] * RnT[1][y] + Ji[x][2] * RnT[2][y]);
          double janc_att = -1.0 / rho * (Ji[x][0] * RS[0][y] + Ji[x][1] * RS[1][y] + Ji[x][2] * RS[2][y]);
          double hfv = 1.0 / rho * (Ji[x][0] * M2[0][y] + Ji[x][1] * M2[1][y] + Ji[x][2] * M2[2][y]);
          h[x + 2 * (size_t)(cp + y)] = jpos;
          h[x + 2 * (size_t)(cpa + y)] = jatt;
          h[x + 2 * (size_t)(ap + y)] = -jpos; /* later assignment wins as in the reference */
          h[x + 2 * (size_t)(aa + y)] = janc_att;
          h[x + 2 * (size_t)(fcol + y)] = hfv;
        }
    }
    /* gate :191-199 */
    gemm(0, 0, 2, n, n, 1.0, h, 2, P, n, 0.0, hp, 2);
    double S2[4], Si[4];
    gemm(0, 1, 2, 2, n, 1.0, hp, 2, h, 2, 0.0, S2, 2);
    S2[0] += var_img;
    S2[3] += var_img;
    double g = INFINITY;
    if (lu_inverse(2, S2, 2, Si) == XO_OK)
      g = r2[0] * (Si[0] * r2[0] + Si[2] * r2[1]) + r2[1] * (Si[1] * r2[0] + Si[3] * r2[1]);
    double chi;
    xo_chi2inv(0.9, 2 * track_sizes[j], &chi);
    if (gamma) gamma[j] = g;
    if (inlier) inlier[j] = 0;
    if (g < chi) {
      for (int col = 0; col < n; ++col) {
        jac[row_h + (size_t)col * rows] = h[0 + 2 * (size_t)col];
        jac[row_h + 1 + (size_t)col * rows] = h[1 + 2 * (size_t)col];
      }
      res[row_h] = r2[0];
      res[row_h + 1] = r2[1];
      cov_diag[row_h] = cov_diag[row_h + 1] = var_img;
      row_h += 2;
      if (inlier) inlier[j] = 1;
    }
  }
  if (rows_used) *rows_used = row_h;
  free(h); free(hp);
  return XO_OK;
}

/* ------------------------------------------------------------------------ */
/* VioUpdater::applyQRDecomposition (vio_updater.cpp:487-512)                */
/* ------------------------------------------------------------------------ */
/* h (rows x cols, ld rows) and res are inputs.  If rows > cols+1: h_out
 * (cols x cols, ld cols), res_out (cols), r_out (cols) = sigma^2, did_qr=1.
 * Otherwise did_qr=0 and the outputs are untouched. */
int xo_qr_compress(const double *h, int rows, int cols, const double *res, double sigma_img,
                   double *h_out, double *res_out, double *r_out, int *did_qr) {
  *did_qr = 0;
  if (!(rows > cols + 1)) return XO_OK;
  int c1 = cols + 1;
  double *hr = malloc(sizeof(double) * (size_t)rows * c1);
  if (!hr) return XO_ENOMEM;
  memcpy(hr, h, sizeof(double) * (size_t)rows * cols);
  memcpy(hr + (size_t)rows * cols, res, sizeof(double) * rows);
  int rc = qr_blocked_r(hr, rows, c1, rows);
  if (rc != XO_OK) { free(hr); return rc; }
  for (int j = 0; j < cols; ++j)
    for (int i = 0; i < cols; ++i) h_out[i + (size_t)j * cols] = (i <= j) ? hr[i + (size_t)j * rows] : 0.0;
  for (int i = 0; i < cols; ++i) res_out[i] = hr[i + (size_t)cols * rows];
  for (int i = 0; i < cols; ++i) r_out[i] = sigma_img * sigma_img;
  *did_qr = 1;
  free(hr);
  return XO_OK;
}

/* ------------------------------------------------------------------------ */
/* Updater::applyUpdate / applyCI (updater.cpp:117-161)                      */
/* ------------------------------------------------------------------------ */
/* P n x n in/out; H m x n (ld m); r_diag m.  correction_total (n) in/out. */
int xo_apply_update(double *P, int n, const double *H, int m, const double *res, const double *r_diag,
                    double *correction_total, int cov_update, double *correction) {
  double *HP = malloc(sizeof(double) * (size_t)m * n), *S = malloc(sizeof(double) * (size_t)m * m);
  double *Si = malloc(sizeof(double) * (size_t)m * m), *PHt = malloc(sizeof(double) * (size_t)n * m);
  double *Kg = malloc(sizeof(double) * (size_t)n * m), *t = malloc(sizeof(double) * (size_t)m);
  double *IKH = NULL, *Pn = NULL;
  if (!HP || !S || !Si || !PHt || !Kg || !t) goto oom;
  gemm(0, 0, m, n, n, 1.0, H, m, P, n, 0.0, HP, m);
  gemm(0, 1, m, m, n, 1.0, HP, m, H, m, 0.0, S, m);
  for (int i = 0; i < m; ++i) S[i + (size_t)i * m] += r_diag[i];
  if (lu_inverse(m, S, m, Si) != XO_OK) { free(HP); free(S); free(Si); free(PHt); free(Kg); free(t); return XO_ESINGULAR; }
  gemm(0, 1, n, m, n, 1.0, P, n, H, m, 0.0, PHt, n);
  gemm(0, 0, n, m, m, 1.0, PHt, n, Si, m, 0.0, Kg, n);
  /* correction = K (res + H corr_tot) - corr_tot   :126 */
  for (int i = 0; i < m; ++i) t[i] = res[i];
  gemm(0, 0, m, 1, n, 1.0, H, m, correction_total, n, 1.0, t, m);
  gemm(0, 0, n, 1, m, 1.0, Kg, n, t, m, 0.0, correction, n);
  for (int i = 0; i < n; ++i) correction[i] -= correction_total[i];
  if (cov_update) { /* :131-133 */
    IKH = malloc(sizeof(double) * (size_t)n * n);
    Pn = malloc(sizeof(double) * (size_t)n * n);
    if (!IKH || !Pn) goto oom;
    gemm(0, 0, n, n, m, -1.0, Kg, n, H, m, 0.0, IKH, n);
    for (int i = 0; i < n; ++i) IKH[i + (size_t)i * n] += 1.0;
    gemm(0, 0, n, n, n, 1.0, IKH, n, P, n, 0.0, Pn, n);
    for (int j = 0; j < n; ++j)
      for (int i = 0; i < n; ++i) P[i + (size_t)j * n] = 0.5 * (Pn[i + (size_t)j * n] + Pn[j + (size_t)i * n]);
  }
  for (int i = 0; i < n; ++i) correction_total[i] += correction[i]; /* :140 */
  free(HP); free(S); free(Si); free(PHt); free(Kg); free(t); free(IKH); free(Pn);
  return XO_OK;
oom:
  free(HP); free(S); free(Si); free(PHt); free(Kg); free(t); free(IKH); free(Pn);
  return XO_ENOMEM;
}

/* applyCI: P_out = sym((I - K H) ci_P), K = ci_P H^T S^-1, corr = K res. */
int xo_apply_ci(double *P_out, const double *ci_P, int n, const double *H, int m, const double *res,
                const double *S_in, double *correction) {
  double *S = malloc(sizeof(double) * (size_t)m * m), *Si = malloc(sizeof(double) * (size_t)m * m);
  double *PHt = malloc(sizeof(double) * (size_t)n * m), *Kg = malloc(sizeof(double) * (size_t)n * m);
  double *IKH = malloc(sizeof(double) * (size_t)n * n), *Pn = malloc(sizeof(double) * (size_t)n * n);
  int rc = XO_OK;
  if (!S || !Si || !PHt || !Kg || !IKH || !Pn) { rc = XO_ENOMEM; goto done; }
  memcpy(S, S_in, sizeof(double) * (size_t)m * m);
  if (lu_inverse(m, S, m, Si) != XO_OK) { rc = XO_ESINGULAR; goto done; }
  gemm(0, 1, n, m, n, 1.0, ci_P, n, H, m, 0.0, PHt, n);
  gemm(0, 0, n, m, m, 1.0, PHt, n, Si, m, 0.0, Kg, n);
  gemm(0, 0, n, 1, m, 1.0, Kg, n, res, m, 0.0, correction, n);
  gemm(0, 0, n, n, m, -1.0, Kg, n, H, m, 0.0, IKH, n);
  for (int i = 0; i < n; ++i) IKH[i + (size_t)i * n] += 1.0;
  gemm(0, 0, n, n, n, 1.0, IKH, n, ci_P, n, 0.0, Pn, n);
  for (int j = 0; j < n; ++j)
    for (int i = 0; i < n; ++i) P_out[i + (size_t)j * n] = 0.5 * (Pn[i + (size_t)j * n] + Pn[j + (size_t)i * n]);
done:
  free(S); free(Si); free(PHt); free(Kg); free(IKH); free(Pn);
  return rc;
}

/* ------------------------------------------------------------------------ */
/* State::correct (state.cpp:197-249, 273-283)                               */
/* ------------------------------------------------------------------------ */
static void error_quat(const double *dth, double q[4]) {
  double nn = sqrt(dth[0] * dth[0] + dth[1] * dth[1] + dth[2] * dth[2]);
  if (nn == 0.0) { q[0] = q[1] = q[2] = 0.0; q[3] = 1.0; return; }
  double s = sin(0.5 * nn), ax[3] = {dth[0] / nn, dth[1] / nn, dth[2] / nn};
  q[0] = ax[0] * s; q[1] = ax[1] * s; q[2] = ax[2] * s; q[3] = cos(0.5 * nn);
}
static void quat_mul_norm(double *a, const double *b) { /* a <- (a*b).normalized(), xyzw */
  double ax = a[0], ay = a[1], az = a[2], aw = a[3], bx = b[0], by = b[1], bz = b[2], bw = b[3];
  double r[4] = {aw * bx + ax * bw + ay * bz - az * by, aw * by + ay * bw + az * bx - ax * bz,
                 aw * bz + az * bw + ax * by - ay * bx, aw * bw - ax * bx - ay * by - az * bz};
  double nn = sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2] + r[3] * r[3]);
  for (int i = 0; i < 4; ++i) a[i] = r[i] / nn;
}
/* core: p[3] v[3] q[4 xyzw] b_w[3] b_a[3]; p_array[3N] q_array[4N xyzw] f_array[3M] */
int xo_state_correct(double *p, double *v, double *q, double *b_w, double *b_a, double *p_array,
                     double *q_array, double *f_array, int N, int M, const double *corr) {
  for (int i = 0; i < 3; ++i) { p[i] += corr[i]; v[i] += corr[3 + i]; b_w[i] += corr[9 + i]; b_a[i] += corr[12 + i]; }
  for (int i = 0; i < 3 * N; ++i) p_array[i] += corr[K_CORE + i];
  for (int i = 0; i < 3 * M; ++i) f_array[i] += corr[K_CORE + 6 * N + i];
  double dq[4];
  error_quat(corr + 6, dq);
  quat_mul_norm(q, dq);
  for (int i = 0; i < N; ++i) {
    error_quat(corr + K_CORE + 3 * N + 3 * i, dq);
    quat_mul_norm(q_array + 4 * i, dq);
  }
  return XO_OK;
}

/* ------------------------------------------------------------------------ */
/* One pass of the IEKF loop body: constructUpdate + applyUpdate             */
/* vio_updater.cpp:267-423, updater.cpp:103-109.  Rows are linearised at the */
/* lists handed in and gated against P as it is (the prior until the last    */
/* iteration); ctot is correction_total (updater.cpp:126,140: read by the    */
/* gain step, then += correction); cov_update = is_last_iter.  P is updated  */
/* in place when cov_update; correction (n) is this pass's. M may be 0.      */
/* ------------------------------------------------------------------------ */
static int visual_update_pass(const double *C_q_G, const double *G_p_C, int n_poses, const int *trk_off,
                     const double *obs_xy, int K, const double *feat, const int *anchor_idxs,
                     const int *track_sizes, const double *z_last, int M, double *P, int n,
                     int n_poses_max, double sigma_img, double *ctot, int cov_update, double *correction,
                     int *inlier_msckf, double *gamma_msckf, int *inlier_slam, double *gamma_slam,
                     int *did_qr_out) {
  int n_obs = trk_off[K] - trk_off[0];
  int rows_m = 2 * n_obs - 3 * K, rows_s = 2 * M, rows = rows_m + rows_s;
  int rc = XO_OK, did = 0;
  double *Pc = malloc(sizeof(double) * (size_t)n * n); /* prior copy, vio_updater.cpp:284 */
  double *jm = malloc(sizeof(double) * (size_t)(rows_m > 0 ? rows_m : 1) * n);
  double *rm = malloc(sizeof(double) * (size_t)(rows_m > 0 ? rows_m : 1));
  double *cm = malloc(sizeof(double) * (size_t)(rows_m > 0 ? rows_m : 1));
  double *js = malloc(sizeof(double) * (size_t)(rows_s > 0 ? rows_s : 1) * n);
  double *rs = malloc(sizeof(double) * (size_t)(rows_s > 0 ? rows_s : 1));
  double *cs = malloc(sizeof(double) * (size_t)(rows_s > 0 ? rows_s : 1));
  double *h = malloc(sizeof(double) * (size_t)(rows > 0 ? rows : 1) * n);
  double *r = malloc(sizeof(double) * (size_t)(rows > 0 ? rows : 1));
  double *cv = malloc(sizeof(double) * (size_t)(rows > 0 ? rows : 1));
  double *hq = malloc(sizeof(double) * (size_t)n * n), *rq = malloc(sizeof(double) * n), *cq = malloc(sizeof(double) * n);
  if (!Pc || !jm || !rm || !cm || !js || !rs || !cs || !h || !r || !cv || !hq || !rq || !cq) { rc = XO_ENOMEM; goto done; }
  memcpy(Pc, P, sizeof(double) * (size_t)n * n);
  for (int i = 0; i < n; ++i) correction[i] = 0.0;
  if (K > 0) {
    rc = xo_msckf_update(C_q_G, G_p_C, n_poses, trk_off, obs_xy, K, Pc, n, n_poses_max, sigma_img, jm, rm,
                         cm, rows_m, inlier_msckf, gamma_msckf, NULL, NULL, NULL);
    if (rc != XO_OK) goto done;
  }
  if (M > 0) {
    rc = xo_slam_update(C_q_G, G_p_C, n_poses, feat, anchor_idxs, track_sizes, z_last, M, Pc, n, n_poses_max,
                        sigma_img, js, rs, cs, inlier_slam, gamma_slam, NULL);
    if (rc != XO_OK) goto done;
  }
  /* stack :406-419 */
  for (int j = 0; j < n; ++j) {
    memcpy(h + (size_t)j * rows, jm + (size_t)j * rows_m, sizeof(double) * rows_m);
    memcpy(h + rows_m + (size_t)j * rows, js + (size_t)j * rows_s, sizeof(double) * rows_s);
  }
  memcpy(r, rm, sizeof(double) * rows_m);
  memcpy(r + rows_m, rs, sizeof(double) * rows_s);
  memcpy(cv, cm, sizeof(double) * rows_m);
  memcpy(cv + rows_m, cs, sizeof(double) * rows_s);
  if (rows > 0) {
    rc = xo_qr_compress(h, rows, n, r, sigma_img, hq, rq, cq, &did);
    if (rc != XO_OK) goto done;
    if (did) rc = xo_apply_update(P, n, hq, n, rq, cq, ctot, cov_update, correction);
    else rc = xo_apply_update(P, n, h, rows, r, cv, ctot, cov_update, correction);
  }
  if (did_qr_out) *did_qr_out = did;
done:
  free(Pc); free(jm); free(rm); free(cm); free(js); free(rs); free(cs); free(h); free(r); free(cv);
  free(hq); free(rq); free(cq);
  return rc;
}

/* Footnote of the CPU baseline: the reference materialises the measurement-noise matrix DENSE,
 * r = r_diag.asDiagonal() assigned to a Matrix (vio_updater.cpp:417: rows x rows doubles, zero-filled, diagonal written), and
 * throws it away in applyQRDecomposition (:508-509).  The baseline keeps the diagonal as a vector; this times what the as-written
 * allocation + fill costs on this host for `rows` stacked rows.  Returns the trace (so that nothing is optimised away), -1 if
 * the allocation fails. */
double xo_dense_noise_footnote(int rows, const double *r_diag) {
  double *R = malloc(sizeof(double) * (size_t)rows * (size_t)rows);
  if (!R) return -1.0;
  /* the dense assignment writes every element (through a volatile function pointer: gcc would otherwise turn malloc + memset into
   * calloc and the kernel would hand out lazily-zeroed pages that are never touched) */
  void *(*volatile fill)(void *, int, size_t) = memset;
  fill(R, 0, sizeof(double) * (size_t)rows * (size_t)rows);
  for (int i = 0; i < rows; ++i) R[(size_t)i * rows + i] = r_diag ? r_diag[i] : 1.0;
  double tr = 0.0;
  for (int i = 0; i < rows; ++i) tr += R[(size_t)i * rows + i];
  free(R);
  return tr;
}

/* Full visual update with iekf_iter = 1 (updater.cpp:99-110).  This is what the CPU baseline times.
 * P is updated in place; correction (n) is written. */
int xo_visual_update(const double *C_q_G, const double *G_p_C, int n_poses, const int *trk_off,
                     const double *obs_xy, int K, const double *feat, const int *anchor_idxs,
                     const int *track_sizes, const double *z_last, int M, double *P, int n,
                     int n_poses_max, double sigma_img, double *correction, int *inlier_msckf,
                     double *gamma_msckf, int *inlier_slam, double *gamma_slam, int *did_qr_out) {
  double *ctot = calloc((size_t)n, sizeof(double));
  if (!ctot) return XO_ENOMEM;
  int rc = visual_update_pass(C_q_G, G_p_C, n_poses, trk_off, obs_xy, K, feat, anchor_idxs, track_sizes, z_last, M,
                              P, n, n_poses_max, sigma_img, ctot, 1, correction, inlier_msckf, gamma_msckf,
                              inlier_slam, gamma_slam, did_qr_out);
  free(ctot);
  return rc;
}

/* The IEKF loop of Updater::update, updater.cpp:99-110, on a whole state:
 *   for i < iekf_iter: constructUpdate(state) -- window lists re-read from the CORRECTED state
 *   (state_manager.cpp:539-584: the first n_poses slots of p_array / q_array), rows gated against the prior,
 *   which stays untouched until the last iteration -- then applyUpdate(..., correction_total, is_last_iter):
 *   corr = K (res + H ctot) - ctot (:126), covariance only when is_last_iter (:130-134), state.correct(corr)
 *   (:137), ctot += corr (:140).
 * State in/out: core p[3] v[3] q[4 xyzw] b_w[3] b_a[3]; p_array[3N] q_array[4N] f_array[3M] (N = n_poses_max).
 * correction_total (n) is what postUpdate receives; inlier flags / gammas are the LAST iteration's.
 * A pass whose stacked h is empty applies nothing (updater.cpp:106). */
int xo_visual_update_iekf(double *p, double *v, double *q, double *b_w, double *b_a, double *p_array,
                          double *q_array, double *f_array, int n_poses, const int *trk_off,
                          const double *obs_xy, int K, const int *anchor_idxs, const int *track_sizes,
                          const double *z_last, int M, double *P, int n, int n_poses_max, double sigma_img,
                          int iekf_iter, double *correction_total, int *inlier_msckf, double *gamma_msckf,
                          int *inlier_slam, double *gamma_slam) {
  double *corr = calloc((size_t)n, sizeof(double));
  if (!corr) return XO_ENOMEM;
  int rc = XO_OK;
  for (int i = 0; i < n; ++i) correction_total[i] = 0.0;          /* :82 */
  for (int it = 0; it < iekf_iter && rc == XO_OK; ++it) {
    const int last = it == iekf_iter - 1;
    int n_obs = trk_off[K] - trk_off[0];
    if (2 * n_obs - 3 * K + 2 * M <= 0) break;                     /* h.size() == 0 */
    rc = visual_update_pass(q_array, p_array, n_poses, trk_off, obs_xy, K, f_array, anchor_idxs, track_sizes,
                            z_last, M, P, n, n_poses_max, sigma_img, correction_total, last, corr, inlier_msckf,
                            gamma_msckf, inlier_slam, gamma_slam, NULL);
    if (rc != XO_OK) break;
    xo_state_correct(p, v, q, b_w, b_a, p_array, q_array, f_array, n_poses_max, (n - K_CORE - 6 * n_poses_max) / 3, corr);
  }
  free(corr);
  return rc;
}

/* ------------------------------------------------------------------------ */
/* Covariance intersection, fixed weights (ci.cpp:49-127)                    */
/* ------------------------------------------------------------------------ */
static int check_w(double w) {
  if (w > 1.0 || w == 0 || w < -1) return XO_EINVAL; /* throws, ci.cpp:59-62,98-101 */
  if (w < 0.0) return XO_EINVAL;                     /* NLopt branch: out of scope */
  return XO_OK;
}

/* S (m x m) = (1/w0) H P H^T + sum_i (1/w) H_i P_i H_i^T; w0 = 1 - k w.
 * Hs[i] is m x ns[i] (ld m), Ps[i] ns[i] x ns[i]. */
int xo_fuse_ci_msckf(const double *P, int n, const double *H, int m, int k, const double *const *Ps,
                     const int *ns, const double *const *Hs, double w_other, double *S, double *w_result) {
  if (check_w(w_other) != XO_OK) return XO_EINVAL;
  double w0 = 1.0 - (double)k * w_other;
  double *HP = malloc(sizeof(double) * (size_t)m * n);
  if (!HP) return XO_ENOMEM;
  gemm(0, 0, m, n, n, 1.0, H, m, P, n, 0.0, HP, m);
  gemm(0, 1, m, m, n, 1.0 / w0, HP, m, H, m, 0.0, S, m);
  free(HP);
  for (int i = 0; i < k; ++i) {
    double *T = malloc(sizeof(double) * (size_t)m * ns[i]);
    if (!T) return XO_ENOMEM;
    gemm(0, 0, m, ns[i], ns[i], 1.0, Hs[i], m, Ps[i], ns[i], 0.0, T, m);
    gemm(0, 1, m, m, ns[i], 1.0 / w_other, T, m, Hs[i], m, 1.0, S, m);
    free(T);
  }
  *w_result = 1.0 / w0;
  return XO_OK;
}

int xo_fuse_ci_slam(const double *Pa, int na, const double *Ha, const double *Pb, int nb, const double *Hb,
                    int m, double w_other, double *S, double *w_result) {
  if (check_w(w_other) != XO_OK) return XO_EINVAL;
  double *T = malloc(sizeof(double) * (size_t)m * (na > nb ? na : nb));
  if (!T) return XO_ENOMEM;
  gemm(0, 0, m, na, na, 1.0, Ha, m, Pa, na, 0.0, T, m);
  gemm(0, 1, m, m, na, 1.0 / (1.0 - w_other), T, m, Ha, m, 0.0, S, m);
  gemm(0, 0, m, nb, nb, 1.0, Hb, m, Pb, nb, 0.0, T, m);
  gemm(0, 1, m, m, nb, 1.0 / w_other, T, m, Hb, m, 1.0, S, m);
  free(T);
  *w_result = 1.0 / (1.0 - w_other);
  return XO_OK;
}

/* MultiSlamUpdate::processOneMatch, multi_slam_update.cpp:61-246.
 * Outputs (only meaningful if *inlier): H (3 x n, ld 3), res (3), S (3x3),
 * P_j (n x n). */
static void slam_match_side(const double *Cq, const double *Gp, const double *f, int a, int fid, int npm,
                            int ncols, double sign, double gpf[3], double *h) {
  double al = f[3 * fid], be = f[3 * fid + 1], rho = f[3 * fid + 2];
  double Ra[3][3], Sk[3][3], RS[3][3], RM[3][3];
  quat_to_rot(Cq + 4 * a, Ra);
  for (int k = 0; k < 3; ++k) gpf[k] = (1.0 / rho) * (Ra[k][0] * al + Ra[k][1] * be + Ra[k][2]) + Gp[3 * a + k];
  double v[3] = {al, be, 1.0};
  skew3(v, Sk);
  mat33_mul(Ra, Sk, RS);
  double mat[3][3] = {{1, 0, -al / rho}, {0, 1, -be / rho}, {0, 0, -1.0 / rho}};
  mat33_mul(Ra, mat, RM);
  for (int i = 0; i < 3 * ncols; ++i) h[i] = 0.0;
  int cp = K_CORE + 3 * a, ca = cp + 3 * npm, cf = K_CORE + (2 * npm + fid) * 3;
  for (int x = 0; x < 3; ++x)
    for (int y = 0; y < 3; ++y) {
      h[x + 3 * (size_t)(cp + y)] = sign * (x == y ? 1.0 : 0.0);
      h[x + 3 * (size_t)(ca + y)] = sign * (-(1.0 / rho) * RS[x][y]);
      h[x + 3 * (size_t)(cf + y)] = sign * ((1.0 / rho) * RM[x][y]);
    }
}

int xo_multi_slam_match(const double *C_q_G, const double *G_p_C, const double *feat, int anchor_idx,
                        int feature_id, const double *P, int n, int n_poses_max, const double *o_C_q_G,
                        const double *o_G_p_C, const double *o_feat, int o_anchor_idx, int o_feature_id,
                        const double *o_P, int no, int o_n_poses_max, double sigma_landmark,
                        double ci_slam_w, int *inlier, double *gamma, double *H, double *res, double *S,
                        double *P_j) {
  if (anchor_idx < 0) return XO_EINVAL;            /* throws :83-85 */
  if (feat[3 * feature_id + 2] == 0) return XO_EINVAL; /* throws :86-88 */
  double var_l = sigma_landmark * sigma_landmark;
  double *oh = malloc(sizeof(double) * 3 * (size_t)no);
  double *T = malloc(sizeof(double) * 3 * (size_t)(n > no ? n : no));
  if (!oh || !T) { free(oh); free(T); return XO_ENOMEM; }
  double gpf[3], ogpf[3];
  slam_match_side(C_q_G, G_p_C, feat, anchor_idx, feature_id, n_poses_max, n, +1.0, gpf, H);
  slam_match_side(o_C_q_G, o_G_p_C, o_feat, o_anchor_idx, o_feature_id, o_n_poses_max, no, -1.0, ogpf, oh);
  for (int k = 0; k < 3; ++k) res[k] = -gpf[k] + ogpf[k]; /* :131 */
  double S0[9], Si[9];
  gemm(0, 0, 3, n, n, 1.0, H, 3, P, n, 0.0, T, 3);
  gemm(0, 1, 3, 3, n, 1.0, T, 3, H, 3, 0.0, S0, 3);
  gemm(0, 0, 3, no, no, 1.0, oh, 3, o_P, no, 0.0, T, 3);
  gemm(0, 1, 3, 3, no, 1.0, T, 3, oh, 3, 1.0, S0, 3);
  S0[0] += var_l; S0[4] += var_l; S0[8] += var_l;
  double g = INFINITY;
  if (lu_inverse(3, S0, 3, Si) == XO_OK) {
    g = 0;
    for (int j = 0; j < 3; ++j) g += (res[0] * Si[0 + 3 * j] + res[1] * Si[1 + 3 * j] + res[2] * Si[2 + 3 * j]) * res[j];
  }
  double chi;
  xo_chi2inv(0.9, 3, &chi);
  *gamma = g;
  *inlier = g < chi;
  int rc = XO_OK;
  if (*inlier) {
    double w_res;
    rc = xo_fuse_ci_slam(P, n, H, o_P, no, oh, 3, ci_slam_w, S, &w_res);
    if (rc == XO_OK) {
      S[0] += var_l; S[4] += var_l; S[8] += var_l;
      memcpy(P_j, P, sizeof(double) * (size_t)n * n);
      int cols[3] = {K_CORE + 3 * anchor_idx, K_CORE + 3 * anchor_idx + 3 * n_poses_max,
                     K_CORE + (2 * n_poses_max + feature_id) * 3};
      for (int b = 0; b < 3; ++b)
        for (int x = 0; x < 3; ++x)
          for (int y = 0; y < 3; ++y) P_j[(cols[b] + x) + (size_t)(cols[b] + y) * n] *= w_res; /* Q7 */
    }
  }
  free(oh); free(T);
  return rc;
}

/* MSCKF-MSCKF CI block of preProcessOneTrack, msckf_update.cpp:96-279, for
 * ONE track with k matched agents (ground-truth association).
 * m_* arrays describe the matched agents; their tracks are concatenated
 * FIRST for triangulation, self last (:113-149).
 * Outputs: self_inlier/self_gamma (own single-agent gate), has_ci, and if
 * has_ci: H (3k x n, ld 3k), res (3k), S (3k x 3k), P_j (n x n). */
int xo_msckf_ci_track(const double *obs, int L, const double *C_q_G, const double *G_p_C, int n_poses,
                      const double *P, int n, int n_poses_max, double sigma_img, int k,
                      const double *const *m_obs, const int *m_L, const double *const *m_q,
                      const double *const *m_p, const int *m_nposes, const double *const *m_P,
                      const int *m_n, double ci_msckf_w, int *self_inlier, double *self_gamma,
                      int *has_ci, double *ci_gamma, double *H, double *res, double *S, double *P_j,
                      double *gpf_out) {
  double var_img = sigma_img * sigma_img;
  int Ltot = L;
  for (int i = 0; i < k; ++i) Ltot += m_L[i];
  double *qa = malloc(sizeof(double) * 4 * (size_t)Ltot), *pa = malloc(sizeof(double) * 3 * (size_t)Ltot);
  double *oa = malloc(sizeof(double) * 2 * (size_t)Ltot);
  if (!qa || !pa || !oa) { free(qa); free(pa); free(oa); return XO_ENOMEM; }
  int at = 0;
  for (int i = 0; i < k; ++i) {
    memcpy(qa + 4 * at, m_q[i] + 4 * (size_t)(m_nposes[i] - m_L[i]), sizeof(double) * 4 * m_L[i]);
    memcpy(pa + 3 * at, m_p[i] + 3 * (size_t)(m_nposes[i] - m_L[i]), sizeof(double) * 3 * m_L[i]);
    memcpy(oa + 2 * at, m_obs[i], sizeof(double) * 2 * m_L[i]);
    at += m_L[i];
  }
  memcpy(qa + 4 * at, C_q_G + 4 * (size_t)(n_poses - L), sizeof(double) * 4 * L);
  memcpy(pa + 3 * at, G_p_C + 3 * (size_t)(n_poses - L), sizeof(double) * 3 * L);
  memcpy(oa + 2 * at, obs, sizeof(double) * 2 * L);
  double ivd[3], gpf[3];
  int rc = xo_triangulate_gn(qa, pa, oa, Ltot, 10, 1e-5, ivd, NULL);
  if (rc != XO_OK) { ivd[0] = ivd[1] = ivd[2] = NAN; }
  global_feature_position(ivd, qa + 4 * (Ltot - 1), pa + 3 * (Ltot - 1), gpf);
  if (gpf_out) { gpf_out[0] = gpf[0]; gpf_out[1] = gpf[1]; gpf_out[2] = gpf[2]; }
  free(qa); free(pa); free(oa);
  *has_ci = 0;
  track_out_t own;
  rc = process_one_track(obs, L, P, n, C_q_G, G_p_C, n_poses, n_poses_max, var_img, gpf, 1, &own);
  if (rc != XO_OK) return rc;
  *self_inlier = own.valid && own.inlier;
  *self_gamma = own.gamma;
  if (!(*self_inlier) || k == 0) { track_out_free(&own); return XO_OK; }
  int mr = 3 * (k + 1), ncols = n;
  for (int i = 0; i < k; ++i) ncols += m_n[i];
  double *jx = calloc((size_t)mr * ncols, sizeof(double)), *jf = calloc((size_t)mr * 3, sizeof(double));
  double *rp = calloc((size_t)mr, sizeof(double));
  for (int j = 0; j < n; ++j)
    for (int x = 0; x < 3; ++x) jx[x + (size_t)j * mr] = own.up_jac[x + 3 * (size_t)j];
  for (int y = 0; y < 3; ++y)
    for (int x = 0; x < 3; ++x) jf[x + (size_t)y * mr] = own.up_hf[x + 3 * y];
  for (int x = 0; x < 3; ++x) rp[x] = own.up_res[x];
  track_out_free(&own);
  int col = n;
  for (int i = 0; i < k; ++i) {
    track_out_t o;
    rc = process_one_track(m_obs[i], m_L[i], m_P[i], m_n[i], m_q[i], m_p[i], m_nposes[i], m_nposes[i], var_img,
                           gpf, 1, &o);
    if (rc != XO_OK) { free(jx); free(jf); free(rp); return rc; }
    if (o.valid) {
      int r0 = 3 * (i + 1);
      for (int j = 0; j < m_n[i]; ++j)
        for (int x = 0; x < 3; ++x) jx[r0 + x + (size_t)(col + j) * mr] = o.up_jac[x + 3 * (size_t)j];
      for (int y = 0; y < 3; ++y)
        for (int x = 0; x < 3; ++x) jf[r0 + x + (size_t)y * mr] = o.up_hf[x + 3 * y];
      for (int x = 0; x < 3; ++x) rp[r0 + x] = o.up_res[x];
    }
    col += m_n[i];
    track_out_free(&o);
  }
  /* nullSpaceProjection :494-501 */
  double tau[3];
  double *Q = malloc(sizeof(double) * (size_t)mr * mr);
  qr_unblocked(jf, mr, 3, mr, tau);
  form_q(jf, mr, 3, mr, tau, Q);
  const double *A = Q + (size_t)3 * mr;
  int m = mr - 3; /* = 3k */
  double *hx = malloc(sizeof(double) * (size_t)m * ncols);
  gemm(1, 0, m, ncols, mr, 1.0, A, mr, jx, mr, 0.0, hx, m);
  gemm(1, 0, m, 1, mr, 1.0, A, mr, rp, mr, 0.0, res, m);
  memcpy(H, hx, sizeof(double) * (size_t)m * n);
  /* S_j :217-237 */
  double *T = malloc(sizeof(double) * (size_t)m * ncols);
  gemm(0, 0, m, n, n, 1.0, H, m, P, n, 0.0, T, m);
  gemm(0, 1, m, m, n, 1.0, T, m, H, m, 0.0, S, m);
  const double **Hs = malloc(sizeof(double *) * (size_t)k);
  col = n;
  for (int i = 0; i < k; ++i) {
    Hs[i] = hx + (size_t)col * m;
    gemm(0, 0, m, m_n[i], m_n[i], 1.0, Hs[i], m, m_P[i], m_n[i], 0.0, T, m);
    gemm(0, 1, m, m, m_n[i], 1.0, T, m, Hs[i], m, 1.0, S, m);
    col += m_n[i];
  }
  for (int i = 0; i < m; ++i) S[i + (size_t)i * m] += var_img;
  double *Sc = malloc(sizeof(double) * (size_t)m * m), *Si = malloc(sizeof(double) * (size_t)m * m);
  memcpy(Sc, S, sizeof(double) * (size_t)m * m);
  double g = INFINITY;
  if (lu_inverse(m, Sc, m, Si) == XO_OK) {
    g = 0;
    for (int j = 0; j < m; ++j) {
      double s = 0;
      for (int i = 0; i < m; ++i) s += res[i] * Si[i + (size_t)j * m];
      g += s * res[j];
    }
  }
  double chi;
  xo_chi2inv(0.95, 2 * Ltot - 3, &chi); /* :245-247 */
  if (ci_gamma) *ci_gamma = g;
  if (g < chi) {
    double w_res;
    rc = xo_fuse_ci_msckf(P, n, H, m, k, m_P, m_n, Hs, ci_msckf_w, S, &w_res);
    if (rc == XO_OK) {
      for (int i = 0; i < m; ++i) S[i + (size_t)i * m] += var_img; /* :255, Q8 */
      memcpy(P_j, P, sizeof(double) * (size_t)n * n);
      for (int i = 0; i < L; ++i) { /* :258-267, Q7 */
        int pos = n_poses - L + i;
        int c2[2] = {K_CORE + 3 * pos, K_CORE + 3 * pos + 3 * n_poses_max};
        for (int b = 0; b < 2; ++b)
          for (int x = 0; x < 3; ++x)
            for (int y = 0; y < 3; ++y) P_j[(c2[b] + x) + (size_t)(c2[b] + y) * n] *= w_res;
      }
      *has_ci = 1;
    }
  }
  free(Hs); free(Sc); free(Si); free(T); free(hx); free(Q); free(jx); free(jf); free(rp);
  return rc;
}

/* achieved-rate probe used by bench.py to report the baseline's dgemm GF/s */
double xo_gemm_probe(int m, int n, int k, int reps) {
  double *A = malloc(sizeof(double) * (size_t)m * k), *B = malloc(sizeof(double) * (size_t)k * n);
  double *C = malloc(sizeof(double) * (size_t)m * n);
  for (size_t i = 0; i < (size_t)m * k; ++i) A[i] = 1.0 / (double)(1 + i % 7);
  for (size_t i = 0; i < (size_t)k * n; ++i) B[i] = 1.0 / (double)(1 + i % 5);
  double s = 0;
  for (int r = 0; r < reps; ++r) { gemm(0, 0, m, n, k, 1.0, A, m, B, k, 0.0, C, m); s += C[(size_t)r % ((size_t)m * n)]; }
  free(A); free(B); free(C);
  return s;
}
