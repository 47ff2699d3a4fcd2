// xk_linalg.hip.h -- Householder CAQR, fp64-MFMA GEMM and blocked Cholesky
// kernels for the EKF update (gfx950).
//
//   CAQR            replaces VioUpdater::applyQRDecomposition
//                   (src/x/vio/vio_updater.cpp:487-512)
//   GEMM / Cholesky replace the Eigen products and S.inverse() of
//                   Updater::applyUpdate / applyCI (src/x/ekf/updater.cpp:117-161)
#pragma once
#include <hip/hip_runtime.h>

#include "xk_chol16.hip.h"

// Sum over the SPLIT (2, 4, 8 or 16) adjacent lanes that share a column, with DPP moves (no LDS round
// trip, unlike ds_bpermute-based shuffles).
template <int CTRL>
__device__ __forceinline__ double xk_dpp_quad(double x) {
  const long long q = __builtin_bit_cast(long long, x);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)(q & 0xffffffffLL), CTRL, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(q >> 32), CTRL, 0xf, 0xf, true);
  return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}
template <int SPLIT>
__device__ __forceinline__ double xk_group_sum(double x) {
  x += xk_dpp_quad<0xB1>(x);                  // quad_perm [1,0,3,2]: lane ^ 1
  if (SPLIT >= 4) x += xk_dpp_quad<0x4E>(x);  // quad_perm [2,3,0,1]: lane ^ 2
  if (SPLIT >= 8) x += xk_dpp_quad<0x141>(x); // row_half_mirror: the other quad of the 8-lane group
  if (SPLIT == 16) x += xk_dpp_quad<0x140>(x); // row_mirror: the other half of the 16-lane row
  return x;
}

// ----------------------------------------------------------------------------
// Generic strided fp64 GEMM on v_mfma_f64_16x16x4_f64, one single-wave workgroup per 16x16
// output tile.  n <= ~350 here, so every operand is
// L2-resident; operands are read straight into the MFMA fragment layout
// (A[i = lane&15][k = lane>>4], B[k = lane>>4][j = lane&15]; C/D: col =
// lane&15, row = (lane>>4) + 4*reg -- the f64 map, not the f32 one).
//
//   mode 0: C = alpha*A*B + beta*D
//   mode 1: as 0, plus diag[i] (or diag_scalar) added on the diagonal
//   mode 2: C = 0.5*((D + alpha*AB) + (D + alpha*AB)^T)   (A*B symmetric)
// xcol: column N-1 is a vector product with its own operands and destination (see XkGemmArgs)
// ----------------------------------------------------------------------------
typedef double xk_d4 __attribute__((ext_vector_type(4)));

struct XkGemmArgs {
  const double *A, *B, *D;
  double *C;
  int M, N, K;
  long sar, sac, sbr, sbc, sdr, sdc, scr, scc;
  double alpha, beta;
  int mode;
  const double *diag;
  double diag_scalar;
  // optional extra column (index N-1) riding along so that a matrix-vector product does not need its own
  // launch:  cx[i] = sum_k A[i][k] bx[k] + dx[i] - ex[i]   (bx / dx / ex may be null = 0)
  int xcol;
  const double *bx, *dx, *ex;
  long sbx, sdx;
  double *cx;
  long scx;
  // optional completion marker: the LAST workgroup to finish (counted in *done_cnt, device memory, left at zero) writes
  // done_seq into *done_flag (pinned host memory) -- a host that polls it sees the launch's host-visible results without
  // waiting for the runtime's completion signal
  unsigned *done_cnt;
  unsigned long long *done_flag;
  unsigned long long done_seq;
  // structure the caller vouches for (all zero = a general product):
  //   tri_a: A[i][k] == 0 for k < i (the compressed measurement matrix, upper trapezoidal) -- row tile tm starts at k = 16 tm
  //   tri_b: B[k][j] == 0 for k < j (its transpose)                                        -- column tile tn starts at k = 16 tn
  //   sym_cols > 0: columns [0, sym_cols) of the output are a symmetric matrix of which only the tiles on and above the
  //     diagonal are wanted (S: the Cholesky reads the upper triangle) -- or, in mode 2, of which the tiles below the diagonal
  //     are written as the mirror image of the ones above (the same bits: the two operands of a mirrored entry are swapped sums)
  int tri_a, tri_b, sym_cols;
};

#ifndef XK_GEMM_WAVES
#define XK_GEMM_WAVES 4
#endif
// workgroups of a launch
static inline int xk_gemm_grid(const XkGemmArgs &g) {
  const int tiles_m = (g.M + 15) >> 4, tiles_n = (g.N + 15) >> 4, ts = g.sym_cols >> 4;
  int n = 0;
  for (int tm = 0; tm < tiles_m; ++tm) n += tiles_n - (g.sym_cols > 0 ? (tm < ts ? tm : ts) : 0);
  return n;
}
__global__ __launch_bounds__(64 * XK_GEMM_WAVES) void xk_gemm_f64(XkGemmArgs g) {
  // One 16 x 16 output tile per workgroup, K split over XK_GEMM_WAVES waves (these GEMMs are a few MFLOP each and pure
  // latency: with one wave per tile the 45 dependent MFMA steps of K = 180 and their three rounds of operand loads were
  // most of a 8 us kernel); the partial tiles meet in LDS and wave 0 adds them in a fixed order.
  __shared__ double red[XK_GEMM_WAVES - 1][4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int tiles_n = (g.N + 15) >> 4;
  // tile of this workgroup: row-major over the tiles that are wanted -- with sym_cols > 0 the tiles below the diagonal of the
  // symmetric part (tn < min(tm, sym_cols / 16)) are not in the grid at all (xk_gemm_grid)
  int tm, tn;
  if (g.sym_cols > 0) {
    const int ts = g.sym_cols >> 4;
    int rest = blockIdx.x;
    tm = 0;
    for (;; ++tm) {
      const int live = tiles_n - min(tm, ts);
      if (rest < live) break;
      rest -= live;
    }
    tn = min(tm, ts) + rest;
  } else {
    tm = (int)blockIdx.x / tiles_n;
    tn = (int)blockIdx.x - tm * tiles_n;
  }
  const int li = lane & 15, lk = lane >> 4;
  const int arow = tm * 16 + li, bcol = tn * 16 + li;
  const bool xc = g.xcol && bcol == g.N - 1;                    // this lane feeds the extra column
  const bool aok = arow < g.M, bok = xc ? g.bx != nullptr : bcol < g.N;
  const double *ap = g.A + (long)arow * g.sar, *bp = xc ? g.bx : g.B + (long)bcol * g.sbc;
  const long sbr = xc ? g.sbx : g.sbr;
  xk_d4 acc = {0.0, 0.0, 0.0, 0.0};
  const int k0t = min(g.K, max(g.tri_a ? 16 * tm : 0, g.tri_b ? 16 * tn : 0));   // the operands are zero before k0t
  const int kq = ((g.K - k0t + 4 * XK_GEMM_WAVES - 1) / (4 * XK_GEMM_WAVES)) * 4;   // K range of a wave, a multiple of the MFMA depth
  const int kbeg = k0t + wave * kq, kend = min(g.K, kbeg + kq);
  // chunks of 64 with the NEXT chunk's 32 loads in flight while this one feeds the matrix core
  constexpr int CH = 16;   // MFMA steps per chunk
  double av[2][CH], bv[2][CH];
  auto load = [&](int buf, int k0) {
#pragma unroll
    for (int u = 0; u < CH; ++u) {
      const int k = k0 + 4 * u + lk;
      av[buf][u] = (aok && k < kend) ? ap[(long)k * g.sac] : 0.0;
      bv[buf][u] = (bok && k < kend) ? bp[(long)k * sbr] : 0.0;
    }
  };
  if (kbeg < kend) {
    load(0, kbeg);
    for (int k0 = kbeg; k0 < kend; k0 += 8 * CH) {
      if (k0 + 4 * CH < kend) load(1, k0 + 4 * CH);
#pragma unroll
      for (int u = 0; u < CH; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[0][u], bv[0][u], acc, 0, 0, 0);
      if (k0 + 4 * CH >= kend) break;
      if (k0 + 8 * CH < kend) load(0, k0 + 8 * CH);
#pragma unroll
      for (int u = 0; u < CH; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[1][u], bv[1][u], acc, 0, 0, 0);
    }
  }
  if (wave > 0) {
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wave - 1][r][lane] = acc[r];
  }
  __syncthreads();
  if (wave > 0) return;
#pragma unroll
  for (int w = 0; w < XK_GEMM_WAVES - 1; ++w)
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] += red[w][r][lane];
  const int col = tn * 16 + li;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = tm * 16 + lk + 4 * r;
    if (row >= g.M || col >= g.N) continue;
    if (xc) {
      g.cx[(long)row * g.scx] = acc[r] + (g.dx ? g.dx[(long)row * g.sdx] : 0.0) - (g.ex ? g.ex[row] : 0.0);
      continue;
    }
    double v = g.alpha * acc[r];
    if (g.mode == 2) {
      const double d1 = g.D[(long)row * g.sdr + (long)col * g.sdc];
      const double d2 = g.D[(long)col * g.sdr + (long)row * g.sdc];
      v = 0.5 * ((d1 + v) + (d2 + v));
    } else {
      if (g.beta != 0.0) v += g.beta * g.D[(long)row * g.sdr + (long)col * g.sdc];
      if (g.mode == 1 && row == col) v += g.diag ? g.diag[row] : g.diag_scalar;
    }
    g.C[(long)row * g.scr + (long)col * g.scc] = v;
    if (g.mode == 2 && g.sym_cols > 0 && tn > tm && col < g.sym_cols && col < g.M) g.C[(long)col * g.scr + (long)row * g.scc] = v;
  }
  if (g.done_flag) {
    if (g.xcol && tn == tiles_n - 1) __threadfence_system();   // the extra column may be host memory: its stores first
    if (lane == 0 && atomicAdd(g.done_cnt, 1u) == gridDim.x - 1) {
      *g.done_cnt = 0;
      __hip_atomic_store(g.done_flag, g.done_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// ----------------------------------------------------------------------------
// Blocked Cholesky of the augmented row-major matrix  Maug = [S | W | z]
// (c x (c + nrhs)), ONE launch per block step kb (xk_chol_step):
//   X[kb:kb+nb, kb+nb:]     = L_kk^-1 * Maug[kb:kb+nb, kb+nb:]
//   Maug[kb+nb:, kb+nb:]   -= X[kb:kb+nb, S cols]^T * X[kb:kb+nb, kb+nb:]
// After the last step X[:, c:] = L^-1 [W | z].
//
// A block step used to be three dependent launches (diagonal factor, row scaling, trailing update)
// of 4-15 us each for a few MFLOP.  Now every workgroup is ONE wave that owns one 16 x 16 tile of
// the step's output and factors the 32 x 32 diagonal block itself: the factorisation is redundant
// across workgroups, but it is 32 dependent steps whoever runs it, and this way nothing waits for
// a launch.  Lane t keeps row t of the block AND row t of L^-1 in registers (static indexing, fully
// unrolled); column k of L and row k of L^-1 travel through LDS as wave-wide broadcasts.  The tile
// products run on the matrix cores: X_i = L^-1 B_i lands in the MFMA C/D layout, whose registers
// are exactly the A / B operands of X_i^T X_j, so no fragment moves between lanes.
// ----------------------------------------------------------------------------
#define XK_CHOL_NB 32

struct XkCholStepArgs {
  double *Maug;      // [c][ld] row-major
  int ld, kb, nb;    // diagonal block rows/cols [kb, kb+nb)
  int c, ncols;      // rows of Maug, columns in use (c + n + 1)
  double *X;         // same shape / ld as Maug
  int ncb;           // column tiles of the trailing range [kb+nb, ncols)
  int *status;       // set to 2 (XK_ESINGULAR) if a pivot is not positive
#ifdef XK_CHOL_PROBE
  long long *dbg;    // clock64 stamps of the last workgroup
#endif
};

#ifdef XK_CHOL_PROBE
#define XK_CSTAMP(i) do { if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) a.dbg[i] = clock64(); } while (0)
#else
#define XK_CSTAMP(i)
#endif

__global__ __launch_bounds__(64) void xk_chol_step(XkCholStepArgs a) {
  constexpr int B = XK_CHOL_NB, LDL = B + 1;
  // comb[k][j]: j <= k -> row k of L^-1 (unscaled), j > k -> L(j,k): the one broadcast row of step k
  __shared__ __attribute__((aligned(16))) double comb[B][B + 2];
  __shared__ double Ls[B * LDL];                                 // L^-1, row-major, for the MFMA gathers
  const int t = threadIdx.x, nb = a.nb, kb = a.kb;
  const int rb = blockIdx.x / a.ncb, cb = blockIdx.x % a.ncb;    // rb = 0: X tile; rb >= 1: trailing row block rb-1
  const int c0 = kb + nb;                                        // first trailing row / column
  if (rb >= 1 && 16 * cb + 15 < 16 * (rb - 1)) return;           // below the diagonal of S: never read
  XK_CSTAMP(0);
  // ---- operands of the tile products: independent of the factorisation, so their HBM latency is
  // hidden behind it.  MFMA 16x16x4 layouts: A/B operand lane l <-> (index l&15, k = l>>4).
  const int li = t & 15, lk = t >> 4;
  const int colj = c0 + 16 * cb + li;            // this lane's column of the tile
  const int rowi = c0 + 16 * (rb > 0 ? rb - 1 : 0) + li;   // as a COLUMN of Maug it carries X_i
  const bool cj_ok = colj < a.ncols, ci_ok = rb > 0 && rowi < a.c;
  double bj[B / 4], bi[B / 4], cold[4];
#pragma unroll
  for (int q = 0; q < B / 4; ++q) {
    const int k = 4 * q + lk;
    bj[q] = (cj_ok && k < nb) ? a.Maug[(size_t)(kb + k) * a.ld + colj] : 0.0;
    bi[q] = (ci_ok && k < nb) ? a.Maug[(size_t)(kb + k) * a.ld + rowi] : 0.0;
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int i = c0 + 16 * (rb - 1) + lk + 4 * r;
    cold[r] = (rb > 0 && cj_ok && i < a.c) ? a.Maug[(size_t)i * a.ld + colj] : 0.0;
  }
  // ---- factor the diagonal block and invert the factor.  Lanes 0..31: lane t keeps row t of the
  // block (Cholesky, row-oriented); lanes 32..63: lane 32+t keeps row t of L^-1, built by applying the
  // same eliminations to the identity.  Both halves run the SAME 32 FMAs per step on v[] against the
  // broadcast row comb[k][]: positions j > k carry column k of L (used by the Cholesky half),
  // positions j <= k carry row k of L^-1 (used by the inverse half).
  const bool inv_half = t >= B;
  const int tt = t & (B - 1);
  double v[B];
  // the upper triangle is current (row-oriented updates): A(t,j) = M(min,max); identity padding past nb
#pragma unroll
  for (int j = 0; j < B; ++j) {
    const int r = tt < j ? tt : j, cc = tt < j ? j : tt;
    const double av = (!inv_half && tt < nb && j < nb) ? a.Maug[(size_t)(kb + r) * a.ld + kb + cc] : ((tt == j) ? 1.0 : 0.0);
    v[j] = av;
  }
  XK_CSTAMP(1);
  bool bad = false;
  // Row k of L^-1 is final, up to the factor 1/L(k,k), as soon as step k-1 has updated it: its lane
  // publishes it UNSCALED at the end of step k-1, off the pivot -> column -> update chain.
  if (t == B) comb[0][0] = 1.0;
#pragma unroll
  for (int k = 0; k < B; ++k) {
    // pivot from lane k (uniform); 1/sqrt from the hardware seed + two Newton steps
    const long long pq = __builtin_bit_cast(long long, v[k]);
    const int plo = __builtin_amdgcn_readlane((int)(pq & 0xffffffffLL), k), phi = __builtin_amdgcn_readlane((int)(pq >> 32), k);
    const double piv = __builtin_bit_cast(double, ((long long)phi << 32) | (unsigned int)plo);
    if (!(piv > 0.0)) bad = true;
    double inv = __builtin_amdgcn_rsq(piv);
    inv = inv * fma(-0.5 * piv * inv, inv, 1.5);
    inv = inv * fma(-0.5 * piv * inv, inv, 1.5);
    const double lik = v[k] * inv;     // Cholesky half: L(t,k) for t >= k
    if (!inv_half && t > k) comb[k][t] = lik;
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
    __builtin_amdgcn_wave_barrier();
    // Cholesky half:  A(t,j) -= L(t,k) L(j,k)           for j > k  (entries j <= k are dead)
    // inverse half:   W(t,j) -= L(t,k)/L(k,k) Wraw(k,j)  for j <= k, t > k;  W(k,:) = Wraw(k,:)/L(k,k)
    // (the owner of row k scales it in the same FMA: bc[j] IS its own v[j], and v - (1 - 1/l) v = v / l)
    const double ltk = !inv_half ? 0.0 : (tt > k) ? comb[k][tt] * inv : (tt == k) ? 1.0 - inv : 0.0;
    const double m_hi = inv_half ? 0.0 : lik;
    xk_d2 bc[B / 2];
#pragma unroll
    for (int jj = 0; jj < B / 2; ++jj) bc[jj] = *reinterpret_cast<const xk_d2 *>(&comb[k][2 * jj]);
#pragma unroll
    for (int j = 0; j < B; ++j) v[j] = fma((j <= k) ? -ltk : -m_hi, bc[j >> 1][j & 1], v[j]);
    if (k + 1 < B && t == B + k + 1) {
#pragma unroll
      for (int j = 0; j <= k + 1; ++j) comb[k + 1][j] = v[j];
    }
    // pin this step's results here: the optimiser otherwise sinks the FMAs towards their first use
    // (the pivot of step j), keeps every step's broadcast operands live meanwhile, and spills
#pragma unroll
    for (int j = 0; j < B; ++j) asm volatile("" : "+v"(v[j]));
  }
  XK_CSTAMP(2);
  if (bad) {
    if (blockIdx.x == 0 && t == 0) *a.status = 2;
    return;
  }
  if (inv_half) {
#pragma unroll
    for (int j = 0; j < B; ++j) Ls[tt * LDL + j] = (j <= tt) ? v[j] : 0.0;
  }
  __builtin_amdgcn_s_waitcnt(0xc07f);
  __builtin_amdgcn_wave_barrier();
  XK_CSTAMP(3);
  // ---- X_j = L^-1 * B_j for the column tile (and X_i for the row block), on the matrix cores
  typedef double xk_d4 __attribute__((ext_vector_type(4)));
  xk_d4 xj[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}}, xi[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
#pragma unroll
  for (int q = 0; q < B / 4; ++q) {
#pragma unroll
    for (int tm = 0; tm < 2; ++tm) {
      if (4 * q >= 16 * tm + 16) continue;       // L^-1 is lower triangular: rows 16tm.. only see k < 16tm+16
      const double lv = Ls[(16 * tm + li) * LDL + 4 * q + lk];
      xj[tm] = __builtin_amdgcn_mfma_f64_16x16x4f64(lv, bj[q], xj[tm], 0, 0, 0);
      if (rb > 0) xi[tm] = __builtin_amdgcn_mfma_f64_16x16x4f64(lv, bi[q], xi[tm], 0, 0, 0);
    }
  }
  XK_CSTAMP(4);
  if (rb == 0) {
    if (!cj_ok) return;
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int k = 16 * tm + lk + 4 * r;
        if (k < nb) a.X[(size_t)(kb + k) * a.ld + colj] = xj[tm][r];
      }
    return;
  }
  // C(i,j) -= sum_k X(k,i) X(k,j): register r of row-tile tm is K-slice 16tm + 4r of both operands
  xk_d4 acc = {0, 0, 0, 0};
#pragma unroll
  for (int tm = 0; tm < 2; ++tm)
#pragma unroll
    for (int r = 0; r < 4; ++r) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(xi[tm][r], xj[tm][r], acc, 0, 0, 0);
  if (!cj_ok) return;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int i = c0 + 16 * (rb - 1) + lk + 4 * r;
    if (i < a.c) a.Maug[(size_t)i * a.ld + colj] = cold[r] - acc[r];
  }
  XK_CSTAMP(5);
}

// ----------------------------------------------------------------------------
// The same factorisation for c <= 192 in ONE launch (xk_chol_whole): a block step is 16 pivots deep
// whoever runs it, so six launches of xk_chol_step are mostly six launch latencies.  Here a workgroup
// keeps the WHOLE upper triangle of S in registers -- 16 x 16 tiles in the MFMA C/D layout, dealt to 9
// worker waves -- plus one 16-column chunk of the right-hand sides [W | z]; there is one workgroup per
// chunk, each factoring S redundantly (same reasoning as above: the pivot chain is sequential anyway
// and the other 240 CUs are idle).  One more wave does nothing but the pivot chains.  Block step j:
//   F  (factor wave) tile (j,j) -> L_jj^-1                      dbuf -> Ls      | workers: C of step j-1
//   -- barrier 1 --
//   B  (workers) row j:  X_jk = L_jj^-1 S_jk  for their tiles k > j and the chunk tile -> Xs, X rows (HBM);
//      the wave that owns (j,j+1) also owns (j+1,j+1) and updates it from its own registers -> dbuf
//   -- barrier 2 --
//   C  (workers) S_ik -= X_ji^T X_jk for j < i <= k and the chunk tiles i > j: operands straight from Xs,
//      because a C/D-layout register is both the A operand of X^T X and its B operand.
// The critical path per step is F + B (~4000 clocks); C hides behind the next F.
// The tile -> (wave, slot) table comes from the host: slots 0,1 = the wave's diagonal tiles, slots 2,3 = the
// tiles above them, slots 4..10 = the rest round-robin; slots 11,12 (chunk tiles) are filled in here.
// ----------------------------------------------------------------------------
#define XK_CHOLW_MAXB 12    // 16-row blocks: c <= 192
#define XK_CHOLW_NW 9       // worker waves (waves 0,1,2, 4,5,6, 8,9,10 of 11: SIMD 3 is left to the factor wave)
#define XK_CHOLW_WAVES 11
#define XK_CHOLW_NS 11      // S tiles per worker wave

struct XkCholWholeTab {
  unsigned char i[XK_CHOLW_NW][XK_CHOLW_NS], k[XK_CHOLW_NW][XK_CHOLW_NS];   // 255 = empty slot
};
// host side: the slot table for nbk row blocks
static inline void xk_cholw_table(int nbk, XkCholWholeTab &t) {
  for (int w = 0; w < XK_CHOLW_NW; ++w)
    for (int s = 0; s < XK_CHOLW_NS; ++s) t.i[w][s] = t.k[w][s] = 255;
  for (int d = 0; d < nbk; ++d) {
    const int w = d % XK_CHOLW_NW, s = d / XK_CHOLW_NW;       // s = 0, 1 (nbk <= 16)
    t.i[w][s] = t.k[w][s] = (unsigned char)d;
    if (d > 0) { t.i[w][2 + s] = (unsigned char)(d - 1); t.k[w][2 + s] = (unsigned char)d; }
  }
  int fill[XK_CHOLW_NW];
  for (int w = 0; w < XK_CHOLW_NW; ++w) fill[w] = 4;
  int w = 0;
  for (int i = 0; i < nbk; ++i)
    for (int k = i + 2; k < nbk; ++k) {
      while (fill[w] >= XK_CHOLW_NS) w = (w + 1) % XK_CHOLW_NW;
      t.i[w][fill[w]] = (unsigned char)i; t.k[w][fill[w]] = (unsigned char)k;
      ++fill[w];
      w = (w + 1) % XK_CHOLW_NW;
    }
}

struct XkCholWholeArgs {
  const double *Maug;   // [c][ld] row-major  [S | W | z]
  int ld, c, ncols;     // ncols = c + n + 1
  double *X;            // [c][ld]: X[:, c:ncols] = L^-1 [W | z]   (the S columns are not written)
  int *status;          // set to 2 (XK_ESINGULAR) if a pivot is not positive
  XkCholWholeTab tab;
#ifdef XK_CHOLW_PROBE
  long long *dbg;       // [wave][8] clock64 stamps of block step XK_CHOLW_PROBE, workgroup 0
#endif
};
#ifdef XK_CHOLW_PROBE
#define XK_WSTAMP(i) do { if (blockIdx.x == 0 && lane == 0 && j == XK_CHOLW_PROBE) a.dbg[wave * 8 + (i)] = clock64(); } while (0)
#else
#define XK_WSTAMP(i)
#endif

__global__ __launch_bounds__(64 * XK_CHOLW_WAVES) void xk_chol_whole(XkCholWholeArgs a) {
  constexpr int NW = XK_CHOLW_NW, NS = XK_CHOLW_NS, NT = NS + 2, XR = XK_CHOLW_MAXB;
  __shared__ __attribute__((aligned(16))) double Xs[2][XK_CHOLW_MAXB + 1][256];   // X_j tiles (slot 12: the chunk)
  __shared__ __attribute__((aligned(16))) double Ls[2][16 * 17];   // L_jj^-1, padded rows (MFMA A-operand gathers)
  __shared__ __attribute__((aligned(16))) double dbuf[256];        // tile (j,j), row-major
  // waves go to the four SIMDs of the CU round-robin: the factor wave (3) has SIMD 3 to itself -- its FP64
  // multiply-adds would otherwise queue behind the workers' FP64 MFMAs, which use the same pipe
  const int lane = threadIdx.x & 63, hw = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  if (hw == 7) return;
  const int wave = (hw & 3) == 3 ? NW : (hw >> 2) * 3 + (hw & 3);
  const int nbk = (a.c + 15) / 16;
  if (wave == NW) {
    // ---- the factor wave
    bool bad = false;
    __syncthreads();
    for (int j = 0; j < nbk; ++j) {
      XK_WSTAMP(0);
      if (xk_chol16_bcast(dbuf, Ls[j & 1], lane)) bad = true;
      XK_WSTAMP(1);
      __syncthreads();                                             // barrier 1
      if (j + 1 == nbk) break;
      __syncthreads();                                             // barrier 2
    }
    if (bad && lane == 0) *a.status = 2;
    return;
  }
  // ---- worker waves
  const int li = lane & 15, lk = lane >> 4;
  int si[NT], sk[NT];
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const int i = a.tab.i[wave][s], k = a.tab.k[wave][s];
    si[s] = (i == 255) ? -1 : i;
    sk[s] = (i == 255) ? -1 : k;
  }
#pragma unroll
  for (int u = 0; u < 2; ++u) {                                    // chunk tiles: row blocks wave, wave + NW
    si[NS + u] = (wave + NW * u < nbk) ? wave + NW * u : -1;
    sk[NS + u] = XR;
  }
  const int rcol = a.c + 16 * (int)blockIdx.x + li;                // the chunk's column of Maug / X
  const bool rc_ok = rcol < a.ncols;
  xk_d4 T[NT];
#pragma unroll
  for (int s = 0; s < NT; ++s) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 16 * si[s] + lk + 4 * r, col = (s < NS) ? 16 * sk[s] + li : rcol;
      double v = 0.0;
      if (si[s] >= 0) {
        if (s < NS) v = (row < a.c && col < a.c) ? a.Maug[(size_t)row * a.ld + col] : (row == col ? 1.0 : 0.0);
        else v = (rc_ok && row < a.c) ? a.Maug[(size_t)row * a.ld + col] : 0.0;
      }
      T[s][r] = v;
    }
  }
  if (wave == 0) {                                                 // tile (0,0) = wave 0, slot 0
#pragma unroll
    for (int r = 0; r < 4; ++r) dbuf[64 * r + lane] = T[0][r];
  }
  __syncthreads();
  for (int j = 0; j < nbk; ++j) {
    const int par = j & 1;
    XK_WSTAMP(0);
    __syncthreads();                                               // barrier 1: Ls[par] = L_jj^-1
    XK_WSTAMP(2);
    // ---- B: row j of X (the tile above the next diagonal tile first)
    double lv[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) lv[q] = Ls[par][li * 17 + 4 * q + lk];
#pragma unroll
    for (int s = 2; s < NT; ++s) {
      if (si[s] == j) {
        xk_d4 x = {0, 0, 0, 0};
#pragma unroll
        for (int q = 0; q < 4; ++q) x = __builtin_amdgcn_mfma_f64_16x16x4f64(lv[q], T[s][q], x, 0, 0, 0);
        if (s == 2 || s == 3) {
          // (j,j+1) -> the next diagonal tile, from registers: S -= X^T X
#pragma unroll
          for (int q = 0; q < 4; ++q) T[s - 2] = __builtin_amdgcn_mfma_f64_16x16x4f64(-x[q], x[q], T[s - 2], 0, 0, 0);
#pragma unroll
          for (int r = 0; r < 4; ++r) dbuf[64 * r + lane] = T[s - 2][r];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) Xs[par][sk[s]][64 * r + lane] = x[r];
        if (s >= NS) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = 16 * j + lk + 4 * r;
            if (rc_ok && row < a.c) a.X[(size_t)row * a.ld + rcol] = x[r];
          }
        }
      }
    }
    XK_WSTAMP(3);
    if (j + 1 == nbk) break;
    __syncthreads();                                               // barrier 2: Xs[par], dbuf
    XK_WSTAMP(4);
    // ---- C: trailing update (FP64-MFMA bound: three waves per SIMD hide the operand fetches)
#pragma unroll
    for (int s = 0; s < NT; ++s) {
      if (si[s] > j && !(si[s] == j + 1 && sk[s] == j + 1)) {
        const double *xi = &Xs[par][si[s]][lane], *xk = &Xs[par][sk[s]][lane];
#pragma unroll
        for (int q = 0; q < 4; ++q) T[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(-xi[64 * q], xk[64 * q], T[s], 0, 0, 0);
      }
    }
    XK_WSTAMP(5);
  }
}

// corr[i] = sum_k X[k][xoff+i] * y[k] - ct[i]      (K z' - corr_tot, updater.cpp:126)
struct XkCorrArgs {
  const double *X;
  int ld, c, n, xoff, yoff;
  const double *ct;  // may be null
  double *corr;
};
__global__ void xk_corr(XkCorrArgs a) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.n) return;
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  int k = 0;
  for (; k + 4 <= a.c; k += 4) {
    s0 = fma(a.X[(size_t)k * a.ld + a.xoff + i], a.X[(size_t)k * a.ld + a.yoff], s0);
    s1 = fma(a.X[(size_t)(k + 1) * a.ld + a.xoff + i], a.X[(size_t)(k + 1) * a.ld + a.yoff], s1);
    s2 = fma(a.X[(size_t)(k + 2) * a.ld + a.xoff + i], a.X[(size_t)(k + 2) * a.ld + a.yoff], s2);
    s3 = fma(a.X[(size_t)(k + 3) * a.ld + a.xoff + i], a.X[(size_t)(k + 3) * a.ld + a.yoff], s3);
  }
  for (; k < a.c; ++k) s0 = fma(a.X[(size_t)k * a.ld + a.xoff + i], a.X[(size_t)k * a.ld + a.yoff], s0);
  a.corr[i] = ((s0 + s1) + (s2 + s3)) - (a.ct ? a.ct[i] : 0.0);
}

// P' = J P J^T for a sparse J in CSR (StateManager::manage, state_manager.cpp:31-149: feature removal,
// window slide, anchor re-parametrisation and pose augmentation are all congruences with a J that is a
// permutation / identity except for a handful of 3-row blocks with <= 15 non-zeros).  One thread per output
// entry; same association as the reference's (J * cov) * J^T.  P column-major, ld = n.
// Covariance propagation (Propagator::propagateCovarianceMatrices, propagator.cpp:166-205) is the same
// operation with J = blkdiag(F_d, I) plus the process noise Q_d on the core block.
struct XkCongArgs {
  const double *Pin;
  double *Pout;
  int n;
  const int *rp, *ci;   // row pointers [n+1], column indices
  const double *v;      // values
  const double *Q;      // optional: qdim x qdim block (column-major) added at rows/cols [qoff, qoff + qdim)
  int qdim, qoff;
  const double *win_src;   // optional: the frame's window lists, delivered by the operand's copy ...
  double *win_dst;         // ... and left where the per-feature kernels read them
  int win_n;
};
__global__ __launch_bounds__(256) void xk_congruence(XkCongArgs a) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx < a.win_n) a.win_dst[idx] = a.win_src[idx];
  if (idx >= (long)a.n * a.n) return;
  const int r = (int)(idx % a.n), c = (int)(idx / a.n);
  const int r0 = a.rp[r], r1 = a.rp[r + 1], c0 = a.rp[c], c1 = a.rp[c + 1];
  double acc = 0.0;
  for (int ib = c0; ib < c1; ++ib) {
    const double *pb = a.Pin + (size_t)a.ci[ib] * a.n;
    double t = 0.0;
    for (int ia = r0; ia < r1; ++ia) t = fma(a.v[ia], pb[a.ci[ia]], t);   // (J P)[r][b]
    acc = fma(t, a.v[ib], acc);
  }
  if (a.Q && r >= a.qoff && r < a.qoff + a.qdim && c >= a.qoff && c < a.qoff + a.qdim)
    acc += a.Q[(r - a.qoff) + (size_t)(c - a.qoff) * a.qdim];
  a.Pout[idx] = acc;
}

// Propagator::propagateCovarianceMatrices (propagator.cpp:166-205), IN PLACE on the resident covariance:
//   P_ii <- F P_ii F^T + Q   (workgroup 0, through LDS)      P_iv <- F P_iv   (one thread per column)
//   P_vi <- P_vi F^T, from the stored P_vi as the reference insists (one thread per row)      P_vv untouched.
// Every output strip depends only on the same strip of the input, so nothing is read after it was overwritten.
// (As a sparse congruence with J = blkdiag(F, I) the 225 core entries each walked 225 dependent loads: 39 us.)
// f_d and q_d (3.6 KB) travel IN the kernel arguments: no staging copy, no blit kernel in front of the propagation
struct XkPropArgs {
  double *P;
  int n;
  double FQ[450];     // f_d then q_d, 15 x 15 column-major each
};
__global__ __launch_bounds__(256) void xk_cov_propagate_k(XkPropArgs a_) {
  __shared__ double Fs[225], Ts[225];
  // (indexed per lane, so read through the kernarg segment pointer: a by-value array indexed dynamically would be copied
  //  to scratch first)
  const XkPropArgs __attribute__((address_space(4))) &a = *(const XkPropArgs __attribute__((address_space(4))) *)__builtin_amdgcn_kernarg_segment_ptr();
  const int t = threadIdx.x, n = a.n;
  if (t < 225) Fs[t] = a.FQ[t];
  if (blockIdx.x == 0) {
    const int r = t % 15, c = t / 15;
    __syncthreads();
    if (t < 225) {
      double acc = 0.0;
      for (int k = 0; k < 15; ++k) acc = fma(Fs[r + 15 * k], a.P[k + (size_t)c * n], acc);     // (F P_ii)[r][c]
      Ts[t] = acc;
    }
    __syncthreads();
    if (t < 225) {
      double acc = 0.0;
      for (int k = 0; k < 15; ++k) acc = fma(Ts[r + 15 * k], Fs[c + 15 * k], acc);             // ((F P_ii) F^T)[r][c]
      a.P[r + (size_t)c * n] = acc + a.FQ[225 + t];
    }
    return;
  }
  __syncthreads();
  const int nv = n - 15, g = ((int)blockIdx.x - 1) * 256 + t;
  if (g >= 2 * nv) return;
  const int j = 15 + g % nv;
  double in[15], out[15];
  if (g < nv) {                                            // column j of P_iv
    for (int k = 0; k < 15; ++k) in[k] = a.P[k + (size_t)j * n];
    for (int r = 0; r < 15; ++r) {
      double acc = 0.0;
      for (int k = 0; k < 15; ++k) acc = fma(Fs[r + 15 * k], in[k], acc);
      out[r] = acc;
    }
    for (int r = 0; r < 15; ++r) a.P[r + (size_t)j * n] = out[r];
  } else {                                                 // row j of P_vi
    for (int k = 0; k < 15; ++k) in[k] = a.P[j + (size_t)k * n];
    for (int c = 0; c < 15; ++c) {
      double acc = 0.0;
      for (int k = 0; k < 15; ++k) acc = fma(in[k], Fs[c + 15 * k], acc);
      out[c] = acc;
    }
    for (int c = 0; c < 15; ++c) a.P[j + (size_t)c * n] = out[c];
  }
}

// strided copy / scale helpers
struct XkCopyArgs {
  const double *src;
  double *dst;
  int rows, cols;
  long ssr, ssc, dsr, dsc;
};
__global__ void xk_copy2d(XkCopyArgs a) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)a.rows * a.cols) return;
  const int r = (int)(idx % a.rows), c = (int)(idx / a.rows);
  a.dst[(long)r * a.dsr + (long)c * a.dsc] = a.src[(long)r * a.ssr + (long)c * a.ssc];
}

// ----------------------------------------------------------------------------
// Communication-avoiding QR (CAQR) of the tile stack  [H | res]  ->  R  (Householder only).
//
// The columns are walked in panels of 16; per panel
//   (1) xk_caqr_tile: every tile (64 or 128 rows) factors its own panel IN PLACE (pivot rows = its
//       rows 0..15) and applies the 16 reflectors to its trailing columns   -- grid = #tiles x column splits
//   (2) xk_caqr_merge: the 16-row strips [R_loc | C_loc] (rows 0..15 of each tile) are merged A at
//       a time by the same in-place QR on the stacked 16A x 16 panel; A = 20 or 40, so 400 tiles
//       need two levels                                                     -- grid = groups x column splits
//   (3) the root strip of the last level is the next 16 rows of the global R.
// The dependent chain is (C1/16) panels x 3 launches of 16 steps, and step (1) runs on every CU.
// (A binary TSQR tree of whole triangles -- the first version of this stage -- serialises ~9 merges
// of ~350 steps each on ONE workgroup: 2.5 ms where this takes 0.7 ms.)
//
// Column splits: a workgroup holds the 16 panel columns plus `chalf` trailing columns; the trailing
// range is spread over gridDim.y workgroups, each of which factors the panel redundantly (the step
// chain is sequential whoever runs it).  Splits run unsynchronised, so nobody overwrites what a
// sibling still has to read: a split writes only its own trailing columns in place, and the
// 16 x 16 panel block of a tile / merge group goes to a per-level scratch (`pout`) that the next
// level reads as `pin` (the last level writes R).  The panel columns left in the tiles are dead.
//
// Tiles are addressed by position (tile t = rows [TS t, TS t + TS) of A): no compaction list, so a
// workgroup's first loads are the data themselves -- a launch is short enough (~15 us) that every
// dependent round trip to HBM/MALL (~1 us each) shows.  Tiles of rejected tracks are all-zero rows.
//
// Lane layout: NP lanes per column, lane `part` holds RPL rows of the stacked block in registers.
// The pivot row of step kk is register kk of the part-0 lane; the 16 steps are fully unrolled so
// that index is static.  The broadcast vector u has u[<kk] = 0 and u[kk] = v_pivot for the part-0
// lane, so the update code is identical for every lane.
// ----------------------------------------------------------------------------
__device__ __forceinline__ double *xk_tile_opaque(double *p) { asm volatile("" : "+v"(p)); return p; }
#ifndef XK_TILE32_WPE
#define XK_TILE32_WPE 3          // waves per SIMD of the 32-rows-per-lane tile kernels: 3 (one workgroup per CU), 4 = two, reflector in halves
#endif
struct XkCaqrArgs {
  double *A;              // tiles [ntiles][tile_rows_max][C1P] row-major (in place)
  const int *tile_rows;   // valid rows per tile before panel 0 (0 = rejected track)
  int ntiles, TS;         // TS = rows per tile slot (64 or 128)
  int rows_max;           // tallest tile of this update (<= TS): rows past it are all-zero by construction
  int C1P, C1, c0;        // panel = columns [c0, min(c0+16, C1))
  int stride;             // merge: group g merges tiles g*A*stride + u*stride, u = 0..A-1
  int final_level;        // merge: root strip -> Rout
  double *Rout;           // [C1P][C1P] row-major
  int chalf;              // trailing columns per workgroup (gridDim.y workgroups cover the range)
  const double *pin;      // merge: 16 x 16 panel blocks of the level below [block][16][16]
  double *pout;           // panel block of this tile / merge group for the level above
  // overlapped schedule (see xk_caqr_fused): tiles t % hole_stride == 0 are the strips of the LAST merge level
  int hole_stride;        // tile kernel: 0 = no tile has a hole (panel 0, or the plain schedule)
  int lead_off;           // first physical row (0 or 16) of such a tile's pivot strip in this panel
  int lead_all;           // merge: every strip is such a tile (last level) / only strip 0 (first level)
  int pend;               // merge (first level): the group leader's hole rows join as strip number ARITY
  long long *dbg;         // optional: clock stamps of workgroup 0 (probe builds only)
  int wt;                 // experiment: write-through (sc1) stores for the rows a launch hands to the next one
  int lead_stride;        // tile kernel: tiles t % lead_stride == 0 receive merged rows (first-level group leaders); the other
                          // tiles of rejected tracks stay all-zero and are skipped (0 = skip nothing)
  // panel 0 only: tiles [0, nhc) do not exist yet -- the per-feature kernel left the FACTOR RECORDS of their tracks instead
  // (xk_feature.hip.h: XkFeatArgs::Hc; hs doubles per record, the first hcvr of them per row) and the first pass forms its rows
  const double *Hc;
  int hs, hcvr, nhc;
};

__device__ __forceinline__ void xk_store_wt(double *p, double v, int wt) {
  if (wt) __hip_atomic_store(reinterpret_cast<unsigned long long *>(p), __builtin_bit_cast(unsigned long long, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else *p = v;
}

// One Householder step of the in-place panel QR shared by the tile and merge kernels.
//   b[RPL]   this lane's rows of its column;  rel = column index relative to the panel start
//            (trailing columns: anything >= 16);  live = the column exists
//   Un-normalised reflectors  H = I - tt v v^T,  v = [c0 - beta; x_below],  tt = 1/(|beta|(|beta|+|c0|))
//   = y^2 / (1 + |c0| y) with y = 1/|beta| from ONE rsq + Newton.
//
// The reciprocal square root and the reciprocal of the scalar chain take ONE Newton step each after v_rsq_f64 / v_rcp_f64
// (raw: 2^-24; one step: <= 18 ulp = 4e-15; two: <= 1 ulp -- tools/exp/rcp_rsq_probe.hip).  A reflector scale that is off by
// 4e-15 perturbs the product like the rounding of its own 64..128-term dot products does; parity against the oracle is
// unchanged to two digits on the headline, config 2 and priors scaled by 1e2 / 1e4 (tools/exp/ab_parity.sh), and a step's
// dependent chain is four operations shorter: QR stage - 1.3 %.  -DXK_NEWTON2 restores the second step.
// Cost model (measured, tools/exp/clock_probe.hip): a wave issues at most one instruction every 4
// clocks, so a step costs (instructions on the owner's path + instructions on a consumer's path) x 4
// clocks plus two LDS round trips -- neither FMA throughput nor LDS bandwidth.  Hence:
//   * the owner publishes its raw column BEFORE the scalar chain, then patches the part-0 segment
//     (zeros above the pivot, v_pivot at it) so that no consumer has to;
//   * the eliminated entries are never zeroed in registers: rows below the pivot of a finished
//     column are dead, the write-back masks the few that are not;
//   * the owner also publishes -tt itself: a SIMD executes one wave64 VALU instruction per 4 clocks in
//     TOTAL (clock_probe), so six redundant reciprocal instructions on every lane of 6-12 waves cost
//     more than six more instructions on the owner's chain.
// (Several columns per lane -- fewer LDS reads, independent FMA chains -- is slower at every width
//  tried: it lengthens exactly these per-wave instruction streams.)
template <int KK, int NP, int RPL>
__device__ __forceinline__ void xk_caqr_step(double (&b)[RPL], int rel, bool live, int part, double *ubuf, double *sc) {
  constexpr int RPLP = RPL + 2;
  constexpr int pb = KK & 1;
  xk_d2 *useg = reinterpret_cast<xk_d2 *>(ubuf + (pb * NP + part) * RPLP);
  double *scp = sc + pb * 4;
  if (rel == KK) {
#pragma unroll
    for (int r = 0; r < RPL; r += 2) {
      xk_d2 tt = {b[r], b[r + 1]};
      useg[r >> 1] = tt;
    }
    // rows 0..KK of the part-0 lane (R entries and the pivot) do not count: one multiply by a 0/1 lane mask
    // instead of a 64-bit select (two v_cndmask) per such row
    const double below = (part != 0) ? 1.0 : 0.0;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll
    for (int r = 0; r < RPL; ++r) {
      const double v = b[r], vm = (r > KK) ? v : v * below;
      if ((r & 3) == 0) s0 = fma(vm, v, s0); else if ((r & 3) == 1) s1 = fma(vm, v, s1);
      else if ((r & 3) == 2) s2 = fma(vm, v, s2); else s3 = fma(vm, v, s3);
    }
    const double tail = xk_group_sum<NP>((s0 + s1) + (s2 + s3));
    if (part == 0) {
      const double c0v = b[KK];
      double y2 = 0.0, tden = 1.0, vp = 0.0, beta = c0v;
      if (tail > 2.2250738585072014e-308) {
        const double n2 = fma(c0v, c0v, tail);
        double y = __builtin_amdgcn_rsq(n2);             // ~ 1/|beta|
        y = y * fma(-0.5 * n2 * y, y, 1.5);
#ifdef XK_NEWTON2
        y = y * fma(-0.5 * n2 * y, y, 1.5);
#endif
        const double ab = n2 * y;                        // |beta|
        beta = (c0v >= 0) ? -ab : ab;
        vp = c0v - beta;
        y2 = y * y;
        tden = fma(fabs(c0v), y, 1.0);
      }
      // -tt = -y^2 / (1 + |c0| y), once for the whole workgroup: the VALU is shared by all waves of a SIMD
      // (one wave64 instruction per 4 clocks in total), so per-lane redundant scalar work is not free
      double rt = __builtin_amdgcn_rcp(tden);
      rt = fma(rt, fma(-tden, rt, 1.0), rt);
#ifdef XK_NEWTON2
      rt = fma(rt, fma(-tden, rt, 1.0), rt);
#endif
      const double mtt = -(y2 * rt);
      // entries 0..KK of the reflector: rows above the pivot do not take part, the pivot entry is vp
#pragma unroll
      for (int q = 0; q <= KK / 2; ++q) {
        xk_d2 pp;
        pp[0] = (2 * q < KK) ? 0.0 : vp;                                        // 2q == KK otherwise
        pp[1] = (2 * q + 1 < KK) ? 0.0 : (2 * q + 1 == KK) ? vp : b[2 * q + 1];
        useg[q] = pp;
      }
      scp[0] = mtt;
      b[KK] = beta;
    }
  }
  __syncthreads();
  const double mtt = scp[0];
#if XK_TILE32_WPE == 4
  if constexpr (RPL == 32) {
    // 128-VGPR budget (two 8-wave workgroups per CU): the reflector is fetched in halves, once for the dot product and once
    // more for the update -- twice the LDS reads, but b[32] + 16 reflector entries fit where b[32] + 32 do not
    if (rel > KK && live && mtt != 0.0) {
      double d0 = 0.0, d1 = 0.0, d2 = 0.0, d3 = 0.0;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        xk_d2 u[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) u[r] = useg[8 * h + r];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          const int q = 8 * h + r;
          if (r & 1) { d2 = fma(u[r][0], b[2 * q], d2); d3 = fma(u[r][1], b[2 * q + 1], d3); }
          else { d0 = fma(u[r][0], b[2 * q], d0); d1 = fma(u[r][1], b[2 * q + 1], d1); }
        }
      }
      const double w = mtt * xk_group_sum<NP>((d0 + d1) + (d2 + d3));
      const xk_d2 *useg2 = reinterpret_cast<const xk_d2 *>(xk_tile_opaque(reinterpret_cast<double *>(useg)));
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        xk_d2 u[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) u[r] = useg2[8 * h + r];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          const int q = 8 * h + r;
          b[2 * q] = fma(w, u[r][0], b[2 * q]);
          b[2 * q + 1] = fma(w, u[r][1], b[2 * q + 1]);
        }
      }
    }
    return;
  }
#endif
  if (rel > KK && live && mtt != 0.0) {
    // (finished columns do not fetch the reflector: a column is one quarter-wave, so its lanes' LDS passes vanish)
    xk_d2 u[RPL / 2];
#pragma unroll
    for (int r = 0; r < RPL / 2; ++r) u[r] = useg[r];
    double d0 = 0.0, d1 = 0.0, d2 = 0.0, d3 = 0.0;
#pragma unroll
    for (int r = 0; r < RPL / 2; ++r) {
      if (r & 1) { d2 = fma(u[r][0], b[2 * r], d2); d3 = fma(u[r][1], b[2 * r + 1], d3); }
      else { d0 = fma(u[r][0], b[2 * r], d0); d1 = fma(u[r][1], b[2 * r + 1], d1); }
    }
    const double w = mtt * xk_group_sum<NP>((d0 + d1) + (d2 + d3));
#pragma unroll
    for (int r = 0; r < RPL / 2; ++r) {
      b[2 * r] = fma(w, u[r][0], b[2 * r]);
      b[2 * r + 1] = fma(w, u[r][1], b[2 * r + 1]);
    }
  }
}

template <int NP, int RPL>
__device__ __forceinline__ void xk_caqr_steps(double (&b)[RPL], int rel, bool live, int part, int nsteps, double *ubuf, double *sc) {
#define XK_STEP(K) if (K < nsteps) xk_caqr_step<K, NP, RPL>(b, rel, live, part, ubuf, sc);
  XK_STEP(0) XK_STEP(1) XK_STEP(2) XK_STEP(3) XK_STEP(4) XK_STEP(5) XK_STEP(6) XK_STEP(7)
  XK_STEP(8) XK_STEP(9) XK_STEP(10) XK_STEP(11) XK_STEP(12) XK_STEP(13) XK_STEP(14) XK_STEP(15)
#undef XK_STEP
}

// (1) per-tile panel step: 4 lanes per column, RPL = 16 (64-row tiles) or 32 (128-row tiles) rows per lane.
// (RPL = 16 must stay at <= 80 VGPRs so that two 12-wave workgroups share a CU.)
// CSPLIT: the trailing columns are spread over `ysplit` workgroups (systems wider than one workgroup).
// Tiles with a hole (overlapped schedule): 16 of the first 32 physical rows are still being merged by the last
// level of the previous panel; the part-0 lanes take the other 16 as the pivot strip and the hole is skipped.
template <int RPL, bool CSPLIT>
__device__ __forceinline__ void xk_caqr_tile_body(const XkCaqrArgs &a, int t, int ysplit, double *ubuf, double *sc) {
  constexpr int NP = 4;
  const int cidx = (int)threadIdx.x / NP, part = threadIdx.x & (NP - 1);
  const bool panel = cidx < 16;
  const int col = (!CSPLIT || panel) ? a.c0 + cidx : a.c0 + 16 + ysplit * a.chalf + (cidx - 16);
  const bool mine = col < a.C1 && (!CSPLIT || panel || cidx - 16 < a.chalf);
  const bool holed = a.hole_stride > 0 && (t % a.hole_stride) == 0;
  if (a.lead_stride > 0 && (t % a.lead_stride) != 0 && a.tile_rows[t] == 0) {
    // a rejected track that leads no merge group: its rows are zero for the whole factorisation.  The merges read its pivot
    // strip (rows 0..15) and its panel block, so those are zeroed -- the strip once (the feature kernel leaves a rejected
    // track's slot as the previous update left it), the block every panel -- and nothing else is touched.
    if (mine && part == 0) {
      if (panel) {
        if (!CSPLIT || ysplit == 0) {
#pragma unroll
          for (int r = 0; r < 16; ++r) a.pout[((size_t)t * 16 + r) * 16 + cidx] = 0.0;
        }
      } else if (a.c0 == 0) {
        double *z = a.A + (size_t)t * a.TS * a.C1P + col;
#pragma unroll
        for (int r = 0; r < 16; ++r) xk_store_wt(z + (size_t)r * a.C1P, 0.0, a.wt);
      }
    }
    return;
  }
  int prow = part * RPL;
  // the loads do not wait for the row count: rows past it are masked after they arrive
  int rlim = a.rows_max - part * RPL;         // rows of the slot past the tallest staged tile are never touched
  int rlo = 0;                                // holed tiles: physical rows 0..31 are the pivot strip + the hole
  if (holed) {
    if (part == 0) { prow = a.lead_off; rlim = 16; }
    else if (part * RPL < 32) rlo = 32 - part * RPL;
  }
  double *rowp = a.A + ((size_t)t * a.TS + prow) * a.C1P + col;
  double b[RPL];
#ifdef XK_CAQR_PROBE
  const long long t0 = clock64();
#endif
  if (a.c0 == 0 && t < a.nhc) {
    // H0 never existed in HBM: the rows of this track from its factor record (xk_h0_entry, written without its selects as in
    // xk_caqr_pipe.hip.h: r0 = -1 / -2 never meet a row, an untouched column is all zeros, + 0.0 last).  A tile is a track here, so
    // the column part is read once per lane; rows past the track's are masked (the record holds nothing there).
    const double *rec = a.Hc + (size_t)t * a.hs;
    const xk_d2 *wc = reinterpret_cast<const xk_d2 *>(rec + a.hcvr + XK_HC_WC * min(col, a.C1P - 1));
    const xk_d2 q0 = wc[0], q1 = wc[1], q2 = wc[2];
    const int r0 = (int)q2[1], nvalid = a.tile_rows[t] - part * RPL, rtop = a.hcvr / 4 - 1;
    const double nw0 = -q0[0], nw1 = -q0[1], nw2 = -q1[0], x0 = q1[1], x1 = q2[0], wres = (r0 == -1) ? 1.0 : 0.0;
#pragma unroll
    for (int r = 0; r < RPL; ++r) {
      const int rr = prow + r + 3, dd = rr - r0;
      const xk_d2 *vr = reinterpret_cast<const xk_d2 *>(rec + 4 * min(rr, rtop));
      const xk_d2 va = vr[0], vb = vr[1];
      const double add = (dd == 0) ? x0 : ((dd == 1) ? x1 : 0.0);
      const double v = fma(wres, vb[1], fma(nw2, vb[0], fma(nw1, va[1], nw0 * va[0]))) + add;
      b[r] = (mine && r < nvalid) ? v : 0.0;
    }
  } else {
#pragma unroll
    for (int r = 0; r < RPL; ++r) b[r] = (mine && r < rlim && r >= rlo) ? rowp[(size_t)r * a.C1P] : 0.0;
    if (a.c0 == 0) {
      const int nvalid = a.tile_rows[t] - part * RPL;
#pragma unroll
      for (int r = 0; r < RPL; ++r) b[r] = (r < nvalid) ? b[r] : 0.0;
    }
  }
#ifdef XK_CAQR_PROBE
  double sink = 0; for (int r = 0; r < RPL; ++r) sink += b[r];
  asm volatile("" :: "v"(sink));
  const long long t1 = clock64(), w1 = wall_clock64();
#endif
  const int nsteps = (a.C1 - a.c0 < 16) ? a.C1 - a.c0 : 16;
  xk_caqr_steps<NP, RPL>(b, cidx, mine, part, nsteps, ubuf, sc);
#ifdef XK_CAQR_PROBE
  const long long t2 = clock64();
  if (a.dbg && t == 0 && ysplit == 0 && threadIdx.x == 0) { a.dbg[0] = t1 - t0; a.dbg[1] = t2 - t1; a.dbg[2] = wall_clock64() - w1; a.dbg[3] = nsteps; }
#endif
  if (!mine) return;
  if (panel) {
    // the strip's panel block (upper triangle, zeros below: eliminated entries are not zeroed in registers)
    // is the next level's input; rows 16.. of a finished column are dead
    if (part == 0 && (!CSPLIT || ysplit == 0)) {
#pragma unroll
      for (int r = 0; r < 16; ++r) a.pout[((size_t)t * 16 + r) * 16 + cidx] = (r > cidx) ? 0.0 : b[r];
    }
  } else {
#pragma unroll
    for (int r = 0; r < RPL; ++r)
      if (r < rlim && r >= rlo) xk_store_wt(rowp + (size_t)r * a.C1P, b[r], a.wt);
  }
}

// RPL = 26: 128-row slots whose tallest tile has <= 104 rows (windows of 34..53 poses).  b[26] + the reflector fit 128 VGPRs,
// so TWO 8-wave workgroups share a CU and one's tile traffic hides behind the other's steps (RPL = 32 needs 168: one per CU).
#define XK_TILE_WAVES_PER_EU(RPL) ((RPL) == 16 ? 6 : (RPL) == 26 ? 4 : XK_TILE32_WPE)
template <int RPL, bool CSPLIT>
__global__ __launch_bounds__(RPL == 16 ? 768 : 512) __attribute__((amdgpu_waves_per_eu(XK_TILE_WAVES_PER_EU(RPL)))) void xk_caqr_tile(XkCaqrArgs a) {
  constexpr int NP = 4, RPLP = RPL + 2;
  __shared__ __attribute__((aligned(16))) double ubuf[2 * NP * RPLP];
  __shared__ __attribute__((aligned(16))) double sc[2 * 4];
  xk_caqr_tile_body<RPL, CSPLIT>(a, blockIdx.x, blockIdx.y, ubuf, sc);
}

// (2) A-way strip merge, A = RPL strips of 16 rows.  Lane layout, transposed with respect to the tile
// kernel: 16 lanes per column, lane p holds ROW p of EVERY strip (register s = strip s).  The pivot strip
// is then register 0 of the 16 lanes: pivot row kk sits in lane kk, so
//   * only register 0 needs a lane mask in the norm (rows above the pivot = lanes < kk),
//   * the lane that owns the pivot runs the scalar chain on its own register -- no broadcast --
//     and patches ONE reflector entry afterwards (the tile layout rewrites kk/2 + 1 vector entries),
//   * the merged 16 x 16 block / the next rows of R are one register across 16 lanes, and a strip is
//     addressed by one uniform stride.
// (~25 fewer instructions on the owner's path of a step; measured step time is unchanged -- the owner is
//  not what the other waves wait for -- so this layout is kept for its simpler addressing.)
template <int KK, int RPL>
__device__ __forceinline__ void xk_caqr_mstep(double (&b)[RPL], int rel, bool live, int part, double *ubuf, double *sc) {
  constexpr int NP = 16, RPLP = RPL + 2;
  constexpr int pb = KK & 1;
  xk_d2 *useg = reinterpret_cast<xk_d2 *>(ubuf + (pb * NP + part) * RPLP);
  double *scp = sc + pb * 4;
  if (rel == KK) {
    // raw column out first (the LDS write latency hides behind the reduction and the scalar chain); rows of
    // the pivot strip above the pivot are not part of the reflector
    const double below = (part > KK) ? 1.0 : 0.0, at_or_below = (part >= KK) ? 1.0 : 0.0;
    {
      xk_d2 t0 = {b[0] * at_or_below, b[1]};
      useg[0] = t0;
    }
#pragma unroll
    for (int r = 2; r < RPL; r += 2) {
      xk_d2 tt = {b[r], b[r + 1]};
      useg[r >> 1] = tt;
    }
    double s0 = (b[0] * below) * b[0], s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll
    for (int r = 1; r < RPL; ++r) {
      if ((r & 3) == 0) s0 = fma(b[r], b[r], s0); else if ((r & 3) == 1) s1 = fma(b[r], b[r], s1);
      else if ((r & 3) == 2) s2 = fma(b[r], b[r], s2); else s3 = fma(b[r], b[r], s3);
    }
    const double tail = xk_group_sum<NP>((s0 + s1) + (s2 + s3));
    if (part == KK) {
      const double c0v = b[0];
      double y2 = 0.0, tden = 1.0, vp = 0.0, beta = c0v;
      if (tail > 2.2250738585072014e-308) {
        const double n2 = fma(c0v, c0v, tail);
        double y = __builtin_amdgcn_rsq(n2);             // ~ 1/|beta|
        y = y * fma(-0.5 * n2 * y, y, 1.5);
#ifdef XK_NEWTON2
        y = y * fma(-0.5 * n2 * y, y, 1.5);
#endif
        const double ab = n2 * y;                        // |beta|
        beta = (c0v >= 0) ? -ab : ab;
        vp = c0v - beta;
        y2 = y * y;
        tden = fma(fabs(c0v), y, 1.0);
      }
      // -tt = -y^2 / (1 + |c0| y), once for the whole workgroup: the VALU is shared by all waves of a SIMD
      // (one wave64 instruction per 4 clocks in total), so per-lane redundant scalar work is not free
      double rt = __builtin_amdgcn_rcp(tden);
      rt = fma(rt, fma(-tden, rt, 1.0), rt);
#ifdef XK_NEWTON2
      rt = fma(rt, fma(-tden, rt, 1.0), rt);
#endif
      const double mtt = -(y2 * rt);
      ubuf[(pb * NP + part) * RPLP] = vp;                // the pivot entry of the reflector
      scp[0] = mtt;
      b[0] = beta;
    }
  }
  __syncthreads();
  const double mtt = scp[0];
  if (rel > KK && live && mtt != 0.0) {
    // (finished columns do not fetch the reflector: a column is one quarter-wave, so its lanes' LDS passes vanish)
    xk_d2 u[RPL / 2];
#pragma unroll
    for (int r = 0; r < RPL / 2; ++r) u[r] = useg[r];
    double d0 = 0.0, d1 = 0.0, d2 = 0.0, d3 = 0.0;
#pragma unroll
    for (int r = 0; r < RPL / 2; ++r) {
      if (r & 1) { d2 = fma(u[r][0], b[2 * r], d2); d3 = fma(u[r][1], b[2 * r + 1], d3); }
      else { d0 = fma(u[r][0], b[2 * r], d0); d1 = fma(u[r][1], b[2 * r + 1], d1); }
    }
    const double w = mtt * xk_group_sum<NP>((d0 + d1) + (d2 + d3));
#pragma unroll
    for (int r = 0; r < RPL / 2; ++r) {
      b[2 * r] = fma(w, u[r][0], b[2 * r]);
      b[2 * r + 1] = fma(w, u[r][1], b[2 * r + 1]);
    }
  }
}

// ----------------------------------------------------------------------------
// The same steps with a ONE-REFLECTOR LOOK-AHEAD (register-resident kernel): iteration K of a panel is
//     every lane: apply reflector K - 1;   the owner of column K: form reflector K and publish it;   barrier
// instead of  owner: form K; barrier; everyone: apply K.  The arithmetic and its order are identical; what changes is
// the critical path of a step: the owner's wave no longer waits for the slowest of the three waves that share its SIMD
// to finish applying the previous reflector before it starts the next scalar chain -- with the waves that hold panel
// columns raised in priority (s_setprio), their apply runs first and the chain overlaps the other waves' apply.
// Double-buffered like xk_caqr_step: reflector K lives in buffer K & 1, rewritten two iterations later, a barrier after
// its last reader.
template <int KK, int NP, int RPL>
__device__ __forceinline__ void xk_caqr_form(double (&b)[RPL], int rel, int part, double *ubuf, double *sc) {
  constexpr int RPLP = RPL + 2;
  constexpr int pb = KK & 1;
  xk_d2 *useg = reinterpret_cast<xk_d2 *>(ubuf + (pb * NP + part) * RPLP);
  double *scp = sc + pb * 4;
  if (rel != KK) return;
  // rows 0..KK of the part-0 lane (R entries and the pivot) are no part of the reflector: the 0/1 lane mask that keeps
  // them out of the norm also makes the published column clean, so that only the pivot entry is patched after the chain
  const double below = (part != 0) ? 1.0 : 0.0;
  double vm[RPL];
#pragma unroll
  for (int r = 0; r < RPL; ++r) vm[r] = (r > KK) ? b[r] : b[r] * below;
#pragma unroll
  for (int r = 0; r < RPL; r += 2) {
    xk_d2 tt = {vm[r], vm[r + 1]};
    useg[r >> 1] = tt;
  }
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll
  for (int r = 0; r < RPL; ++r) {
    if ((r & 3) == 0) s0 = fma(vm[r], b[r], s0); else if ((r & 3) == 1) s1 = fma(vm[r], b[r], s1);
    else if ((r & 3) == 2) s2 = fma(vm[r], b[r], s2); else s3 = fma(vm[r], b[r], s3);
  }
  const double tail = xk_group_sum<NP>((s0 + s1) + (s2 + s3));
  if (part == 0) {
    const double c0v = b[KK];
    double y2 = 0.0, tden = 1.0, vp = 0.0, beta = c0v;
    if (tail > 2.2250738585072014e-308) {
      const double n2 = fma(c0v, c0v, tail);
      double y = __builtin_amdgcn_rsq(n2);
      y = y * fma(-0.5 * n2 * y, y, 1.5);
#ifdef XK_NEWTON2
      y = y * fma(-0.5 * n2 * y, y, 1.5);
#endif
      const double ab = n2 * y;
      beta = (c0v >= 0) ? -ab : ab;
      vp = c0v - beta;
      y2 = y * y;
      tden = fma(fabs(c0v), y, 1.0);
    }
    double rt = __builtin_amdgcn_rcp(tden);
    rt = fma(rt, fma(-tden, rt, 1.0), rt);
#ifdef XK_NEWTON2
    rt = fma(rt, fma(-tden, rt, 1.0), rt);
#endif
    const double mtt = -(y2 * rt);
    ubuf[(pb * NP + part) * RPLP + KK] = vp;             // the pivot entry of the reflector
    scp[0] = mtt;
    b[KK] = beta;
  }
}

template <int KK, int NP, int RPL>
__device__ __forceinline__ void xk_caqr_apply(double (&b)[RPL], int rel, bool live, int part, const double *ubuf, const double *sc) {
  constexpr int RPLP = RPL + 2;
  constexpr int pb = KK & 1;
  const xk_d2 *useg = reinterpret_cast<const xk_d2 *>(ubuf + (pb * NP + part) * RPLP);
  const double mtt = sc[pb * 4];
  if (rel > KK && live && mtt != 0.0) {
    xk_d2 u[RPL / 2];
#pragma unroll
    for (int r = 0; r < RPL / 2; ++r) u[r] = useg[r];
    double d0 = 0.0, d1 = 0.0, d2 = 0.0, d3 = 0.0;
#pragma unroll
    for (int r = 0; r < RPL / 2; ++r) {
      if (r & 1) { d2 = fma(u[r][0], b[2 * r], d2); d3 = fma(u[r][1], b[2 * r + 1], d3); }
      else { d0 = fma(u[r][0], b[2 * r], d0); d1 = fma(u[r][1], b[2 * r + 1], d1); }
    }
    const double w = mtt * xk_group_sum<NP>((d0 + d1) + (d2 + d3));
#pragma unroll
    for (int r = 0; r < RPL / 2; ++r) {
      b[2 * r] = fma(w, u[r][0], b[2 * r]);
      b[2 * r + 1] = fma(w, u[r][1], b[2 * r + 1]);
    }
  }
}

// `hot` = this wave holds panel columns (wave-uniform)
template <int NP, int RPL>
__device__ __forceinline__ void xk_caqr_steps_la(double (&b)[RPL], int rel, bool live, int part, int nsteps, bool hot, double *ubuf, double *sc) {
  if (hot) __builtin_amdgcn_s_setprio(3);
  xk_caqr_form<0, NP, RPL>(b, rel, part, ubuf, sc);
  __syncthreads();
#define XK_IT(K)                                                                                  \
  if (K < nsteps) { xk_caqr_apply<K - 1, NP, RPL>(b, rel, live, part, ubuf, sc); xk_caqr_form<K, NP, RPL>(b, rel, part, ubuf, sc); __syncthreads(); } \
  else if (K == nsteps) xk_caqr_apply<K - 1, NP, RPL>(b, rel, live, part, ubuf, sc);
  XK_IT(1) XK_IT(2) XK_IT(3) XK_IT(4) XK_IT(5) XK_IT(6) XK_IT(7) XK_IT(8)
  XK_IT(9) XK_IT(10) XK_IT(11) XK_IT(12) XK_IT(13) XK_IT(14) XK_IT(15)
#undef XK_IT
  if (nsteps == 16) xk_caqr_apply<15, NP, RPL>(b, rel, live, part, ubuf, sc);
  if (hot) __builtin_amdgcn_s_setprio(0);
}

// the merge layout (xk_caqr_mstep): 16 lanes per column, lane p = row p of every strip
template <int KK, int RPL>
__device__ __forceinline__ void xk_caqr_mform(double (&b)[RPL], int rel, int part, double *ubuf, double *sc) {
  constexpr int NP = 16, RPLP = RPL + 2;
  constexpr int pb = KK & 1;
  xk_d2 *useg = reinterpret_cast<xk_d2 *>(ubuf + (pb * NP + part) * RPLP);
  double *scp = sc + pb * 4;
  if (rel != KK) return;
  const double below = (part > KK) ? 1.0 : 0.0, at_or_below = (part >= KK) ? 1.0 : 0.0;
  {
    xk_d2 t0 = {b[0] * at_or_below, b[1]};
    useg[0] = t0;
  }
#pragma unroll
  for (int r = 2; r < RPL; r += 2) {
    xk_d2 tt = {b[r], b[r + 1]};
    useg[r >> 1] = tt;
  }
  double s0 = (b[0] * below) * b[0], s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll
  for (int r = 1; r < RPL; ++r) {
    if ((r & 3) == 0) s0 = fma(b[r], b[r], s0); else if ((r & 3) == 1) s1 = fma(b[r], b[r], s1);
    else if ((r & 3) == 2) s2 = fma(b[r], b[r], s2); else s3 = fma(b[r], b[r], s3);
  }
  const double tail = xk_group_sum<NP>((s0 + s1) + (s2 + s3));
  if (part == KK) {
    const double c0v = b[0];
    double y2 = 0.0, tden = 1.0, vp = 0.0, beta = c0v;
    if (tail > 2.2250738585072014e-308) {
      const double n2 = fma(c0v, c0v, tail);
      double y = __builtin_amdgcn_rsq(n2);
      y = y * fma(-0.5 * n2 * y, y, 1.5);
#ifdef XK_NEWTON2
      y = y * fma(-0.5 * n2 * y, y, 1.5);
#endif
      const double ab = n2 * y;
      beta = (c0v >= 0) ? -ab : ab;
      vp = c0v - beta;
      y2 = y * y;
      tden = fma(fabs(c0v), y, 1.0);
    }
    double rt = __builtin_amdgcn_rcp(tden);
    rt = fma(rt, fma(-tden, rt, 1.0), rt);
#ifdef XK_NEWTON2
    rt = fma(rt, fma(-tden, rt, 1.0), rt);
#endif
    const double mtt = -(y2 * rt);
    ubuf[(pb * NP + part) * RPLP] = vp;
    scp[0] = mtt;
    b[0] = beta;
  }
}

template <int RPL>
__device__ __forceinline__ void xk_caqr_msteps_la(double (&b)[RPL], int rel, bool live, int part, int nsteps, bool hot, double *ubuf, double *sc) {
  if (hot) __builtin_amdgcn_s_setprio(3);
  xk_caqr_mform<0, RPL>(b, rel, part, ubuf, sc);
  __syncthreads();
#define XK_IT(K)                                                                                  \
  if (K < nsteps) { xk_caqr_apply<K - 1, 16, RPL>(b, rel, live, part, ubuf, sc); xk_caqr_mform<K, RPL>(b, rel, part, ubuf, sc); __syncthreads(); } \
  else if (K == nsteps) xk_caqr_apply<K - 1, 16, RPL>(b, rel, live, part, ubuf, sc);
  XK_IT(1) XK_IT(2) XK_IT(3) XK_IT(4) XK_IT(5) XK_IT(6) XK_IT(7) XK_IT(8)
  XK_IT(9) XK_IT(10) XK_IT(11) XK_IT(12) XK_IT(13) XK_IT(14) XK_IT(15)
#undef XK_IT
  if (nsteps == 16) xk_caqr_apply<15, 16, RPL>(b, rel, live, part, ubuf, sc);
  if (hot) __builtin_amdgcn_s_setprio(0);
}

// Merge body.  RPL = ARITY (20, 40) or ARITY + 2 (22, 42: register ARITY is the pending strip of the
// overlapped schedule).  `group` / `split` = which strips / which trailing columns this workgroup owns.
template <int RPL>
__device__ __forceinline__ void xk_caqr_merge_body(const XkCaqrArgs &a, int group, int split, double *ubuf, double *sc) {
  constexpr int NP = 16, ARITY = (RPL % 20 == 0) ? RPL : RPL - 2;
  constexpr bool PEND = ARITY != RPL;
  static_assert(RPL % 2 == 0, "reflector segments are read two doubles at a time");
#ifdef XK_CAQR_PROBE
  const long long w0 = wall_clock64();
#endif
  const int cidx = (int)threadIdx.x / NP, part = threadIdx.x & (NP - 1);
  const bool panel = cidx < 16;
  const int col = panel ? a.c0 + cidx : a.c0 + 16 + split * a.chalf + (cidx - 16);
  const bool mine = col < a.C1 && (panel || cidx - 16 < a.chalf);
  const int base = group * ARITY * a.stride;
  // strip s of this group = the pivot strip of tile base + s*stride (trailing columns) or block
  // group*ARITY + s of the level below (panel columns); this lane's row of it is `part`.  The pivot strip
  // is rows 0..15 of the tile, except rows lead_off.. for the tiles the last level works on.
  const size_t lane_off = panel ? (size_t)part * 16 + cidx : (size_t)part * a.C1P + col;
  const size_t strip_step = panel ? 256 : (size_t)a.stride * a.TS * a.C1P;
  const size_t lead = panel ? 0 : (size_t)a.lead_off * a.C1P;
  double *g0 = panel ? const_cast<double *>(a.pin) + (size_t)group * ARITY * 256 + lane_off
                     : a.A + (size_t)base * a.TS * a.C1P + lane_off + (a.lead_all ? lead : 0);
  double *g00 = g0 + (a.lead_all ? 0 : lead);                                      // strip 0
  double *gp = a.A + ((size_t)base * a.TS + (16 - a.lead_off) + part) * a.C1P + col;   // pending strip (hole rows)
  const int nstrips = min(ARITY, (a.ntiles - base + a.stride - 1) / a.stride);   // strips that exist
  double b[RPL];
  b[0] = mine ? g00[0] : 0.0;
#pragma unroll
  for (int r = 1; r < ARITY; ++r) b[r] = (mine && r < nstrips) ? g0[(size_t)r * strip_step] : 0.0;
  if (PEND) {
    b[ARITY] = (mine && a.pend) ? gp[0] : 0.0;
    b[ARITY + 1] = 0.0;
  }
  const int nsteps = (a.C1 - a.c0 < 16) ? a.C1 - a.c0 : 16;
#ifdef XK_CAQR_PROBE
  double sink = 0; for (int r = 0; r < RPL; ++r) sink += b[r];
  asm volatile("" :: "v"(sink));
  const long long t1 = clock64(), w1 = wall_clock64();
#endif
#define XK_STEP(K) if (K < nsteps) xk_caqr_mstep<K, RPL>(b, cidx, mine, part, ubuf, sc);
  XK_STEP(0) XK_STEP(1) XK_STEP(2) XK_STEP(3) XK_STEP(4) XK_STEP(5) XK_STEP(6) XK_STEP(7)
  XK_STEP(8) XK_STEP(9) XK_STEP(10) XK_STEP(11) XK_STEP(12) XK_STEP(13) XK_STEP(14) XK_STEP(15)
#undef XK_STEP
#ifdef XK_CAQR_PROBE
  const long long w2 = wall_clock64();
  if (a.dbg && group == 0 && split == 0 && threadIdx.x == 0) {
    a.dbg[1] = clock64() - t1; a.dbg[2] = w2 - w1; a.dbg[3] = nsteps;
    a.dbg[4] = w0; a.dbg[5] = w1; a.dbg[6] = w2;
  }
#endif
  if (!mine) return;
  if (panel) {
    // the merged panel block = register 0 across the 16 lanes; split 0 publishes it
    if (split == 0) {
      const double v = (part > cidx) ? 0.0 : b[0];   // eliminated entries are not zeroed in registers
      if (a.final_level) { if (a.c0 + part < a.C1) a.Rout[(size_t)(a.c0 + part) * a.C1P + col] = v; }
      else a.pout[(size_t)group * 256 + part * 16 + cidx] = v;
    }
  } else {
    if (a.final_level) {                              // row `part` of the root strip = row c0 + part of R
      if (a.c0 + part < a.C1) a.Rout[(size_t)(a.c0 + part) * a.C1P + col] = b[0];
      b[0] = 0.0;
    }
    xk_store_wt(g00, b[0], a.wt);
#pragma unroll
    for (int r = 1; r < ARITY; ++r)
      if (r < nstrips) xk_store_wt(g0 + (size_t)r * strip_step, b[r], a.wt);
    if (PEND) { if (a.pend) xk_store_wt(gp, b[ARITY], a.wt); }
  }
#ifdef XK_CAQR_PROBE
  if (a.dbg && group == 0 && split == 0 && threadIdx.x == 0) {
    __builtin_amdgcn_s_waitcnt(0);
    a.dbg[7] = wall_clock64();
  }
#endif
}

template <int RPL>
__global__ __launch_bounds__(RPL > 22 ? 512 : 1024) void xk_caqr_merge(XkCaqrArgs a) {
  constexpr int NP = 16, RPLP = RPL + 2;
  __shared__ __attribute__((aligned(16))) double ubuf[2 * NP * RPLP];
  __shared__ __attribute__((aligned(16))) double sc[2 * 4];
  xk_caqr_merge_body<RPL>(a, blockIdx.x, blockIdx.y, ubuf, sc);
}

// The last merge level again, for the copy that shares a kernel (and its 80-VGPR budget) with the tile step:
// 32 lanes per column, lane (h, p) = row p of strips 10h .. 10h+9, so a lane holds 10 rows and the reflector
// (10 more) stays in registers -- the 16-lane layout above needs 20 + 20 and had to fetch the reflector twice.
// The reduction over a column's 32 lanes is the 16-lane DPP butterfly plus ONE v_permlane16_swap (rows 2i and
// 2i+1 of the wave trade places: x + swap(x) is the sum over the row pair in both rows).
// doubles per lane of the reflector broadcast buffer: RH + 2, bumped when that makes the lane stride a multiple of 128 B
// (RH = 14: all 32 lanes of a column would sit on the same banks)
// (RH = 22, the first level below: 26 doubles = 13 x 16 B, an odd number of 16-byte bank groups, so that the 32 lanes of a
//  column read their segments conflict-free)
#define XK_M32_STRIDE(RH) ((RH) == 22 ? 26 : (((RH) + 2) % 16 == 0) ? (RH) + 4 : (RH) + 2)
__device__ __forceinline__ double xk_rowpair_sum(double x) {
  const long long q = __builtin_bit_cast(long long, x);
  const unsigned lo = (unsigned)q, hi = (unsigned)(q >> 32);
  const auto r0 = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
  const auto r1 = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  const double a = __builtin_bit_cast(double, ((long long)r1[0] << 32) | (unsigned int)r0[0]);
  const double b = __builtin_bit_cast(double, ((long long)r1[1] << 32) | (unsigned int)r0[1]);
  return a + b;
}
template <int KK, int RH>
__device__ __forceinline__ void xk_caqr_mstep32(double (&b)[RH], int rel, bool live, int part, double *ubuf, double *sc) {
  constexpr int NP = 32, RHP = XK_M32_STRIDE(RH);
  constexpr int pb = KK & 1;
  const int p = part & 15, half = part >> 4;
  xk_d2 *useg = reinterpret_cast<xk_d2 *>(ubuf + (pb * NP + part) * RHP);
  double *scp = sc + pb * 4;
  if (rel == KK) {
    // strip 0 (register 0 of half 0) is the pivot strip: its rows above the pivot are not part of the reflector
    const double below = (half || p > KK) ? 1.0 : 0.0, at_or_below = (half || p >= KK) ? 1.0 : 0.0;
    {
      xk_d2 t0 = {b[0] * at_or_below, b[1]};
      useg[0] = t0;
    }
#pragma unroll
    for (int r = 2; r < RH; r += 2) {
      xk_d2 tt = {b[r], b[r + 1]};
      useg[r >> 1] = tt;
    }
    double s0 = (b[0] * below) * b[0], s1 = 0.0;
#pragma unroll
    for (int r = 1; r < RH; ++r) {
      if (r & 1) s1 = fma(b[r], b[r], s1); else s0 = fma(b[r], b[r], s0);
    }
    const double tail = xk_rowpair_sum(xk_group_sum<16>(s0 + s1));
    if (part == KK) {
      const double c0v = b[0];
      double y2 = 0.0, tden = 1.0, vp = 0.0, beta = c0v;
      if (tail > 2.2250738585072014e-308) {
        const double n2 = fma(c0v, c0v, tail);
        double y = __builtin_amdgcn_rsq(n2);
        y = y * fma(-0.5 * n2 * y, y, 1.5);
#ifdef XK_NEWTON2
        y = y * fma(-0.5 * n2 * y, y, 1.5);
#endif
        const double ab = n2 * y;
        beta = (c0v >= 0) ? -ab : ab;
        vp = c0v - beta;
        y2 = y * y;
        tden = fma(fabs(c0v), y, 1.0);
      }
      double rt = __builtin_amdgcn_rcp(tden);
      rt = fma(rt, fma(-tden, rt, 1.0), rt);
#ifdef XK_NEWTON2
      rt = fma(rt, fma(-tden, rt, 1.0), rt);
#endif
      const double mtt = -(y2 * rt);
      ubuf[(pb * NP + part) * RHP] = vp;                 // the pivot entry of the reflector
      scp[0] = mtt;
      b[0] = beta;
    }
  }
  __syncthreads();
  const double mtt = scp[0];
  if (rel > KK && live && mtt != 0.0) {
    xk_d2 u[RH / 2];
#pragma unroll
    for (int r = 0; r < RH / 2; ++r) u[r] = useg[r];
    double d0 = 0.0, d1 = 0.0;
#pragma unroll
    for (int r = 0; r < RH / 2; ++r) { d0 = fma(u[r][0], b[2 * r], d0); d1 = fma(u[r][1], b[2 * r + 1], d1); }
    const double w = mtt * xk_rowpair_sum(xk_group_sum<16>(d0 + d1));
#pragma unroll
    for (int r = 0; r < RH / 2; ++r) {
      b[2 * r] = fma(w, u[r][0], b[2 * r]);
      b[2 * r + 1] = fma(w, u[r][1], b[2 * r + 1]);
    }
  }
}

// Last-level merge body in that layout: ONE group of up to 20 strips (all of them pivot strips of first-level
// group leaders: lead_all), `split` = which trailing columns.  Same addressing as xk_caqr_merge_body.
__device__ __forceinline__ void xk_caqr_last32_body(const XkCaqrArgs &a, int split, double *ubuf, double *sc) {
  constexpr int NP = 32, RH = 10, ARITY = 20;
  const int cidx = (int)threadIdx.x / NP, part = threadIdx.x & (NP - 1);
  const int p = part & 15, half = part >> 4;
  const bool panel = cidx < 16;
  const int col = panel ? a.c0 + cidx : a.c0 + 16 + split * a.chalf + (cidx - 16);
  const bool mine = col < a.C1 && (panel || cidx - 16 < a.chalf);
  const size_t lane_off = panel ? (size_t)p * 16 + cidx : (size_t)p * a.C1P + col;
  const size_t strip_step = panel ? 256 : (size_t)a.stride * a.TS * a.C1P;
  double *g0 = (panel ? const_cast<double *>(a.pin) + lane_off : a.A + lane_off + (size_t)a.lead_off * a.C1P) + (size_t)(RH * half) * strip_step;
  const int nstrips = min(ARITY, (a.ntiles + a.stride - 1) / a.stride) - RH * half;   // of this half
  double b[RH];
#pragma unroll
  for (int r = 0; r < RH; ++r) b[r] = (mine && r < nstrips) ? g0[(size_t)r * strip_step] : 0.0;
  const int nsteps = (a.C1 - a.c0 < 16) ? a.C1 - a.c0 : 16;
#define XK_STEP(K) if (K < nsteps) xk_caqr_mstep32<K, RH>(b, cidx, mine, part, ubuf, sc);
  XK_STEP(0) XK_STEP(1) XK_STEP(2) XK_STEP(3) XK_STEP(4) XK_STEP(5) XK_STEP(6) XK_STEP(7)
  XK_STEP(8) XK_STEP(9) XK_STEP(10) XK_STEP(11) XK_STEP(12) XK_STEP(13) XK_STEP(14) XK_STEP(15)
#undef XK_STEP
  if (!mine) return;
  const bool root = half == 0;                               // register 0 of half 0 = the root strip
  if (panel) {
    if (split == 0 && root) {
      const double v = (p > cidx) ? 0.0 : b[0];              // eliminated entries are not zeroed in registers
      if (a.c0 + p < a.C1) a.Rout[(size_t)(a.c0 + p) * a.C1P + col] = v;
    }
  } else {
    if (root) {                                              // row p of the root strip = row c0 + p of R
      if (a.c0 + p < a.C1) a.Rout[(size_t)(a.c0 + p) * a.C1P + col] = b[0];
      b[0] = 0.0;
    }
#pragma unroll
    for (int r = 0; r < RH; ++r)
      if (r < nstrips) xk_store_wt(g0 + (size_t)r * strip_step, b[r], a.wt);
  }
}

// First merge level of the overlapped schedule at arity 40 in the 32-lane layout: lane (h, p) = row p of slots 22h .. 22h+21,
// slot s < 40 = strip s of the group, slot 40 = the pending strip (the leader's hole rows), slots 41..43 empty.  22 rows and
// 22 reflector entries per lane (the 16-lane xk_caqr_merge<42> holds 42 + 42 and its per-step instruction stream is twice as
// long: 43 us per launch at config 3 against 2x us here), 16 waves per workgroup instead of 8.
__device__ __forceinline__ void xk_caqr_first32_body(const XkCaqrArgs &a, int group, int split, double *ubuf, double *sc) {
  constexpr int NP = 32, RH = 22, ARITY = 40, PR = ARITY - RH;     // PR = register of the pending strip in half 1
  const int cidx = (int)threadIdx.x / NP, part = threadIdx.x & (NP - 1);
  const int p = part & 15, half = part >> 4;
  // (16 panel columns x 32 lanes = 8 whole waves: `panel` is wave-uniform, and so are the strides that depend on it)
  const bool panel = __builtin_amdgcn_readfirstlane((int)(cidx < 16)) != 0;
  const int col = panel ? a.c0 + cidx : a.c0 + 16 + split * a.chalf + (cidx - 16);
  const bool mine = col < a.C1 && (panel || cidx - 16 < a.chalf);
  const int base = group * ARITY * a.stride;
  const size_t lane_off = panel ? (size_t)p * 16 + cidx : (size_t)p * a.C1P + col;
  const size_t strip_step = panel ? 256 : (size_t)a.stride * a.TS * a.C1P;
  const size_t lead = panel ? 0 : (size_t)a.lead_off * a.C1P;
  double *g0 = panel ? const_cast<double *>(a.pin) + (size_t)group * ARITY * 256 + lane_off
                     : a.A + (size_t)base * a.TS * a.C1P + lane_off + (a.lead_all ? lead : 0);
  double *g00 = g0 + (a.lead_all ? 0 : lead);                                      // strip 0: the leader's pivot strip
  double *gp = a.A + ((size_t)base * a.TS + (16 - a.lead_off) + p) * a.C1P + col;   // pending strip (hole rows)
  double *gh = g0 + (size_t)(RH * half) * strip_step;                              // slot RH * half
  const int nstrips = min(ARITY, (a.ntiles - base + a.stride - 1) / a.stride) - RH * half;   // strips of this half that exist
  double b[RH];
#pragma unroll
  for (int r = 0; r < RH; ++r) {
    double *q = gh + (size_t)r * strip_step;
    bool ok = mine && r < nstrips;
    if (r == 0 && !half) q = g00;
    if (r >= PR && half) { q = gp; ok = mine && r == PR && a.pend; }
    b[r] = ok ? q[0] : 0.0;
  }
  const int nsteps = (a.C1 - a.c0 < 16) ? a.C1 - a.c0 : 16;
#define XK_STEP(K) if (K < nsteps) xk_caqr_mstep32<K, RH>(b, cidx, mine, part, ubuf, sc);
  XK_STEP(0) XK_STEP(1) XK_STEP(2) XK_STEP(3) XK_STEP(4) XK_STEP(5) XK_STEP(6) XK_STEP(7)
  XK_STEP(8) XK_STEP(9) XK_STEP(10) XK_STEP(11) XK_STEP(12) XK_STEP(13) XK_STEP(14) XK_STEP(15)
#undef XK_STEP
  if (!mine) return;
  if (panel) {
    // the merged panel block = register 0 of half 0 across its 16 lanes; split 0 publishes it
    if (split == 0 && !half) a.pout[(size_t)group * 256 + p * 16 + cidx] = (p > cidx) ? 0.0 : b[0];
  } else {
    gh = xk_tile_opaque(gh);                                 // (or 22 pointers stay live across the 16 steps)
#pragma unroll
    for (int r = 0; r < RH; ++r) {
      double *q = gh + (size_t)r * strip_step;
      bool ok = r < nstrips;
      if (r == 0 && !half) q = g00;
      if (r >= PR && half) { q = gp; ok = r == PR && a.pend; }
      if (ok) xk_store_wt(q, b[r], a.wt);
    }
  }
}
__global__ __launch_bounds__(1024) void xk_caqr_merge32(XkCaqrArgs a) {
  constexpr int NP = 32, RHP = XK_M32_STRIDE(22);
  __shared__ __attribute__((aligned(16))) double ubuf[2 * NP * RHP];
  __shared__ __attribute__((aligned(16))) double sc[2 * 4];
  xk_caqr_first32_body(a, blockIdx.x, blockIdx.y, ubuf, sc);
}

// Overlapped schedule (two merge levels).  The last merge level of panel k only touches the pivot strips of
// the first-level group leaders; every other row is final for panel k once the first level has run.  So ONE
// launch runs the last level of panel k (workgroups [0, n_last)) next to the
// tile step of panel k+1 (the rest of the grid), in which the leaders leave those 16 rows out (their "hole")
// and use the other 16 of their first 32 rows as the pivot strip.  The rows the last level leaves behind join
// the first level of panel k+1 as a 21st (41st) dense strip.  Per panel: 2 dependent launches instead of 3.
template <int RPL, bool CSPLIT>
__global__ __launch_bounds__(RPL == 16 ? 768 : 512) __attribute__((amdgpu_waves_per_eu(XK_TILE_WAVES_PER_EU(RPL)))) void xk_caqr_fused(XkCaqrArgs ta, XkCaqrArgs la, int n_last, int tsplit) {
  constexpr int LDS_T = 2 * 4 * (RPL + 2), LDS_L = (RPL == 16) ? 2 * 32 * (10 + 2) : 2 * 16 * (20 + 2);
  __shared__ __attribute__((aligned(16))) double ubuf[LDS_T > LDS_L ? LDS_T : LDS_L];
  __shared__ __attribute__((aligned(16))) double sc[2 * 4];
  const int id = blockIdx.x;
  if (id < n_last) {
    if (RPL == 16) {                                             // 80-VGPR budget: the 32-lane layout
      if ((int)threadIdx.x >= 32 * (16 + la.chalf)) return;      // whole waves
      xk_caqr_last32_body(la, id, ubuf, sc);
    } else {                                                     // 128-row tiles leave room for the 16-lane one
      if ((int)threadIdx.x >= 16 * (16 + la.chalf)) return;      // whole waves: chalf % 4 == 0
      xk_caqr_merge_body<20>(la, 0, id, ubuf, sc);
    }
  } else {
    const int w = id - n_last;
    xk_caqr_tile_body<RPL, CSPLIT>(ta, CSPLIT ? w / tsplit : w, CSPLIT ? w % tsplit : 0, ubuf, sc);
  }
}
