// xk_caqr_blocked.hip.h -- the per-tile panel step of the register-resident CAQR as a BLOCKED Householder step:
//   * the 16 panel columns are factored by ONE wave with no LDS traffic and no barrier inside the chain: lane (part, c)
//     of the wave holds rows 4 q + part (q = 0..23) of column c, so the reflector entries a consumer needs from the
//     owner column are one lane of its OWN 16-lane row -- the 64-bit DPP control row_newbcast delivers them inside
//     v_fmac_f64 (the trick of xk_chol16.hip.h) -- and sums over a column's 4 lanes are two cross-row lane swaps;
//   * the 16 reflectors are applied to the trailing columns in compact-WY form, C <- C - V T^T (V^T C), on
//     v_mfma_f64_16x16x4_f64 by the other waves: with rows interleaved over the parts (row = 4 q + part) a wave's
//     register q IS the B operand of the q-th K-step of V^T C, and registers 4Q..4Q+3 ARE the C/D tile of rows
//     16Q..16Q+15 of C - V Z, so no element moves between lanes; V and T^T travel through LDS once per panel;
//   * T comes from the Gram matrix: T^-1 = striu(V^T V) + diag(V^T V) / 2 (24 MFMAs in the panel wave), inverted by
//     16 lanes with a 16-step back substitution while the trailing waves already compute V^T C.
// Two workgroup barriers per panel instead of sixteen.  The arithmetic is the same Householder QR (same reflectors,
// same R up to rounding order in the trailing update).
#pragma once
#include <hip/hip_runtime.h>

#include "xk_chol16.hip.h"

#define XK_BLK_LDV 17                       // padded row of V / T^T / M in LDS (doubles)
#define XK_BLK_LDS (96 * XK_BLK_LDV + 2 * 16 * XK_BLK_LDV + 16)

// sum over the four rows of the wave (the four lanes that share a column)
__device__ __forceinline__ double xk_parts_sum(double v) {
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

// One Householder step of the panel wave.  b[q] = rows 4 q + part of this lane's column cl; pivot row KK lives in
// register KK >> 2 of the lanes with part == (KK & 3).  Same reflector as xk_caqr_step: H = I - tt v v^T,
// v = [c0 - beta; x_below], tt = y^2 / (1 + |c0| y).
template <int KK>
__device__ __forceinline__ void xk_blk_pstep(double (&b)[24], int cl, int part, double &vpsave, double *tts) {
  constexpr int q0 = KK >> 2, p0 = KK & 3;
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll
  for (int q = q0 + 1; q < 24; ++q) {
    if ((q & 3) == 0) s0 = fma(b[q], b[q], s0); else if ((q & 3) == 1) s1 = fma(b[q], b[q], s1);
    else if ((q & 3) == 2) s2 = fma(b[q], b[q], s2); else s3 = fma(b[q], b[q], s3);
  }
  const double own = b[q0];
  s0 = fma(own * ((part > p0) ? 1.0 : 0.0), own, s0);
  const double tail = xk_parts_sum((s0 + s1) + (s2 + s3));      // rows below the pivot, per column
  const double c0v = xk_parts_sum((part == p0) ? own : 0.0);     // the pivot entry, in all four lanes of the column
  double y2 = 0.0, tden = 1.0, vp = 0.0, beta = c0v;
  if (tail > 2.2250738585072014e-308) {
    const double n2 = fma(c0v, c0v, tail);
    double y = __builtin_amdgcn_rsq(n2);
    y = y * fma(-0.5 * n2 * y, y, 1.5);
    y = y * fma(-0.5 * n2 * y, y, 1.5);
    const double ab = n2 * y;
    beta = (c0v >= 0) ? -ab : ab;
    vp = c0v - beta;
    y2 = y * y;
    tden = fma(fabs(c0v), y, 1.0);
  }
  double rt = __builtin_amdgcn_rcp(tden);
  rt = fma(rt, fma(-tden, rt, 1.0), rt);
  rt = fma(rt, fma(-tden, rt, 1.0), rt);
  const double mtt = -(y2 * rt);
  // the reflector's entries in register q0: nothing above the pivot, vp at it, the column itself below
  const double ue = (part < p0) ? 0.0 : (part == p0) ? vp : own;
  if (cl == KK && part == p0) { b[q0] = beta; vpsave = vp; tts[KK] = -mtt; }
  const double ueb = xk_mov_bcast<KK>(ue), mttb = xk_mov_bcast<KK>(mtt);
  double d0 = ueb * own, d1 = 0.0, d2 = 0.0, d3 = 0.0;
#pragma unroll
  for (int q = q0 + 1; q < 24; ++q) {
    if ((q & 3) == 0) d0 = xk_fmac_bcast<KK>(d0, b[q], b[q]); else if ((q & 3) == 1) d1 = xk_fmac_bcast<KK>(d1, b[q], b[q]);
    else if ((q & 3) == 2) d2 = xk_fmac_bcast<KK>(d2, b[q], b[q]); else d3 = xk_fmac_bcast<KK>(d3, b[q], b[q]);
  }
  const double dd = xk_parts_sum((d0 + d1) + (d2 + d3));
  const double w = (cl > KK) ? mttb * dd : 0.0;                  // finished and owner columns are left alone
  b[q0] = fma(w, ueb, b[q0]);
#pragma unroll
  for (int q = q0 + 1; q < 24; ++q) b[q] = xk_fmac_bcast<KK>(b[q], b[q], w);
}

// The panel wave: 16 (or nsteps) reflector steps on its 16 columns, then V -> LDS (row-major, padded), the Gram matrix
// on the matrix cores, M = T^-1 -> LDS, and T^T = (M^-1)^T -> LDS by 16 lanes.  `lds` is XK_BLK_LDS doubles.
// Barrier protocol (all waves of the workgroup): [barrier A after V is in LDS] ... [barrier B after T^T is in LDS].
__device__ __forceinline__ void xk_blk_panel_wave(double (&b)[24], int lane, int nsteps, double *lds, long long *dbg = nullptr) {
  double *Vs = lds, *Ms = lds + 96 * XK_BLK_LDV, *Tts = Ms + 16 * XK_BLK_LDV, *tts = Tts + 16 * XK_BLK_LDV;
  const int cl = lane & 15, part = lane >> 4;
  double vpsave = 0.0;
  if (lane < 16) tts[lane] = 0.0;
#define XK_PS(K) if (K < nsteps) xk_blk_pstep<K>(b, cl, part, vpsave, tts);
  XK_PS(0) XK_PS(1) XK_PS(2) XK_PS(3) XK_PS(4) XK_PS(5) XK_PS(6) XK_PS(7)
  XK_PS(8) XK_PS(9) XK_PS(10) XK_PS(11) XK_PS(12) XK_PS(13) XK_PS(14) XK_PS(15)
#undef XK_PS
  if (dbg && lane == 0) dbg[0] = wall_clock64();
  // cleaned reflectors: rows below the diagonal as they stand, vp on it, zero above; an identity step (tt = 0) has none
  const double act = (tts[cl] != 0.0) ? 1.0 : 0.0;
  xk_d4 G = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int q = 0; q < 24; ++q) {
    const int row = 4 * q + part;
    double v = b[q];
    if (q < 4) v = (row > cl) ? v : (row == cl) ? vpsave : 0.0;
    v *= act;
    Vs[row * XK_BLK_LDV + cl] = v;
    G = __builtin_amdgcn_mfma_f64_16x16x4f64(v, v, G, 0, 0, 0);    // V^T V: this register is both operands of K-step q
  }
  // M = T^-1 = striu(G) + diag(G) / 2   (C/D layout: row = part + 4 r, column = cl)
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = part + 4 * r;
    Ms[row * XK_BLK_LDV + cl] = (row < cl) ? G[r] : (row == cl) ? 0.5 * G[r] : 0.0;
  }
  __syncthreads();                                               // ---- barrier A: V (and M, for my own lanes) in LDS
  if (dbg && lane == 0) dbg[1] = wall_clock64();
  // T = M^-1 by back substitution, column j on lane j; 1 / M_ii = tt_i (0 for an identity step).  T^T row j = that column.
  if (lane < 16) {
    const int j = lane;
    double x[16];
#pragma unroll
    for (int i = 15; i >= 0; --i) {
      double a0 = (i == j) ? 1.0 : 0.0, a1 = 0.0;
#pragma unroll
      for (int m = i + 1; m < 16; ++m) {
        if (m & 1) a1 = fma(-Ms[i * XK_BLK_LDV + m], x[m], a1); else a0 = fma(-Ms[i * XK_BLK_LDV + m], x[m], a0);
      }
      x[i] = tts[i] * (a0 + a1);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) Tts[j * XK_BLK_LDV + i] = x[i];
  }
  __syncthreads();                                               // ---- barrier B: T^T in LDS
  if (dbg && lane == 0) dbg[2] = wall_clock64();
}

// A trailing wave: C <- C - V T^T (V^T C) on its 16 columns, all on the matrix cores.
__device__ __forceinline__ void xk_blk_trailing_wave(double (&b)[24], int lane, const double *lds) {
  const double *Vs = lds, *Tts = lds + 96 * XK_BLK_LDV + 16 * XK_BLK_LDV;
  const int li = lane & 15, lk = lane >> 4;
  __syncthreads();                                               // ---- barrier A
  xk_d4 W = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int q = 0; q < 24; ++q)                                   // W = V^T C: A[i = reflector li][k] = V[4 q + lk][li]
    W = __builtin_amdgcn_mfma_f64_16x16x4f64(Vs[(4 * q + lk) * XK_BLK_LDV + li], b[q], W, 0, 0, 0);
  __syncthreads();                                               // ---- barrier B
  xk_d4 Z = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int kk = 0; kk < 4; ++kk)                                 // Z = T^T W: A[i][k] = T^T[li][4 kk + lk]
    Z = __builtin_amdgcn_mfma_f64_16x16x4f64(Tts[li * XK_BLK_LDV + 4 * kk + lk], W[kk], Z, 0, 0, 0);
#pragma unroll
  for (int Q = 0; Q < 6; ++Q) {                                  // rows 16 Q .. 16 Q + 15 of C - V Z
    xk_d4 acc = {b[4 * Q], b[4 * Q + 1], b[4 * Q + 2], b[4 * Q + 3]};
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-Vs[(16 * Q + li) * XK_BLK_LDV + 4 * kk + lk], Z[kk], acc, 0, 0, 0);
    b[4 * Q] = acc[0]; b[4 * Q + 1] = acc[1]; b[4 * Q + 2] = acc[2]; b[4 * Q + 3] = acc[3];
  }
}
