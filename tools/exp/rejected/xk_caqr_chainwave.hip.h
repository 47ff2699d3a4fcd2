// xk_caqr_chainwave.hip.h -- EXPERIMENT, not part of libxk.so unless built with -DXK_RES_CHAINWAVE=1.
// Measured on MI355X (tools/exp/ab_libs.sh, same box): QR 0.441 ms against 0.410 ms for the one-barrier look-ahead steps --
// the tile phase of panels 1..3 goes from 10.5 to 16.6 us (1.04 us per step): two workgroup barriers per step and a
// 64-lane reduction in the chain wave cost more than taking norm + scalar chain off the owner's wave saves.  Parity
// identical (tests/test_gpu_resident_caqr.py passes with this build).
#pragma once
// The tile step with a CHAIN WAVE.  In every step the owner of the pivot column only
// publishes it (masked as in xk_caqr_form) plus the pivot entry; an otherwise idle wave of the workgroup (wave 0 from
// panel 1 on: its columns are finished) reads the column back, forms the norm and runs the scalar chain
// (|beta|, v_pivot, tt) WHILE every other wave already forms its dot products with the raw column -- the dot of a column
// is  P + v_pivot B  with P = sum over the rows below the pivot and B = the column's pivot-row entry, both available
// before v_pivot is.  Two barriers per step, but norm -> chain no longer sits between a wave's apply and the next
// owner's publish.  Same reflectors; the dot is summed in a different order (P and v_pivot B separately).
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ double xk_res_wave_sum(double v) {           // over all 64 lanes, result in every lane
  v = xk_group_sum<16>(v);
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

// ucol: [4][RPL + 2] the published column;  sc: [0] -tt  [1] v_pivot  [2] beta  [3] the raw pivot entry
template <int KK, int RPL>
__device__ __forceinline__ void xk_cw_step(double (&b)[RPL], int rel, bool live, int part, bool chain_wave, int lane,
                                           double *ucol, double *sc) {
  constexpr int RPLP = RPL + 2;
  xk_d2 *useg = reinterpret_cast<xk_d2 *>(ucol + part * RPLP);
  if (rel == KK) {                                               // phase 1: the owner publishes
    const double below = (part != 0) ? 1.0 : 0.0;
#pragma unroll
    for (int r = 0; r < RPL; r += 2) {
      xk_d2 tt = {(r > KK) ? b[r] : b[r] * below, (r + 1 > KK) ? b[r + 1] : b[r + 1] * below};
      useg[r >> 1] = tt;
    }
    if (part == 0) sc[3] = b[KK];
  }
  __syncthreads();                                               // ---- barrier A
  xk_d2 u[RPL / 2];
  double P = 0.0, B = 0.0;
  const bool cons = rel > KK && live;
  if (chain_wave) {                                              // phase 2, chain wave: norm of the column + scalar chain
    const int i0 = lane, i1 = lane + 64;                         // entries of the 4 x RPL column (RPL = 24: 96 of them)
    const double v0 = ucol[(i0 / RPL) * RPLP + i0 % RPL];
    const double v1 = (i1 < 4 * RPL) ? ucol[(i1 / RPL) * RPLP + i1 % RPL] : 0.0;
    const double tail = xk_res_wave_sum(fma(v0, v0, v1 * v1));
    if (lane == 0) {
      const double c0v = sc[3];
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
      sc[0] = -(y2 * rt); sc[1] = vp; sc[2] = beta;
    }
  } else if (cons) {                                             // phase 2, everyone to the right: dots with the raw column
#pragma unroll
    for (int r = 0; r < RPL / 2; ++r) u[r] = useg[r];
    double d0 = 0.0, d1 = 0.0, d2 = 0.0, d3 = 0.0;
#pragma unroll
    for (int r = 0; r < RPL / 2; ++r) {
      if (r & 1) { d2 = fma(u[r][0], b[2 * r], d2); d3 = fma(u[r][1], b[2 * r + 1], d3); }
      else { d0 = fma(u[r][0], b[2 * r], d0); d1 = fma(u[r][1], b[2 * r + 1], d1); }
    }
    P = xk_group_sum<4>((d0 + d1) + (d2 + d3));
    B = xk_dpp_quad<0x00>(b[KK]);                                // the column's pivot-row entry: lane `part 0` of the quad
  }
  __syncthreads();                                               // ---- barrier B
  if (cons) {
    const double mtt = sc[0];
    if (mtt != 0.0) {
      const double vp = sc[1];
      const double w = mtt * fma(vp, B, P);
#pragma unroll
      for (int r = 0; r < RPL / 2; ++r) {
        b[2 * r] = fma(w, u[r][0], b[2 * r]);
        b[2 * r + 1] = fma(w, u[r][1], b[2 * r + 1]);
      }
      if (part == 0) b[KK] = fma(w, vp, b[KK]);
    }
  } else if (rel == KK && part == 0) {
    b[KK] = sc[2];
  }
}

template <int RPL>
__device__ __forceinline__ void xk_cw_steps(double (&b)[RPL], int rel, bool live, int part, int nsteps, bool chain_wave, int lane,
                                            double *ucol, double *sc) {
#define XK_STEP(K) if (K < nsteps) xk_cw_step<K, RPL>(b, rel, live, part, chain_wave, lane, ucol, sc);
  XK_STEP(0) XK_STEP(1) XK_STEP(2) XK_STEP(3) XK_STEP(4) XK_STEP(5) XK_STEP(6) XK_STEP(7)
  XK_STEP(8) XK_STEP(9) XK_STEP(10) XK_STEP(11) XK_STEP(12) XK_STEP(13) XK_STEP(14) XK_STEP(15)
#undef XK_STEP
}

