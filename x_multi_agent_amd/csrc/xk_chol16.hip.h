// xk_chol16.hip.h -- the 16 x 16 pivot chain shared by the Kalman-stage Cholesky (xk_chol_whole) and the
// chi-square gate of the feature kernel.
#pragma once
#include <hip/hip_runtime.h>

typedef double xk_d2 __attribute__((ext_vector_type(2)));
typedef double xk_d4 __attribute__((ext_vector_type(4)));

// Cholesky factor of one 16 x 16 block and the inverse of the factor, in ONE wave and without LDS traffic
// or scalar broadcasts inside the pivot chain.  Lane t (every row of 16 lanes runs the same thing) keeps row t
// of the block (v) and COLUMN t of L^-1 (w).  Step k needs L(j,k) of every other row j -- for the block,
// A(t,j) -= L(t,k) L(j,k), and for the inverse, W(j,t) -= L(j,k) W(k,t) -- which is one lane's value of ONE
// register: exactly what the 64-bit DPP control row_newbcast:j delivers inside v_fmac_f64.  A step is then
// the pivot's rsqrt chain plus 2 (15 - k) multiply-adds, ~200 clocks instead of ~700 through LDS.
template <int LANE>
__device__ __forceinline__ double xk_fmac_bcast(double acc, double src, double mul) {
  asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(mul), "n"(LANE));
  return acc;
}
template <int LANE>
__device__ __forceinline__ double xk_mov_bcast(double src) {
  double r;
  // (a DPP read needs two wait states after the VALU write of its source)
  asm("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(src), "n"(LANE));
  return r;
}
template <int K>
__device__ __forceinline__ void xk_chol16_step(double (&v)[16], double (&w)[16], bool &bad) {
  const double piv = xk_mov_bcast<K>(v[K]);
  if (!(piv > 0.0)) bad = true;
  double inv = __builtin_amdgcn_rsq(piv);
  const double h = -0.5 * piv;
  inv = inv * fma(h * inv, inv, 1.5);
  inv = inv * fma(h * inv, inv, 1.5);
  const double lik = v[K] * inv;          // L(t,K); rows t < K carry dead values
  double lneg;
  asm("v_mul_f64 %0, %1, %2\n\ts_nop 1" : "=v"(lneg) : "v"(v[K]), "v"(-inv));
  const double wk = w[K] * inv;           // W(K,t), final
  w[K] = wk;
#define XK_CJ(J) if (J > K) { v[J] = xk_fmac_bcast<J>(v[J], lneg, lik); w[J] = xk_fmac_bcast<J>(w[J], lneg, wk); }
  XK_CJ(1) XK_CJ(2) XK_CJ(3) XK_CJ(4) XK_CJ(5) XK_CJ(6) XK_CJ(7) XK_CJ(8)
  XK_CJ(9) XK_CJ(10) XK_CJ(11) XK_CJ(12) XK_CJ(13) XK_CJ(14) XK_CJ(15)
#undef XK_CJ
}
// blk: the block, row-major 16 x 16 (LDS);  linv: L^-1, rows padded to 17 (LDS).  Returns true on a bad pivot.
__device__ __forceinline__ bool xk_chol16_bcast(const double *blk, double *linv, int lane) {
  const int tt = lane & 15;
  double v[16], w[16];
#pragma unroll
  for (int q = 0; q < 16; q += 2) {
    const xk_d2 p = *reinterpret_cast<const xk_d2 *>(&blk[16 * tt + q]);
    v[q] = p[0]; v[q + 1] = p[1];
    w[q] = (tt == q) ? 1.0 : 0.0; w[q + 1] = (tt == q + 1) ? 1.0 : 0.0;
  }
  bool bad = false;
  xk_chol16_step<0>(v, w, bad); xk_chol16_step<1>(v, w, bad); xk_chol16_step<2>(v, w, bad); xk_chol16_step<3>(v, w, bad);
  xk_chol16_step<4>(v, w, bad); xk_chol16_step<5>(v, w, bad); xk_chol16_step<6>(v, w, bad); xk_chol16_step<7>(v, w, bad);
  xk_chol16_step<8>(v, w, bad); xk_chol16_step<9>(v, w, bad); xk_chol16_step<10>(v, w, bad); xk_chol16_step<11>(v, w, bad);
  xk_chol16_step<12>(v, w, bad); xk_chol16_step<13>(v, w, bad); xk_chol16_step<14>(v, w, bad); xk_chol16_step<15>(v, w, bad);
  if (lane < 16) {
#pragma unroll
    for (int q = 0; q < 16; ++q) linv[q * 17 + tt] = w[q];
  }
  return bad;
}

