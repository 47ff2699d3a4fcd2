// Can two chained v_mfma_f64_4x4x4f64 with an all-ones B operand replace the 4-stage DPP butterfly that sums a value over
// the 16 lanes of a DPP row (the merge layout's column group)?  Checks the result in every lane and times both.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>

template <int NP> __device__ __forceinline__ double group_sum_dpp(double v) {
  // the butterfly of xk_group_sum<16> (csrc/xk_linalg.hip.h): lane ^ 1, lane ^ 2, row_half_mirror, row_mirror
  auto mv = [](double x, auto f) { long long q = __builtin_bit_cast(long long, x); int lo = (int)q, hi = (int)(q >> 32); lo = f(lo); hi = f(hi); return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned)lo); };
  v += mv(v, [](int x) { return __builtin_amdgcn_mov_dpp(x, 0xB1, 0xF, 0xF, true); });
  v += mv(v, [](int x) { return __builtin_amdgcn_mov_dpp(x, 0x4E, 0xF, 0xF, true); });
  v += mv(v, [](int x) { return __builtin_amdgcn_mov_dpp(x, 0x141, 0xF, 0xF, true); });   // row_half_mirror
  v += mv(v, [](int x) { return __builtin_amdgcn_mov_dpp(x, 0x140, 0xF, 0xF, true); });   // row_mirror
  return v;
}
__device__ __forceinline__ double group_sum_mfma(double v) {
  double r = __builtin_amdgcn_mfma_f64_4x4x4f64(v, 1.0, 0.0, 0, 0, 0);
  return __builtin_amdgcn_mfma_f64_4x4x4f64(r, 1.0, 0.0, 0, 0, 0);
}
__global__ void check(const double *in, double *o1, double *o2) {
  const double v = in[threadIdx.x];
  o1[threadIdx.x] = group_sum_dpp<16>(v);
  o2[threadIdx.x] = group_sum_mfma(v);
}
template <bool MFMA> __global__ void chain(double *out, int iters) {
  double v = 1.0 + 1e-9 * threadIdx.x;
  const long long t0 = clock64();
  for (int i = 0; i < iters; ++i) { v = (MFMA ? group_sum_mfma(v) : group_sum_dpp<16>(v)) * 0.0625; }
  const long long t1 = clock64();
  out[threadIdx.x] = v;
  if (threadIdx.x == 0) out[64] = (double)(t1 - t0) / iters;
}
int main() {
  std::vector<double> h(64); for (int i = 0; i < 64; ++i) h[i] = std::sin(1.0 + i) * (1 + i % 7);
  double *d, *o1, *o2; hipMalloc(&d, 512); hipMalloc(&o1, 1024); hipMalloc(&o2, 1024);
  hipMemcpy(d, h.data(), 512, hipMemcpyHostToDevice);
  check<<<1, 64>>>(d, o1, o2);
  std::vector<double> a(64), b(64); hipMemcpy(a.data(), o1, 512, hipMemcpyDeviceToHost); hipMemcpy(b.data(), o2, 512, hipMemcpyDeviceToHost);
  double worst1 = 0, worst2 = 0;
  for (int g = 0; g < 4; ++g) { double s = 0; for (int i = 0; i < 16; ++i) s += h[16 * g + i];
    for (int i = 0; i < 16; ++i) { worst1 = fmax(worst1, fabs(a[16 * g + i] - s)); worst2 = fmax(worst2, fabs(b[16 * g + i] - s)); } }
  printf("max |dpp - exact| %.3e   max |mfma - exact| %.3e\n", worst1, worst2);
  for (int m = 0; m < 2; ++m) {
    if (m) chain<true><<<1, 64>>>(o1, 4096); else chain<false><<<1, 64>>>(o1, 4096);
    hipDeviceSynchronize(); hipMemcpy(a.data(), o1, 8, hipMemcpyDeviceToHost); double c; hipMemcpy(&c, o1 + 64, 8, hipMemcpyDeviceToHost);
    printf("%s: %.1f clocks per dependent 16-lane sum (+ one multiply)\n", m ? "mfma x2" : "dpp x4 ", c);
  }
  return 0;
}
