// Clocks of the 16 x 16 factor-and-invert pivot chain (xk_chol16_bcast) on one wave, and its result against the host.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/exp/chol16_probe.hip -o tools/exp/bin/chol16_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <cstring>
#include <vector>
#include "../../x_multi_agent_amd/csrc/xk_chol16.hip.h"
__global__ void k(const double *A, double *W, long long *clk, int reps) {
  __shared__ double blk[256], linv[16 * 17];
  const int lane = threadIdx.x;
  for (int i = lane; i < 256; i += 64) blk[i] = A[i];
  __syncthreads();
  bool bad = false;
  const long long t0 = clock64();
  for (int r = 0; r < reps; ++r) {
    bad |= xk_chol16_bcast(blk, linv, lane);
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
  }
  const long long t1 = clock64();
  __syncthreads();
  for (int i = lane; i < 256; i += 64) W[i] = linv[(i / 16) * 17 + (i % 16)];
  if (lane == 0) { clk[0] = t1 - t0; clk[1] = bad; }
}
int main() {
  std::vector<double> A(256), L(256, 0.0), Wh(256, 0.0), W(256);
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) A[16 * i + j] = (i == j ? 6.0 + 0.3 * i : 0.0) + 0.4 * std::cos(0.7 * (i + 1) * (j + 1)) * std::cos(0.7 * (i + 1) * (j + 1) + 0.0) * 0 + 0.25 * std::sin(0.9 * i) * std::sin(0.9 * j) + 0.2 * std::cos(1.3 * i) * std::cos(1.3 * j);
  // host Cholesky + inverse of the factor
  for (int j = 0; j < 16; ++j) {
    double s = A[16 * j + j];
    for (int k = 0; k < j; ++k) s -= L[16 * j + k] * L[16 * j + k];
    L[16 * j + j] = std::sqrt(s);
    for (int i = j + 1; i < 16; ++i) {
      double t = A[16 * i + j];
      for (int k = 0; k < j; ++k) t -= L[16 * i + k] * L[16 * j + k];
      L[16 * i + j] = t / L[16 * j + j];
    }
  }
  for (int c = 0; c < 16; ++c)
    for (int i = 0; i < 16; ++i) {
      double t = (i == c) ? 1.0 : 0.0;
      for (int k = 0; k < i; ++k) t -= L[16 * i + k] * Wh[16 * k + c];
      Wh[16 * i + c] = t / L[16 * i + i];
    }
  double *dA, *dW; long long *dc;
  hipMalloc(&dA, 2048); hipMalloc(&dW, 2048); hipMalloc(&dc, 16);
  hipMemcpy(dA, A.data(), 2048, hipMemcpyHostToDevice);
  const int reps = 200;
  for (int it = 0; it < 3; ++it) hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dW, dc, reps);
  long long c[2];
  hipMemcpy(c, dc, 16, hipMemcpyDeviceToHost);
  hipMemcpy(W.data(), dW, 2048, hipMemcpyDeviceToHost);
  double err = 0, nrm = 0;
  for (int i = 0; i < 256; ++i) { err = std::fmax(err, std::fabs(W[i] - Wh[i])); nrm = std::fmax(nrm, std::fabs(Wh[i])); }
  unsigned long long h = 1469598103934665603ull;
  for (int i = 0; i < 256; ++i) { unsigned long long b; memcpy(&b, &W[i], 8); h = (h ^ b) * 1099511628211ull; }
  printf("chol16: %.0f clocks per block (shader clock), bad=%lld, max |W - W_host| = %.3e (|W| max %.3f), hash %016llx  %s\n", (double)c[0] / reps, c[1], err, nrm, h,
         hipGetErrorString(hipGetLastError()));
}
