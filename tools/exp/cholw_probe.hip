// Phase timing of xk_chol_whole (single-launch blocked Cholesky with the right-hand sides carried along):
// per-wave clock stamps of one block step, and a check of L^-1 [W | z] against a host factorisation.
#ifndef XK_CHOLW_PROBE
#define XK_CHOLW_PROBE 1
#endif
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>
#include "../../x_multi_agent_amd/csrc/xk_linalg.hip.h"
int main() {
  const int c = 180, n = 195, ncols = c + n + 1, ld = 416;
  std::vector<double> M((size_t)c * ld, 0.0);
  for (int i = 0; i < c; ++i) for (int j = 0; j < ncols; ++j) {
    double v = std::sin(0.37 * i + 0.11 * j) * 0.1;
    if (j < c) v = (i == j) ? 8.0 : 0.02 * std::cos(0.3 * (i + j));
    M[(size_t)i * ld + j] = v;
  }
  double *dM, *dX; int *st; long long *dbg;
  hipMalloc(&dM, 8 * M.size()); hipMalloc(&dX, 8 * M.size()); hipMalloc(&st, 4); hipMalloc(&dbg, 8 * 128);
  hipMemset(st, 0, 4); hipMemset(dbg, 0, 8 * 128);
  hipMemcpy(dM, M.data(), 8 * M.size(), hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  XkCholWholeArgs a;
  a.Maug = dM; a.ld = ld; a.c = c; a.ncols = ncols; a.X = dX; a.status = st; a.dbg = dbg;
  xk_cholw_table((c + 15) / 16, a.tab);
  float ms = 0, best = 1e9;
  for (int rep = 0; rep < 10; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(xk_chol_whole, dim3((n + 1 + 15) / 16), dim3(64 * XK_CHOLW_WAVES), 0, 0, a);
    hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  long long d[128]; hipMemcpy(d, dbg, 8 * 128, hipMemcpyDeviceToHost);
  printf("launch %.1f us; block step %d, clocks relative to wave 0's step start\n", 1e3 * best, XK_CHOLW_PROBE);
  printf("wave  start  factored  past-bar1  B-done  past-bar2  C-done   (wave 9 = factor wave)\n");
  for (int w = 0; w < 10; ++w) {
    printf("%4d", w);
    for (int i = 0; i < 6; ++i) printf(" %9lld", d[w * 8 + i] ? d[w * 8 + i] - d[0] : -1LL);
    printf("\n");
  }
  // host check
  std::vector<double> L((size_t)c * c, 0.0), X((size_t)c * ld);
  hipMemcpy(X.data(), dX, 8 * X.size(), hipMemcpyDeviceToHost);
  for (int i = 0; i < c; ++i) for (int j = 0; j <= i; ++j) {
    double s = M[(size_t)i * ld + j];
    for (int k = 0; k < j; ++k) s -= L[i * c + k] * L[j * c + k];
    L[i * c + j] = (i == j) ? std::sqrt(s) : s / L[j * c + j];
  }
  double worst = 0;
  for (int col = c; col < ncols; ++col) {
    std::vector<double> y(c);
    for (int i = 0; i < c; ++i) {
      double s = M[(size_t)i * ld + col];
      for (int k = 0; k < i; ++k) s -= L[i * c + k] * y[k];
      y[i] = s / L[i * c + i];
      worst = std::fmax(worst, std::fabs(y[i] - X[(size_t)i * ld + col]));
    }
  }
  printf("max |X - host| = %.3e   (%s)\n", worst, hipGetErrorString(hipGetLastError()));
}
