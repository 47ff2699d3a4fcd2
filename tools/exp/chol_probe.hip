// Phase timing of xk_chol_step (one 32-column block step of the blocked Cholesky), last workgroup.
#define XK_CHOL_PROBE 1
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>
#include "../../x_multi_agent_amd/csrc/xk_linalg.hip.h"
int main() {
  const int c = 181, n = 195, ncols = c + n + 1, ld = 384;
  std::vector<double> M((size_t)c * ld, 0.0);
  for (int i = 0; i < c; ++i) for (int j = 0; j < ncols; ++j) {
    double v = std::sin(0.37 * i + 0.11 * j) * 0.1;
    if (j < c) v = (i == j) ? 8.0 : 0.02 * std::cos(0.3 * (i + j));
    M[(size_t)i * ld + j] = v;
  }
  double *dM, *dX; int *st; long long *dbg;
  hipMalloc(&dM, 8 * M.size()); hipMalloc(&dX, 8 * M.size()); hipMalloc(&st, 4); hipMalloc(&dbg, 64);
  hipMemset(st, 0, 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int kb : {0, 96}) {
    const int nb = 32, rest = ncols - kb - nb, mrem = c - kb - nb;
    XkCholStepArgs a{dM, ld, kb, nb, c, ncols, dX, (rest + 15) / 16, st, dbg};
    float ms = 0;
    for (int rep = 0; rep < 5; ++rep) {
      hipMemcpy(dM, M.data(), 8 * M.size(), hipMemcpyHostToDevice);
      hipEventRecord(e0);
      hipLaunchKernelGGL(xk_chol_step, dim3(a.ncb * (1 + (mrem + 15) / 16)), dim3(64), 0, 0, a);
      hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    }
    long long d[8]; hipMemcpy(d, dbg, 64, hipMemcpyDeviceToHost);
    printf("kb=%d grid=%d: %.1f us | cycles: load %lld, factor+inverse %lld (%.0f/step), Ls write %lld, X_j tile %lld, X_i + product + store %lld\n", kb,
           a.ncb * (1 + (mrem + 15) / 16), 1e3 * ms, d[1] - d[0], d[2] - d[1], (d[2] - d[1]) / 32.0, d[3] - d[2], d[4] - d[3], d[5] - d[4]);
  }
  printf("%s\n", hipGetErrorString(hipGetLastError()));
}
