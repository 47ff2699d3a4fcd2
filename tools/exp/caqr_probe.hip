// Phase timing of xk_caqr_panel (load / 16 steps) with s_memtime, tile and strip mode.
#define XK_CAQR_PROBE 1
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../../x_multi_agent_amd/csrc/xk_linalg.hip.h"
int main() {
  const int C1 = 181, C1P = 192, nt = 328;
  double *A, *R; int *rows, *list, *ntl; long long *dbg;
  hipMalloc(&A, sizeof(double) * (size_t)nt * 64 * C1P); hipMalloc(&R, sizeof(double) * C1P * C1P);
  hipMalloc(&rows, 4 * nt); hipMalloc(&list, 4 * (nt + 8)); hipMalloc(&ntl, 16); hipMalloc(&dbg, 64);
  std::vector<double> hA((size_t)nt * 64 * C1P);
  for (size_t i = 0; i < hA.size(); ++i) hA[i] = ((i * 2654435761u) % 1000) / 1000.0 - 0.5;
  std::vector<int> hr(nt, 57);
  hipMemcpy(rows, hr.data(), 4 * nt, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int c0 : {0, 96}) {
    hipMemcpy(A, hA.data(), sizeof(double) * hA.size(), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(xk_compact_tiles, dim3(1), dim3(1024), 0, 0, rows, nt, list, ntl);
    XkCaqrArgs a{A, rows, list, ntl, C1P, C1, c0, 1, 0, R, 1, C1 - c0 - 16, dbg};
    const int threads = (4 * (C1 - c0) + 63) / 64 * 64;
    for (int mode = 0; mode < 4; ++mode) {
      long long d[4];
      float ms;
      int grid = mode == 0 ? nt : mode == 1 ? 41 : mode == 2 ? 6 : 1;
      a.stride = mode <= 1 ? 1 : mode == 2 ? 8 : 64;
      a.final_level = mode == 3;
      for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        if (mode == 0) hipLaunchKernelGGL((xk_caqr_panel<16, false>), dim3(grid), dim3(threads), 0, 0, a);
        else hipLaunchKernelGGL((xk_caqr_panel<32, true>), dim3(grid), dim3(threads), 0, 0, a);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
      }
      hipMemcpy(d, dbg, 32, hipMemcpyDeviceToHost);
      printf("c0=%3d mode=%d grid=%3d threads=%d: %.1f us | WG0: load %lld ticks, %lld steps %lld ticks (%.0f/step)\n", c0, mode, grid,
             threads, 1e3 * ms, d[0], d[3], d[1], (double)d[1] / d[3]);
    }
  }
  // MFMA kernels
  hipFree(dbg); hipMalloc(&dbg, 128);
  for (int c0 : {0, 96}) {
    hipMemcpy(A, hA.data(), sizeof(double) * hA.size(), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(xk_compact_tiles, dim3(1), dim3(1024), 0, 0, rows, nt, list, ntl);
    XkCaqrArgs a{A, rows, list, ntl, C1P, C1, c0, 1, 0, R, 1, C1 - c0 - 16, dbg};
    const int threads = 64 * (1 + (C1 - c0 - 16 + 15) / 16);
    for (int mode = 0; mode < 2; ++mode) {
      long long d[16]; float ms;
      int grid = mode == 0 ? nt : 41;
      for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        if (mode == 0) hipLaunchKernelGGL((xk_caqr_mfma<4, false>), dim3(grid), dim3(threads), 0, 0, a);
        else hipLaunchKernelGGL((xk_caqr_mfma<8, true>), dim3(grid), dim3(threads), 0, 0, a);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
      }
      hipMemcpy(d, dbg, 128, hipMemcpyDeviceToHost);
      printf("MFMA c0=%3d mode=%d threads=%d: %.1f us | w0: load %lld, +steps %lld, +T %lld | w1: at-barrier %lld, released %lld, mfma-done %lld\n",
             c0, mode, threads, 1e3 * ms, d[4], d[5], d[6], d[7], d[8], d[9]);
    }
  }
  printf("%s\n", hipGetErrorString(hipGetLastError()));
}
