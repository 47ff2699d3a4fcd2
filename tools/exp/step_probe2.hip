// Per-step cost of the CAQR kernels in core cycles AND in wall time (clock64 vs wall_clock64):
// tells what the shader clock really is while these kernels run back to back.
#define XK_CAQR_PROBE 1
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../../x_multi_agent_amd/csrc/xk_linalg.hip.h"
int main() {
  const int C1 = 181, C1P = 192, nt = 400;
  double *A, *R, *P0, *P1; int *rows; long long *dbg;
  hipMalloc(&A, sizeof(double) * (size_t)nt * 64 * C1P); hipMalloc(&R, sizeof(double) * C1P * C1P);
  hipMalloc(&rows, 4 * nt); hipMalloc(&P0, 8 * 256 * (nt + 2)); hipMalloc(&P1, 8 * 256 * (nt + 2)); hipMemset(P0, 0, 8 * 256 * (nt + 2)); hipMalloc(&dbg, 64);
  std::vector<double> hA((size_t)nt * 64 * C1P);
  for (size_t i = 0; i < hA.size(); ++i) hA[i] = ((i * 2654435761u) % 1000) / 1000.0 - 0.5;
  std::vector<int> hr(nt, 57);
  hipMemcpy(rows, hr.data(), 4 * nt, hipMemcpyHostToDevice);
  hipMemcpy(A, hA.data(), sizeof(double) * hA.size(), hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  int wcr = 0; hipDeviceGetAttribute(&wcr, hipDeviceAttributeWallClockRate, 0);
  int clk = 0; hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
  printf("wall clock rate %d kHz, core clock attr %d kHz\n", wcr, clk);
  for (int c0 : {0, 96, 160}) {
    const int trail = C1 - c0 - 16 > 0 ? C1 - c0 - 16 : 0;
    for (int mode = 0; mode < 3; ++mode) {
      XkCaqrArgs a{A, rows, nt, 64, 64, C1P, C1, c0, 1, 0, R, 8, P0, P1, 0, 0, 0, 0, dbg, 0};
      float ms; long long d[4];
      const int reps = 200;
      int threads, gx, gy = 1;
      if (mode == 0) { threads = (4 * (16 + trail) + 63) / 64 * 64; gx = nt; }
      else { gy = (trail + 7) / 8; if (gy < 1) gy = 1; threads = 16 * 24; a.stride = mode == 1 ? 1 : 20; gx = mode == 1 ? 20 : 1; }
      for (int rep = 0; rep < reps + 20; ++rep) {
        if (rep == 20) hipEventRecord(e0);
        if (mode == 0) { a.chalf = trail; hipLaunchKernelGGL((xk_caqr_tile<16, false>), dim3(gx), dim3(threads), 0, 0, a); a.chalf = 8; }
        else hipLaunchKernelGGL(xk_caqr_merge<20>, dim3(gx, gy), dim3(threads), 0, 0, a);
      }
      hipEventRecord(e1); hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1);
      hipMemcpy(d, dbg, 32, hipMemcpyDeviceToHost);
      const double wall_us = (double)d[2] / (wcr * 1e-3);
      printf("c0=%3d %s grid=%dx%d thr=%d: %.2f us/launch | WG0 %lld steps: %lld core ticks (%.0f/step), %.2f us wall (%.3f us/step) -> %.0f MHz\n", c0,
             mode == 0 ? "tile  " : mode == 1 ? "stripL1" : "stripL2", gx, gy, threads, 1e3 * ms / reps, d[3], d[1], (double)d[1] / d[3], wall_us, wall_us / d[3], d[1] / wall_us);
    }
  }
  printf("%s\n", hipGetErrorString(hipGetLastError()));
}
