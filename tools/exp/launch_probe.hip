// Where does a strip-merge launch spend its time?  Wall-clock (100 MHz) stamps of workgroup (0,0):
// entry, after the loads, after the 16 steps, after the stores -- for a train of dependent launches.
#define XK_CAQR_PROBE 1
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../../x_multi_agent_amd/csrc/xk_linalg.hip.h"
int main() {
  const int C1 = 181, C1P = 192, nt = 400, NL = 24;
  double *A, *R, *P0, *P1; int *rows; long long *dbg;
  hipMalloc(&A, sizeof(double) * (size_t)nt * 64 * C1P); hipMalloc(&R, sizeof(double) * C1P * C1P);
  hipMalloc(&rows, 4 * nt); hipMalloc(&P0, 8 * 256 * (nt + 2)); hipMalloc(&P1, 8 * 256 * (nt + 2)); hipMemset(P0, 0, 8 * 256 * (nt + 2)); hipMalloc(&dbg, 64 * NL);
  std::vector<double> hA((size_t)nt * 64 * C1P);
  for (size_t i = 0; i < hA.size(); ++i) hA[i] = ((i * 2654435761u) % 1000) / 1000.0 - 0.5;
  std::vector<int> hr(nt, 57);
  hipMemcpy(rows, hr.data(), 4 * nt, hipMemcpyHostToDevice);
  hipMemcpy(A, hA.data(), sizeof(double) * hA.size(), hipMemcpyHostToDevice);
  for (int c0 : {0, 96}) {
    const int trail = C1 - c0 - 16;
    for (int rep = 0; rep < 2; ++rep) {
      for (int i = 0; i < NL; ++i) {
        XkCaqrArgs a{A, rows, nt, 64, 64, C1P, C1, c0, (i & 1) ? 20 : 1, 0, R, 8, P0, P1, 0, 0, 0, 0, dbg + 8 * i, 0};
        hipLaunchKernelGGL(xk_caqr_merge<20>, dim3((i & 1) ? 1 : 20, (trail + 7) / 8), dim3(384), 0, 0, a);
      }
      hipDeviceSynchronize();
    }
    std::vector<long long> d(8 * NL);
    hipMemcpy(d.data(), dbg, 64 * NL, hipMemcpyDeviceToHost);
    printf("c0=%d (alternating L1 grid 20x%d / L2 grid 1x%d); times in us, WG(0,0)\n", c0, (trail + 7) / 8, (trail + 7) / 8);
    for (int i = 1; i < NL; ++i) {
      const long long *p = &d[8 * (i - 1)], *q = &d[8 * i];
      printf("  launch %2d %s: prev-exit->entry %5.2f | load %5.2f | steps %5.2f | store %5.2f | entry->entry %5.2f\n", i, (i & 1) ? "L2" : "L1",
             (q[4] - p[7]) / 100.0, (q[5] - q[4]) / 100.0, (q[6] - q[5]) / 100.0, (q[7] - q[6]) / 100.0, (q[4] - p[4]) / 100.0);
    }
  }
  printf("%s\n", hipGetErrorString(hipGetLastError()));
}
