// Micro-probe: cycles per Householder step of the [R;B] update (column per thread,
// B in registers, R packed in LDS), measured with s_memtime inside the kernel and
// HIP events outside.  Build: hipcc --offload-arch=gfx950 -O3 -o qr_step_probe qr_step_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../../x_multi_agent_amd/csrc/xk_linalg.hip.h"

template <int NB, bool RLDS>
__global__ __launch_bounds__(768) void probe(double *Rg, const double *A, int C1, int C1P, int passes, long long *cyc) {
  constexpr int SPLIT = 4, RPL = NB / SPLIT;
  __shared__ __attribute__((aligned(16))) double vbuf[XK_QR_VBUF(NB, SPLIT)];
  __shared__ double sc[4];
  const int j = threadIdx.x / SPLIT, part = threadIdx.x % SPLIT;
  double b[RPL];
  XkRLds rl{xk_dyn_lds, C1};
  XkRGlb rg{Rg + (size_t)blockIdx.x * C1P * C1P, C1P};
  long long t0 = clock64();
  for (int p = 0; p < passes; ++p) {
#pragma unroll
    for (int r = 0; r < RPL; ++r) b[r] = (j < C1) ? A[(size_t)(p * NB + part * RPL + r) * C1P + j] : 0.0;
    if (RLDS) xk_qr_pass<NB, SPLIT>(b, rl, C1, 0, p == 0, vbuf, sc);
    else xk_qr_pass<NB, SPLIT>(b, rg, C1, 0, p == 0, vbuf, sc);
  }
  long long t1 = clock64();
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
  if (RLDS && j < C1 && part == 0) Rg[(size_t)blockIdx.x * C1P * C1P + j] = rl.ld(0, j);
}

int main() {
  const int C1 = 181, C1P = 192, NB = 64, passes = 4;
  double *R, *A; long long *cyc;
  const int maxg = 256;
  hipMalloc(&R, sizeof(double) * (size_t)maxg * C1P * C1P);
  hipMalloc(&A, sizeof(double) * (size_t)passes * NB * C1P);
  hipMalloc(&cyc, sizeof(long long) * maxg);
  std::vector<double> hA((size_t)passes * NB * C1P);
  for (size_t i = 0; i < hA.size(); ++i) hA[i] = ((i * 2654435761u) % 1000) / 1000.0 - 0.5;
  hipMemcpy(A, hA.data(), sizeof(double) * hA.size(), hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const size_t lds = sizeof(double) * ((size_t)C1 * (C1 + 1) / 2 + 2);
  hipFuncSetAttribute((const void *)probe<NB, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 2048);
  for (int rl = 0; rl < 2; ++rl)
    for (int grid : {1, 8, 64, 256}) {
      float best = 1e9;
      long long c = 0;
      for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        if (rl) hipLaunchKernelGGL((probe<NB, true>), dim3(grid), dim3(768), lds, 0, R, A, C1, C1P, passes, cyc);
        else hipLaunchKernelGGL((probe<NB, false>), dim3(grid), dim3(768), 0, 0, R, A, C1, C1P, passes, cyc);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
        hipMemcpy(&c, cyc, sizeof(c), hipMemcpyDeviceToHost);
      }
      const double steps = (double)passes * C1;
      printf("rlds=%d grid=%3d  %.3f ms  -> %.3f us/step, %.0f memtime-ticks/step (tick=%.2f ns => %.2f GHz if tick=cycle)\n", rl, grid, best,
             1e3 * best / steps, c / steps, 1e6 * best / c, c / (1e6 * best));
    }
  printf("err=%s\n", hipGetErrorString(hipGetLastError()));
  return 0;
}
