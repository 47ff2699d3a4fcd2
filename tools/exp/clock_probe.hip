// What does clock64() (s_memtime) count, and how fast can waves issue?
//   * 4096 s_nop wait states and a chain of 1024 dependent v_fma_f64, timed with clock64() and wall_clock64()
//   * 2048 INDEPENDENT v_fma_f64 per wave with 1, 2 and 4 waves per SIMD: does a SIMD issue one wave64 VALU
//     instruction per 4 clocks in total, or per wave?
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(long long *out, double x) {
  long long c0 = clock64(), w0 = wall_clock64();
#pragma unroll
  for (int i = 0; i < 256; ++i) asm volatile("s_nop 15");
  long long c1 = clock64(), w1 = wall_clock64();
  double y = x;
#pragma unroll
  for (int i = 0; i < 1024; ++i) asm volatile("v_fma_f64 %0, %0, %0, %1" : "+v"(y) : "v"(x));
  long long c2 = clock64(), w2 = wall_clock64();
  long long c3 = clock64();
  for (int rep = 0; rep < 64; ++rep) {
#pragma unroll
    for (int i = 0; i < 256; ++i) asm volatile("s_nop 15");
  }
  long long c4 = clock64(), w4 = wall_clock64();
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    out[0] = c1 - c0; out[1] = w1 - w0; out[2] = c2 - c1; out[3] = w2 - w1; out[4] = c4 - c3; out[5] = w4 - w2; out[6] = (long long)y;
  }
}
__global__ void issue(long long *out, double x) {
  double a[8];
  for (int i = 0; i < 8; ++i) a[i] = x + i;
  __syncthreads();
  long long c0 = clock64();
#pragma unroll
  for (int i = 0; i < 256; ++i) {   // 8 independent chains: 2048 FMAs, never waiting on a result
    asm volatile("v_fma_f64 %0, %0, %8, %8\n v_fma_f64 %1, %1, %8, %8\n v_fma_f64 %2, %2, %8, %8\n v_fma_f64 %3, %3, %8, %8\n"
                 "v_fma_f64 %4, %4, %8, %8\n v_fma_f64 %5, %5, %8, %8\n v_fma_f64 %6, %6, %8, %8\n v_fma_f64 %7, %7, %8, %8"
                 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(x));
  }
  long long c1 = clock64();
  double s = 0; for (int i = 0; i < 8; ++i) s += a[i];
  if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) { out[2 + 2 * (threadIdx.x >> 6)] = c0; out[3 + 2 * (threadIdx.x >> 6)] = c1; }
  if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = c1 - c0; out[1] = (long long)s; }
}
int main() {
  long long *d, h[40];
  hipMalloc(&d, 320);
  for (int grid : {1, 256}) {
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(probe, dim3(grid), dim3(256), 0, 0, d, 0.5);
    hipDeviceSynchronize();
    hipMemcpy(h, d, 64, hipMemcpyDeviceToHost);
    printf("grid %4d: 4096 s_nop states: %lld ticks | 1024 dependent v_fma_f64: %lld ticks (%.2f/op) | 262144 s_nop states: %lld ticks, %.2f us wall -> tick rate %.0f MHz\n",
           grid, h[0], h[2], h[2] / 1024.0, h[4], h[5] / 100.0, h[4] / (h[5] / 100.0));
  }
  for (int threads : {256, 512, 1024}) {   // 1, 2, 4 waves per SIMD on one CU
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(issue, dim3(1), dim3(threads), 0, 0, d, 0.5);
    hipDeviceSynchronize();
    hipMemcpy(h, d, 320, hipMemcpyDeviceToHost);
    long long lo = h[2], hi = h[3];
    for (int w = 0; w < threads / 64; ++w) { if (h[2 + 2 * w] < lo) lo = h[2 + 2 * w]; if (h[3 + 2 * w] > hi) hi = h[3 + 2 * w]; }
    printf("%d waves/SIMD: 2048 independent v_fma_f64 per wave; oldest wave %lld ticks, all waves %lld ticks -> %.2f ticks per wave64 instruction per SIMD\n",
           threads / 256, h[0], hi - lo, (double)(hi - lo) / (2048.0 * (threads / 256)));
  }
}
