// What does clock64() (s_memtime) count?  Time a block of 4096 s_nop wait states (= 4096 shader cycles)
// and a chain of 1024 dependent v_fma_f64 with clock64() and wall_clock64() (100 MHz).
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
  // long spin so the wall clock has resolution
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
int main() {
  long long *d, h[8];
  hipMalloc(&d, 64);
  for (int grid : {1, 256, 2048}) {
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(probe, dim3(grid), dim3(256), 0, 0, d, 0.5);
    hipDeviceSynchronize();
    hipMemcpy(h, d, 64, hipMemcpyDeviceToHost);
    printf("grid %4d: 4096 s_nop states: %lld ticks | 1024 dependent v_fma_f64: %lld ticks (%.2f/op) | 262144 s_nop states: %lld ticks, %.2f us wall -> s_nop rate %.0f MHz, tick rate %.0f MHz\n",
           grid, h[0], h[2], h[2] / 1024.0, h[4], h[5] / 100.0, 262144.0 / (h[5] / 100.0), h[4] / (h[5] / 100.0));
  }
}
