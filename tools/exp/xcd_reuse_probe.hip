// Does a consumer kernel read a producer kernel's output faster when both workgroups sit on the same XCD
// (workgroup id mod 8), i.e. does the XCD's L2 keep serving the lines across a kernel boundary?
// Producer: workgroup w writes slab w (92 KB, the size of one 64 x 181 tile).  Consumer: workgroup w reads slab
// (w + shift) % n and reduces it.  shift = 0 / 8: same XCD (same / other CU); shift = 1: the neighbouring XCD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
constexpr int SLAB = 64 * 184;   // doubles
__global__ __launch_bounds__(768) void produce(double *buf, double v) {
  double *s = buf + (size_t)blockIdx.x * SLAB;
  for (int i = threadIdx.x; i < SLAB; i += blockDim.x) s[i] = v + i;
}
__global__ __launch_bounds__(768) void consume(const double *buf, double *out, int shift, int n) {
  const double *s = buf + (size_t)((blockIdx.x + shift) % n) * SLAB;
  double acc = 0;
  for (int i = threadIdx.x; i < SLAB; i += blockDim.x) acc += s[i];
  if (acc == 12345.678) out[blockIdx.x] = acc;
}
int main() {
  const int n = 400;
  double *buf, *out;
  hipMalloc(&buf, sizeof(double) * (size_t)n * SLAB); hipMalloc(&out, 8 * n);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int shift : {0, 8, 1, 3, 0, 1}) {
    float best = 1e9, sum = 0;
    for (int rep = 0; rep < 20; ++rep) {
      hipLaunchKernelGGL(produce, dim3(n), dim3(768), 0, 0, buf, (double)rep);
      hipEventRecord(e0);
      hipLaunchKernelGGL(consume, dim3(n), dim3(768), 0, 0, buf, out, shift, n);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (rep >= 2) { sum += ms; if (ms < best) best = ms; }
    }
    printf("shift %d: consumer launch %.2f us min, %.2f us mean\n", shift, 1e3 * best, 1e3 * sum / 18);
  }
  printf("%s\n", hipGetErrorString(hipGetLastError()));
}
