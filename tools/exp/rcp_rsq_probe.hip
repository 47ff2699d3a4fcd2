// Accuracy of v_rcp_f64 / v_rsq_f64 on gfx950, raw and after one / two Newton steps (relative error in units of 2^-52),
// over 4M logarithmically spaced arguments.  hipcc --offload-arch=gfx950 -O3 -o tools/exp/bin/rcp_rsq_probe tools/exp/rcp_rsq_probe.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
__global__ void probe(double *out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double x = exp2(-40.0 + 80.0 * (double)i / n) * (1.0 + 0.37 * ((i * 2654435761u) & 1023) / 1024.0);
  // exact references in double-double would be better; use long-double-free trick: error via fma residuals
  double r0 = __builtin_amdgcn_rcp(x);
  double r1 = fma(r0, fma(-x, r0, 1.0), r0);
  double r2 = fma(r1, fma(-x, r1, 1.0), r1);
  // residual e = 1 - x r (exact with fma) ~ relative error of r
  out[6 * (size_t)i + 0] = fabs(fma(-x, r0, 1.0));
  out[6 * (size_t)i + 1] = fabs(fma(-x, r1, 1.0));
  out[6 * (size_t)i + 2] = fabs(fma(-x, r2, 1.0));
  double y0 = __builtin_amdgcn_rsq(x);
  double y1 = y0 * fma(-0.5 * x * y0, y0, 1.5);
  double y2 = y1 * fma(-0.5 * x * y1, y1, 1.5);
  // residual 1 - x y^2 ~ 2 x relative error of y
  auto res = [&](double y) { const double xy = x * y; const double lo = fma(x, y, -xy); return fabs(fma(-xy, y, 1.0) - lo * y) * 0.5; };
  out[6 * (size_t)i + 3] = res(y0);
  out[6 * (size_t)i + 4] = res(y1);
  out[6 * (size_t)i + 5] = res(y2);
}
int main() {
  const int n = 1 << 22;
  double *d; hipMalloc(&d, sizeof(double) * 6 * n);
  probe<<<(n + 255) / 256, 256>>>(d, n);
  double *h = (double *)malloc(sizeof(double) * 6 * n);
  hipMemcpy(h, d, sizeof(double) * 6 * n, hipMemcpyDeviceToHost);
  double mx[6] = {0, 0, 0, 0, 0, 0};
  for (int i = 0; i < n; ++i) for (int k = 0; k < 6; ++k) if (h[6 * (size_t)i + k] > mx[k]) mx[k] = h[6 * (size_t)i + k];
  const double u = ldexp(1.0, -52);
  printf("rcp: raw %.3g  one step %.3g  two steps %.3g   (x 2^-52)\n", mx[0] / u, mx[1] / u, mx[2] / u);
  printf("rsq: raw %.3g  one step %.3g  two steps %.3g   (x 2^-52)\n", mx[3] / u, mx[4] / u, mx[5] / u);
  return 0;
}
