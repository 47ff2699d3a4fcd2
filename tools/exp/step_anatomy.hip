// Anatomy of ONE Householder step of the 20-way merge (16 lanes/column, 20 rows/lane, 24 columns,
// 384 threads): s_memtime stamps inside step 8 for the owner lane and for a trailing-column lane.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../../x_multi_agent_amd/csrc/xk_linalg.hip.h"

#define STAMP(i) do { if (KK == 8) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); st[i] = clock64(); } } while (0)

template <int KK>
__device__ __forceinline__ void step(double (&b)[20], int cidx, int part, double *ubuf, double *sc, long long *st, int variant) {
  constexpr int NP = 16, RPL = 20, RPLP = 22, pb = KK & 1;
  xk_d2 *useg = reinterpret_cast<xk_d2 *>(ubuf + (pb * NP + part) * RPLP);
  double *scp = sc + pb * 4;
  STAMP(0);
  if (cidx == KK) {
#pragma unroll
    for (int r = 0; r < RPL; r += 2) { xk_d2 tt = {b[r], b[r + 1]}; useg[r >> 1] = tt; }
    STAMP(1);
    double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
#pragma unroll
    for (int r = 0; r < RPL; ++r) {
      const double x = ((part != 0) || (r > KK)) ? b[r] : 0.0;
      if ((r & 3) == 0) s0 = fma(x, x, s0); else if ((r & 3) == 1) s1 = fma(x, x, s1);
      else if ((r & 3) == 2) s2 = fma(x, x, s2); else s3 = fma(x, x, s3);
    }
    double tail = (s0 + s1) + (s2 + s3);
    STAMP(2);
    tail = xk_group_sum<NP>(tail);
    STAMP(3);
    if (part == 0) {
      const double c0v = b[KK];
      double y2 = 0.0, tden = 1.0, vp = 0.0, beta = c0v;
      if (tail > 2.2250738585072014e-308) {
        const double n2 = fma(c0v, c0v, tail);
        double y = __builtin_amdgcn_rsq(n2);
        y = y * fma(-0.5 * n2 * y, y, 1.5);
        y = y * fma(-0.5 * n2 * y, y, 1.5);
        const double ab = n2 * y;
        beta = (c0v >= 0) ? -ab : ab;
        vp = c0v - beta; y2 = y * y; tden = fma(fabs(c0v), y, 1.0);
      }
      xk_d2 s01 = {y2, tden};
      *reinterpret_cast<xk_d2 *>(scp) = s01;
      scp[2] = vp;
      b[KK] = beta;
    }
#pragma unroll
    for (int r = 0; r < RPL; ++r) if ((part != 0) || (r > KK)) b[r] = 0.0;
    STAMP(4);
  }
  STAMP(5);
  __syncthreads();
  STAMP(6);
  const xk_d2 s01 = *reinterpret_cast<const xk_d2 *>(scp);
  const double vp = scp[2];
  xk_d2 u[RPL / 2];
#pragma unroll
  for (int r = 0; r < RPL / 2; ++r) u[r] = useg[r];
  STAMP(7);
  if (cidx > KK && s01[0] != 0.0) {
    double rt = __builtin_amdgcn_rcp(s01[1]);
    rt = fma(rt, fma(-s01[1], rt, 1.0), rt);
    rt = fma(rt, fma(-s01[1], rt, 1.0), rt);
    if (part == 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) { if (r < KK) u[r >> 1][r & 1] = 0.0; else if (r == KK) u[r >> 1][r & 1] = vp; }
    }
    double d0 = 0, d1 = 0, d2 = 0, d3 = 0;
#pragma unroll
    for (int r = 0; r < RPL / 2; ++r) {
      if (r & 1) { d2 = fma(u[r][0], b[2 * r], d2); d3 = fma(u[r][1], b[2 * r + 1], d3); }
      else { d0 = fma(u[r][0], b[2 * r], d0); d1 = fma(u[r][1], b[2 * r + 1], d1); }
    }
    double dd = (d0 + d1) + (d2 + d3);
    STAMP(8);
    dd = xk_group_sum<NP>(dd);
    STAMP(9);
    const double w = -(s01[0] * rt) * dd;
#pragma unroll
    for (int r = 0; r < RPL / 2; ++r) { b[2 * r] = fma(w, u[r][0], b[2 * r]); b[2 * r + 1] = fma(w, u[r][1], b[2 * r + 1]); }
    STAMP(10);
  }
}

__global__ __launch_bounds__(1024) void anatomy(const double *A, double *out, long long *dbg, int ncols) {
  __shared__ __attribute__((aligned(16))) double ubuf[2 * 16 * 22];
  __shared__ __attribute__((aligned(16))) double sc[8];
  const int cidx = threadIdx.x / 16, part = threadIdx.x & 15;
  double b[20];
  for (int r = 0; r < 20; ++r) b[r] = A[((size_t)blockIdx.x * 320 + part * 20 + r) * 192 + cidx];
  long long st[12] = {0};
  const long long t0 = clock64();
#define S(K) step<K>(b, cidx, part, ubuf, sc, st, 0);
  S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7) S(8) S(9) S(10) S(11) S(12) S(13) S(14) S(15)
  const long long t1 = clock64();
  double acc = 0; for (int r = 0; r < 20; ++r) acc += b[r];
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = acc;
  if (blockIdx.x == 0 && part == 0 && (cidx == 8 || cidx == ncols - 1 || cidx == 9)) {
    long long *d = dbg + (cidx == 8 ? 0 : cidx == 9 ? 16 : 32);
    for (int i = 0; i < 11; ++i) d[i] = st[i] - st[0];
    d[11] = t1 - t0;
  }
}

int main() {
  const int nb = 512;
  double *A, *out; long long *dbg;
  hipMalloc(&A, sizeof(double) * (size_t)nb * 320 * 192); hipMalloc(&out, 8 * nb * 1024); hipMalloc(&dbg, 8 * 48);
  std::vector<double> hA((size_t)nb * 320 * 192);
  for (size_t i = 0; i < hA.size(); ++i) hA[i] = ((i * 2654435761u) % 1000) / 1000.0 - 0.5;
  hipMemcpy(A, hA.data(), sizeof(double) * hA.size(), hipMemcpyHostToDevice);
  const char *names[] = {"step entry", "owner: u written", "owner: norm partials", "owner: norm reduced", "owner: scalars written", "at barrier", "past barrier",
                         "u + scalars read", "dot partials", "dot reduced", "updated"};
  for (int grid : {1, 256, 512}) for (int ncols : {24, 48}) {
    hipMemset(dbg, 0, 8 * 48);
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(anatomy, dim3(grid), dim3(16 * ncols), 0, 0, A, out, dbg, ncols);
    hipDeviceSynchronize();
    long long d[48]; hipMemcpy(d, dbg, 8 * 48, hipMemcpyDeviceToHost);
    printf("grid=%d threads=%d  (16 steps: owner-lane %lld, next-col %lld, last-col %lld cycles)\n", grid, 16 * ncols, d[11], d[16 + 11], d[32 + 11]);
    printf("  %-24s %8s %8s %8s\n", "cycles since step entry", "owner", "col 9", "last col");
    for (int i = 1; i < 11; ++i) printf("  %-24s %8lld %8lld %8lld\n", names[i], d[i], d[16 + i], d[32 + i]);
  }
  printf("%s\n", hipGetErrorString(hipGetLastError()));
}
