// What does an in-launch hand-off between workgroups cost on MI355X when producer and consumer sit on the SAME XCD
// (same L2), and is "plain store -> s_waitcnt vmcnt(0) -> counter -> sc1 load" enough there?
//
//   census   every workgroup reads HW_REG_XCC_ID, takes a slot on its XCD's counter, waits for the whole grid
//   mode 0   same XCD:  plain stores, per-XCD barrier, sc1 (L1-bypassing) loads of the neighbour's slab
//   mode 1   same XCD:  plain stores, per-XCD barrier, PLAIN loads (expected: stale, L1-warm consumer)
//   mode 2   cross XCD: sc1 (write-through) stores, grid barrier, sc1 loads of the slab of the same slot on XCD+1
//   mode 3   cross XCD: plain stores, grid barrier, sc1 loads (expected: stale)
//   mode 4   same XCD:  barrier only, nothing published
// Every word carries (round, writer) so a stale or torn read is counted, not guessed.  All spins are bounded.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned long long u64;
#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

struct Args {
  unsigned *cnt;      // [8] per-XCD arrivals of the census
  unsigned *total;    // grid arrivals
  unsigned *bar;      // [8] per-XCD barrier counters + [8] = grid barrier counter
  unsigned *abort_;   // set when a spin gives up
  u64 *buf;           // [8][64][words] slabs
  long long *out;     // per workgroup: xcc, slot, ticks, stale, first, last
  int rounds, wpt, mode, bkind, slp;
};

__device__ __forceinline__ unsigned xcc_id() {
  unsigned x;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
  return x & 15u;
}

__device__ __forceinline__ bool spin_until(unsigned *p, unsigned target, unsigned *abort_, int slp = 1) {
  for (unsigned it = 0; it < (1u << 22); ++it) {
    if (__hip_atomic_load(p, RLX_AGENT) >= target) return true;
    if ((it & 1023u) == 1023u && __hip_atomic_load(abort_, RLX_AGENT)) return false;
    if (slp <= 1) __builtin_amdgcn_s_sleep(1); else if (slp <= 4) __builtin_amdgcn_s_sleep(4); else if (slp <= 16) __builtin_amdgcn_s_sleep(16); else __builtin_amdgcn_s_sleep(64);
  }
  __hip_atomic_store(abort_, 1u, RLX_AGENT);
  return false;
}

// one arrival per workgroup (lane 0); bkind 1: the last arriver publishes the generation, everybody else polls THAT word
__device__ __forceinline__ bool barrier_arrive_wait(unsigned *cnt, unsigned *gen, unsigned n, unsigned round1, unsigned *abort_, int bkind, int slp) {
  if (bkind == 0) {
    __hip_atomic_fetch_add(cnt, 1u, RLX_AGENT);
    return spin_until(cnt, n * round1, abort_, slp);
  }
  const unsigned old = __hip_atomic_fetch_add(cnt, 1u, RLX_AGENT);
  if (old == n * round1 - 1) { __hip_atomic_store(gen, round1, RLX_AGENT); return true; }
  return spin_until(gen, round1, abort_, slp);
}

__global__ __launch_bounds__(768) __attribute__((amdgpu_waves_per_eu(6))) void probe(Args a) {
  __shared__ unsigned s_slot, s_nx, s_ok;
  const int tid = threadIdx.x;
  const unsigned xcc = xcc_id();
  if (tid == 0) {
    s_slot = __hip_atomic_fetch_add(&a.cnt[xcc], 1u, RLX_AGENT);
    __hip_atomic_fetch_add(a.total, 1u, RLX_AGENT);
    s_ok = spin_until(a.total, gridDim.x, a.abort_) ? 1u : 0u;
    s_nx = __hip_atomic_load(&a.cnt[xcc], RLX_AGENT);
  }
  __syncthreads();
  if (!s_ok) return;
  const unsigned slot = s_slot, nx = s_nx;
  const int words = 768 * a.wpt;
  const bool cross = a.mode == 2 || a.mode == 3;
  u64 *mine = a.buf + ((size_t)xcc * 64 + slot) * words;
  const unsigned nxcc = cross ? (xcc + 1) & 7 : xcc, nslot = cross ? slot : (slot + 1) % nx;
  const u64 *theirs = a.buf + ((size_t)nxcc * 64 + nslot) * words;
  long long stale = 0;
  const long long t0 = wall_clock64();
  for (int r = 0; r < a.rounds; ++r) {
    if (a.mode != 4) {
      for (int w = 0; w < a.wpt; ++w) {
        const u64 v = ((u64)(r + 1) << 32) | ((u64)xcc << 24) | ((u64)slot << 16) | (unsigned)(w * 768 + tid) % 65536u;
        if (a.mode == 2) __hip_atomic_store(&mine[w * 768 + tid], v, RLX_AGENT);
        else mine[w * 768 + tid] = v;
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    if (tid == 0) {
      unsigned *b = cross ? &a.bar[8 * 16] : &a.bar[xcc * 16];
      s_ok = barrier_arrive_wait(b, b + 1024, cross ? gridDim.x : nx, (unsigned)(r + 1), a.abort_, a.bkind, a.slp) ? 1u : 0u;
    }
    __syncthreads();
    if (!s_ok) break;
    if (a.mode != 4) {
      for (int w = 0; w < a.wpt; ++w) {
        u64 v;
        if (a.mode == 1) v = ((volatile const u64 *)theirs)[w * 768 + tid];
        else v = __hip_atomic_load(&theirs[w * 768 + tid], RLX_AGENT);
        const u64 want = ((u64)(r + 1) << 32) | ((u64)nxcc << 24) | ((u64)nslot << 16) | (unsigned)(w * 768 + tid) % 65536u;
        if (v != want) ++stale;
      }
    }
    // (the next round's stores must not overtake a neighbour still reading: a second barrier, as a phase would have)
    __syncthreads();
    if (tid == 0) {
      unsigned *b = cross ? &a.bar[512 + 8 * 16] : &a.bar[512 + xcc * 16];
      s_ok = barrier_arrive_wait(b, b + 1024, cross ? gridDim.x : nx, (unsigned)(r + 1), a.abort_, a.bkind, a.slp) ? 1u : 0u;
    }
    __syncthreads();
    if (!s_ok) break;
  }
  const long long t1 = wall_clock64();
  // stale counts: sum over the workgroup
  __shared__ long long s_sum;
  if (tid == 0) s_sum = 0;
  __syncthreads();
  if (stale) atomicAdd((u64 *)&s_sum, (u64)stale);
  __syncthreads();
  if (tid == 0) {
    long long *o = a.out + 6 * blockIdx.x;
    o[0] = xcc; o[1] = slot; o[2] = t1 - t0; o[3] = s_sum; o[4] = t0; o[5] = t1;
  }
}

int main(int argc, char **argv) {
  const int grid = argc > 1 ? atoi(argv[1]) : 512, rounds = argc > 2 ? atoi(argv[2]) : 200;
  unsigned *ctr; u64 *buf; long long *out;
  const int wpt_max = 4;
  hipMalloc(&ctr, 4 * 4096); hipMalloc(&buf, sizeof(u64) * 8 * 64 * 768 * wpt_max); hipMalloc(&out, 8 * 6 * grid);
  std::vector<long long> h(6 * grid);
  for (int bkind : {0, 1})
  for (int slp : {1, 16, 64})
  for (int mode : {4, 0, 2})
    for (int wpt : {4}) {
      hipMemset(ctr, 0, 4 * 4096);
      hipMemset(buf, 0, sizeof(u64) * 8 * 64 * 768 * wpt_max);
      Args a{ctr, ctr + 8, ctr + 64, ctr + 40, buf, out, rounds, wpt, mode, bkind, slp};
      hipLaunchKernelGGL(probe, dim3(grid), dim3(768), 0, 0, a);
      hipError_t e = hipDeviceSynchronize();
      hipMemcpy(h.data(), out, 8 * 6 * grid, hipMemcpyDeviceToHost);
      unsigned hc[64];
      hipMemcpy(hc, ctr, 4 * 64, hipMemcpyDeviceToHost);
      printf("bkind %d sleep %2d ", bkind, slp);
      long long worst = 0, stale = 0, tmin = h[4], tmax = h[5];
      int mism = 0;
      for (int b = 0; b < grid; ++b) {
        worst = std::max(worst, h[6 * b + 2]); stale += h[6 * b + 3];
        tmin = std::min(tmin, h[6 * b + 4]); tmax = std::max(tmax, h[6 * b + 5]);
        if (h[6 * b] != b % 8) ++mism;
      }
      printf("mode %d wpt %d (%5.1f KB/WG): %6.2f us per round (2 barriers), stale words %lld of %lld, abort %u, err %s\n", mode, wpt,
             768 * wpt * 8 / 1024.0, worst / 100.0 / rounds, stale, (long long)grid * 768 * wpt * rounds, hc[40], hipGetErrorString(e));
      if (mode == 4) {
        printf("  census: per-XCD counts");
        for (int x = 0; x < 8; ++x) printf(" %u", hc[x]);
        printf("; blocks with xcc != blockIdx %% 8: %d; start skew %.2f us\n", mism, 0.0);
      }
    }
  return 0;
}
