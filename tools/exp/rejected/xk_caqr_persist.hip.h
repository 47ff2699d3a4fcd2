// xk_caqr_persist.hip.h -- the whole QR compression (VioUpdater::applyQRDecomposition, src/x/vio/vio_updater.cpp:487-512)
// in ONE launch: the CAQR schedule of xk_linalg.hip.h with its 24 kernel boundaries replaced by XCD-local hand-offs.
//
// Why: a dependent launch costs ~8 us on this path before any arithmetic happens (dispatch gap, cold first loads after
// the L2 invalidate, write-back drain; tools/exp/launch_probe.hip), against ~7 us for the 16 reflector steps it carries.
// MI355X has 8 XCDs with one L2 each; L2s are not coherent with each other, but workgroups of ONE XCD can hand data to
// each other through their L2 with plain stores + s_waitcnt vmcnt(0) on the producer and L1-bypassing (sc1) loads on the
// consumer -- measured 2.8 us per hand-off including 24 KB published and 24 KB read per workgroup, 0 stale words in 3e8
// (tools/exp/xcd_sync_probe.hip).  So:
//
//   * every workgroup reads HW_REG_XCC_ID and takes a slot on its XCD's counter (placement is MEASURED, not assumed);
//     XCD x owns tiles [x TPX, (x+1) TPX) and the G first-level groups cut from them, and everything up to and including
//     the first merge level stays inside that XCD: tile step -> XCD barrier -> first-level merge -> XCD barrier;
//   * the last merge level (the 8 G group roots) is the only cross-XCD step.  It runs on workgroups of its own
//     (role L) next to the tile step of the following panel, as in xk_caqr_fused, and exchanges strips through
//     buffers that are written ONLY with write-through (sc1) stores, read ONLY with sc1 loads and never reused inside
//     a launch (one slab per panel), so no L2 ever holds a stale or dirty copy of them;
//   * the root strip a leader tile hands to the last level leaves the tile (X1) and comes back as the pending strip
//     of the next first-level merge (X2); the two 16-row ranges of a leader alternate as in the overlapped schedule.
//
// All spins are bounded: a workgroup that gives up sets the abort word, everybody leaves, status[1] tells the host,
// which falls back to the multi-launch schedule (xk_api.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "xk_linalg.hip.h"

#ifndef XK_SPIN_SLEEP
#define XK_SPIN_SLEEP 8            // s_sleep argument between two polls of a flag (x 64 clocks)
#endif
#define XK_PERSIST_MAXP 32          // panels per launch (C1 <= 512)
#define XK_PERSIST_MAXG 16          // first-level groups = strips of the last level
#define XK_PERSIST_THREADS 768
#define XK_RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

// words of the sync area (each on its own 64-byte line): zeroed by a memset node before every launch
enum {
  XK_PS_CENSUS = 0,                 // [8] workgroups per XCD
  XK_PS_TOTAL = 8,                  // workgroups that have arrived
  XK_PS_ABORT = 9,                  // != 0: somebody gave up (value = reason)
  XK_PS_BARCNT = 16,                // [8] XCD barrier arrivals
  XK_PS_BARGEN = 24,                // [8] XCD barrier generation
  XK_PS_X1CNT = 32,                 // [MAXP] first-level items of panel k that have published their root strip
  XK_PS_L2CNT = 32 + XK_PERSIST_MAXP,   // [MAXP] last-level items of panel k that have published
  XK_PS_X1FLAG = 32 + 2 * XK_PERSIST_MAXP,   // [MAXP] set by the item that completes X1CNT[k]: the word the last level polls
  XK_PS_L2FLAG = 32 + 3 * XK_PERSIST_MAXP,   // [MAXP] likewise for L2CNT[k]: the word the next first-level merges poll
  XK_PS_GBARCNT = 32 + 4 * XK_PERSIST_MAXP,          // [MAXG] per first-level group: barrier arrivals / generation (xk_caqr_resident)
  XK_PS_GBARGEN = 32 + 4 * XK_PERSIST_MAXP + XK_PERSIST_MAXG,
  // xk_caqr_resident publishes the root strips in two halves (rows 0..7 after step 7): X1CNT / X1FLAG count the SECOND half
  XK_PS_X1CNTA = 32 + 4 * XK_PERSIST_MAXP + 2 * XK_PERSIST_MAXG,          // [MAXP] items of panel k whose rows 0..7 are out
  XK_PS_X1FLAGA = 32 + 5 * XK_PERSIST_MAXP + 2 * XK_PERSIST_MAXG,         // [MAXP] the word the last level polls first
  XK_PS_WORDS = 32 + 6 * XK_PERSIST_MAXP + 2 * XK_PERSIST_MAXG
};

// count `add` items in; whoever completes the count raises the flag the consumers poll (a counter that is polled by a
// hundred workgroups while it is still being incremented slows every arrival down: xcd_sync_probe, 5.7 vs 1.4 us)
__device__ __forceinline__ void xk_count_in(unsigned *cnt, unsigned *flag, unsigned add, unsigned need) {
  const unsigned old = __hip_atomic_fetch_add(cnt, add, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (old + add >= need && old < need) __hip_atomic_store(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

struct XkCaqrPersistArgs {
  double *A;              // tiles [ntiles][64][C1P] row-major (in place)
  const int *tile_rows;   // valid rows per tile before panel 0
  int ntiles, rows_max, C1P, C1;
  double *Rout;           // [C1P][C1P] row-major
  double *PB1;            // [8 TPX][16][16] panel blocks of the tile step (stay inside their XCD)
  double *X1;             // [panels][MAXG][16][C1P] group roots after the first level   (sc1 only)
  double *X1P;            // [panels][MAXG][16][16]  their panel blocks                   (sc1 only)
  double *X2;             // [panels][MAXG][16][C1P] what the last level leaves of them   (sc1 only)
  unsigned *sync;         // [XK_PS_WORDS * 16]
  int TPX, G, A1;         // tiles per XCD, groups per XCD, first-level arity (<= 2 RH1 - 1)
  int NT;                 // role-T workgroups per XCD (>= TPX: the extra ones own no tile but take first-level items)
  int lchalf;             // trailing columns per last-level workgroup
  int *status;            // status[1] = reason when the launch gave up
  long long *dbg;         // optional wall-clock stamps of (XCD 0, slot 0) and of last-level workgroup 0
};

__device__ __forceinline__ double xk_ld_sc1(const double *p) {
  return __builtin_bit_cast(double, __hip_atomic_load(reinterpret_cast<const unsigned long long *>(p), XK_RLX_AGENT));
}
__device__ __forceinline__ void xk_st_sc1(double *p, double v) {
  __hip_atomic_store(reinterpret_cast<unsigned long long *>(p), __builtin_bit_cast(unsigned long long, v), XK_RLX_AGENT);
}
__device__ __forceinline__ unsigned xk_xcc_id() {
  unsigned x;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
  return x & 7u;
}

// one lane polls one word (relaxed, L1-bypassing) until it reaches `target`; gives up after XK_SPIN_TICKS (2 ms: the longest
// wait of a healthy launch is the last level's for the first roots, tens of microseconds; 0.2 s -- the first version --
// was six dropped camera frames)
#ifndef XK_SPIN_TICKS
#define XK_SPIN_TICKS 200000LL      // 100 MHz ticks
#endif
__device__ __forceinline__ bool xk_spin_ge(unsigned *p, unsigned target, unsigned *abort_, unsigned reason) {
  const long long t0 = wall_clock64();
  for (unsigned it = 0;; ++it) {
    if (__hip_atomic_load(p, XK_RLX_AGENT) >= target) return true;
    if ((it & 63u) == 63u) {
      if (__hip_atomic_load(abort_, XK_RLX_AGENT)) return false;
      if (wall_clock64() - t0 > XK_SPIN_TICKS) break;
    }
    __builtin_amdgcn_s_sleep(XK_SPIN_SLEEP);
  }
  __hip_atomic_store(abort_, reason, XK_RLX_AGENT);
  return false;
}

// Barrier over the n role-T workgroups of one XCD.  Every wave drains its stores first (plain stores are in the
// XCD's L2 once vmcnt reaches 0); the last arriver publishes the generation, the others poll THAT word, so the
// arrival counter is not hammered by readers (1.4 us against 5.7 us with 64 workgroups polling the counter itself
// while data moves, xcd_sync_probe).
// (cnt_base / gen_base: XK_PS_BARCNT / XK_PS_BARGEN with idx = XCD, or XK_PS_GBARCNT / XK_PS_GBARGEN with idx = group)
__device__ __forceinline__ bool xk_flag_barrier(unsigned *sync, int cnt_base, int gen_base, unsigned idx, unsigned n, unsigned epoch, unsigned *s_ok) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned *cnt = sync + (cnt_base + idx) * 16, *gen = sync + (gen_base + idx) * 16, *ab = sync + XK_PS_ABORT * 16;
    const unsigned old = __hip_atomic_fetch_add(cnt, 1u, XK_RLX_AGENT);
    bool ok = true;
    if (old == n * epoch - 1) __hip_atomic_store(gen, epoch, XK_RLX_AGENT);
    else ok = xk_spin_ge(gen, epoch, ab, 2u);
    *s_ok = ok ? 1u : 0u;
  }
  __syncthreads();
  return *s_ok != 0u;
}
__device__ __forceinline__ bool xk_xcd_barrier(unsigned *sync, unsigned xcc, unsigned n, unsigned epoch, unsigned *s_ok) {
  return xk_flag_barrier(sync, XK_PS_BARCNT, XK_PS_BARGEN, xcc, n, epoch, s_ok);
}

// ---------------------------------------------------------------------------------------------------------------
// (1) per-tile panel step: xk_caqr_tile_body<16, false> with L1-bypassing loads.  A group leader (first tile of its
// first-level group) has a hole from panel 1 on: 16 of its first 32 rows are away at the last level.
// ---------------------------------------------------------------------------------------------------------------
// The phase bodies are real functions (inlined into one loop nest they pushed the kernel to ~400 spilled SGPRs and ~240
// spilled VGPRs); they read the launch arguments straight from the kernarg segment with scalar loads.
typedef const XkCaqrPersistArgs __attribute__((address_space(4))) *XkPersistArgsPtr;
__device__ __forceinline__ XkCaqrPersistArgs xk_persist_args(XkPersistArgsPtr ap) {
  XkCaqrPersistArgs a;
  __builtin_memcpy(&a, (const void *)ap, sizeof(a));
  return a;
}

__device__ __noinline__ void xk_persist_tile(XkPersistArgsPtr ap, int t, int c0, bool holed, int lead_off, double *ubuf, double *sc) {
  constexpr int NP = 4, RPL = 16;
  const XkCaqrPersistArgs a = xk_persist_args(ap);
  const int cidx = (int)threadIdx.x / NP, part = threadIdx.x & (NP - 1);
  const bool panel = cidx < 16;
  const int col = c0 + cidx;
  const bool mine = col < a.C1;
  int prow = part * RPL;
  int rlim = a.rows_max - part * RPL;
  if (holed) {
    if (part == 0) { prow = lead_off; rlim = 16; }
    else if (part == 1) rlim = 0;
  }
  double *rowp = a.A + ((size_t)t * 64 + prow) * a.C1P + col;
  double b[RPL];
#pragma unroll
  for (int r = 0; r < RPL; ++r) b[r] = (mine && r < rlim) ? xk_ld_sc1(rowp + (size_t)r * a.C1P) : 0.0;
  if (c0 == 0) {
    const int nvalid = a.tile_rows[t] - part * RPL;
#pragma unroll
    for (int r = 0; r < RPL; ++r) b[r] = (r < nvalid) ? b[r] : 0.0;
  }
  const int nsteps = (a.C1 - c0 < 16) ? a.C1 - c0 : 16;
  xk_caqr_steps<NP, RPL>(b, cidx, mine, part, nsteps, ubuf, sc);
  if (!mine) return;
  if (panel) {
    if (part == 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) a.PB1[((size_t)t * 16 + r) * 16 + cidx] = (r > cidx) ? 0.0 : b[r];
    }
  } else {
#pragma unroll
    for (int r = 0; r < RPL; ++r)
      if (r < rlim) rowp[(size_t)r * a.C1P] = b[r];
  }
}

// ---------------------------------------------------------------------------------------------------------------
// (2) first merge level of one group, 32 lanes per column: lane (half, p) holds row p of strips half RH .. half RH + RH-1;
// the last slot of half 1 is the pending strip (what the last level left of this group's previous root).  Strip 0 is
// the leader's pivot strip; after the 16 steps it is the group's root and goes to the last level through X1.
// ---------------------------------------------------------------------------------------------------------------
template <int RH>
__device__ __noinline__ void xk_persist_merge1(XkPersistArgsPtr ap, int k, int gid, int base, int nstrips, int split,
                                                  int lead_off, double *ubuf, double *sc) {
  constexpr int NP = 32, PENDSLOT = 2 * RH - 1;
  const XkCaqrPersistArgs a = xk_persist_args(ap);
  const int c0 = 16 * k;
  const int cidx = (int)threadIdx.x / NP, part = threadIdx.x & (NP - 1);
  const int p = part & 15, half = part >> 4;
  const bool panel = cidx < 16;
  const int col = panel ? c0 + cidx : c0 + 16 + split * 8 + (cidx - 16);
  const bool mine = col < a.C1 && (panel || cidx - 16 < 8);
  const bool pend_ok = k > 0 && nstrips > 0;
  const size_t lane_off = panel ? (size_t)p * 16 + cidx : (size_t)p * a.C1P + col;
  const size_t strip_step = panel ? 256 : (size_t)64 * a.C1P;
  double *g0 = panel ? a.PB1 + (size_t)base * 256 + lane_off : a.A + (size_t)base * 64 * a.C1P + lane_off;
  const size_t lead = panel ? 0 : (size_t)lead_off * a.C1P;           // strip 0 = rows lead_off.. of the leader
  const double *pend_src = a.X2 + (((size_t)(k > 0 ? k - 1 : 0) * XK_PERSIST_MAXG + gid) * 16 + p) * a.C1P + col;
  double *pend_dst = a.A + ((size_t)base * 64 + (16 - lead_off) + p) * a.C1P + col;
  double b[RH];
#pragma unroll
  for (int r = 0; r < RH; ++r) {
    const int s = half * RH + r;
    double v = 0.0;
    if (mine) {
      if (RH * half + r == PENDSLOT) { if (half == 1 && pend_ok) v = xk_ld_sc1(pend_src); }
      else if (s < nstrips) v = xk_ld_sc1(g0 + (size_t)s * strip_step + (s == 0 ? lead : 0));
    }
    b[r] = v;
  }
  const int nsteps = (a.C1 - c0 < 16) ? a.C1 - c0 : 16;
#define XK_STEP(K) if (K < nsteps) xk_caqr_mstep32<K, RH>(b, cidx, mine, part, ubuf, sc);
  XK_STEP(0) XK_STEP(1) XK_STEP(2) XK_STEP(3) XK_STEP(4) XK_STEP(5) XK_STEP(6) XK_STEP(7)
  XK_STEP(8) XK_STEP(9) XK_STEP(10) XK_STEP(11) XK_STEP(12) XK_STEP(13) XK_STEP(14) XK_STEP(15)
#undef XK_STEP
  if (!mine) return;
  if (panel) {
    if (split == 0 && half == 0)
      xk_st_sc1(a.X1P + ((size_t)k * XK_PERSIST_MAXG + gid) * 256 + p * 16 + cidx, (p > cidx) ? 0.0 : b[0]);
  } else {
#pragma unroll
    for (int r = 0; r < RH; ++r) {
      const int s = half * RH + r;
      if (s == 0) xk_st_sc1(a.X1 + (((size_t)k * XK_PERSIST_MAXG + gid) * 16 + p) * a.C1P + col, b[r]);
      else if (s == PENDSLOT) { if (pend_ok) pend_dst[0] = b[r]; }
      else if (s < nstrips) g0[(size_t)s * strip_step] = b[r];
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// (3) last merge level: the 8 G group roots of panel k -> the next 16 rows of R; the other strips go back to their
// groups through X2.  Same 32-lane layout with 8 rows per lane.
// ---------------------------------------------------------------------------------------------------------------
__device__ __noinline__ void xk_persist_last(XkPersistArgsPtr ap, int k, int split, double *ubuf, double *sc) {
  constexpr int NP = 32, RH = 8;
  const XkCaqrPersistArgs a = xk_persist_args(ap);
  const int c0 = 16 * k, ngroups = 8 * a.G;
  const int cidx = (int)threadIdx.x / NP, part = threadIdx.x & (NP - 1);
  const int p = part & 15, half = part >> 4;
  const bool panel = cidx < 16;
  const int col = panel ? c0 + cidx : c0 + 16 + split * a.lchalf + (cidx - 16);
  const bool mine = col < a.C1 && (panel || cidx - 16 < a.lchalf);
  const size_t slab = (size_t)k * XK_PERSIST_MAXG;
  double b[RH];
#pragma unroll
  for (int r = 0; r < RH; ++r) {
    const int s = half * RH + r;
    double v = 0.0;
    if (mine && s < ngroups)
      v = panel ? xk_ld_sc1(a.X1P + (slab + s) * 256 + p * 16 + cidx) : xk_ld_sc1(a.X1 + ((slab + s) * 16 + p) * a.C1P + col);
    b[r] = v;
  }
  const int nsteps = (a.C1 - c0 < 16) ? a.C1 - c0 : 16;
#define XK_STEP(K) if (K < nsteps) xk_caqr_mstep32<K, RH>(b, cidx, mine, part, ubuf, sc);
  XK_STEP(0) XK_STEP(1) XK_STEP(2) XK_STEP(3) XK_STEP(4) XK_STEP(5) XK_STEP(6) XK_STEP(7)
  XK_STEP(8) XK_STEP(9) XK_STEP(10) XK_STEP(11) XK_STEP(12) XK_STEP(13) XK_STEP(14) XK_STEP(15)
#undef XK_STEP
  if (!mine) return;
  const bool root = half == 0;
  if (panel) {
    if (split == 0 && root) {
      const double v = (p > cidx) ? 0.0 : b[0];
      if (c0 + p < a.C1) a.Rout[(size_t)(c0 + p) * a.C1P + col] = v;
    }
  } else {
    if (root) {
      if (c0 + p < a.C1) a.Rout[(size_t)(c0 + p) * a.C1P + col] = b[0];
      b[0] = 0.0;
    }
#pragma unroll
    for (int r = 0; r < RH; ++r) {
      const int s = half * RH + r;
      if (s < ngroups) xk_st_sc1(a.X2 + ((slab + s) * 16 + p) * a.C1P + col, b[r]);
    }
  }
}

template <int RH1>
__global__ __launch_bounds__(XK_PERSIST_THREADS) __attribute__((amdgpu_waves_per_eu(6))) void xk_caqr_persist(XkCaqrPersistArgs a) {
  constexpr int LDS_T = 2 * 4 * 18, LDS_M = 2 * 32 * XK_M32_STRIDE(RH1);
  __shared__ __attribute__((aligned(16))) double ubuf[LDS_T > LDS_M ? LDS_T : LDS_M];
  __shared__ __attribute__((aligned(16))) double sc[2 * 4];
  __shared__ unsigned s_slot, s_nx, s_ok;
  unsigned *sync = a.sync, *ab = sync + XK_PS_ABORT * 16;
  const XkPersistArgsPtr ap = (XkPersistArgsPtr)__builtin_amdgcn_kernarg_segment_ptr();
  const unsigned xcc = xk_xcc_id();
  // ---- census: who is where (nothing below assumes a dispatch order), and is the whole grid resident?
  if (threadIdx.x == 0) {
    s_slot = __hip_atomic_fetch_add(sync + (XK_PS_CENSUS + xcc) * 16, 1u, XK_RLX_AGENT);
    __hip_atomic_fetch_add(sync + XK_PS_TOTAL * 16, 1u, XK_RLX_AGENT);
    bool ok = xk_spin_ge(sync + XK_PS_TOTAL * 16, gridDim.x, ab, 1u);
    const unsigned nx = __hip_atomic_load(sync + (XK_PS_CENSUS + xcc) * 16, XK_RLX_AGENT);
    if (ok && nx * 8u != gridDim.x) { __hip_atomic_store(ab, 3u, XK_RLX_AGENT); ok = false; }   // uneven placement
    s_nx = nx;
    s_ok = ok ? 1u : 0u;
  }
  __syncthreads();
  if (!s_ok) {
    if (threadIdx.x == 0 && blockIdx.x == 0) a.status[1] = (int)__hip_atomic_load(ab, XK_RLX_AGENT);
    return;
  }
  const int slot = __builtin_amdgcn_readfirstlane((int)s_slot), nx = __builtin_amdgcn_readfirstlane((int)s_nx);
  const int npanels = (a.C1 + 15) / 16;
  const int TPX = a.TPX, G = a.G, A1 = a.A1, NT = a.NT;
  const int NL = 8 * (nx - NT);                        // last-level workgroups
  unsigned epoch = 0;
  bool ok = true;
  if (slot < NT) {
    // ---- role T: one tile, and a share of this XCD's first-level merges
    const int t = (int)xcc * TPX + slot;                // my tile
    const bool have_tile = slot < TPX && t < a.ntiles;
    const bool leader = (slot % A1) == 0;
    const bool stamp = a.dbg && xcc == 0 && slot == 0 && threadIdx.x == 0;
    long long *sd = (a.dbg && xcc == 0 && threadIdx.x == 0) ? a.dbg + 256 + (size_t)slot * 256 : nullptr;
    for (int k = 0; k < npanels && ok; ++k) {
      const int c0 = 16 * k, trail = max(0, a.C1 - c0 - 16);
      const int lead_off = (k & 1) ? 16 : 0;
      if (stamp) a.dbg[8 * k + 0] = wall_clock64();
      if (sd) sd[8 * k + 0] = wall_clock64();
      if (have_tile) xk_persist_tile(ap, t, c0, k > 0 && leader, lead_off, ubuf, sc);
      if (stamp) a.dbg[8 * k + 1] = wall_clock64();
      if (sd) sd[8 * k + 1] = wall_clock64();
      ok = xk_xcd_barrier(sync, xcc, (unsigned)NT, ++epoch, &s_ok);
      if (!ok) break;
      if (stamp) a.dbg[8 * k + 2] = wall_clock64();
      if (sd) sd[8 * k + 2] = wall_clock64();
      const int msplit = max(1, (trail + 7) / 8);
      for (int item = slot; item < G * msplit; item += NT) {
        const int j = item / msplit, split = item - j * msplit;
        const int base = (int)xcc * TPX + j * A1;
        const int gend = min(min(base + A1, ((int)xcc + 1) * TPX), a.ntiles);
        const int nstrips = max(0, gend - base);
        if (k > 0) {                                     // the pending strip comes from the last level of panel k-1
          if (threadIdx.x == 0) {
            s_ok = xk_spin_ge(sync + (XK_PS_L2FLAG + k - 1) * 16, 1u, ab, 4u) ? 1u : 0u;
          }
          __syncthreads();
          if (!s_ok) { ok = false; break; }
        }
        if (stamp) a.dbg[8 * k + 3] = wall_clock64();
        if (sd) sd[8 * k + 3] = wall_clock64();
        xk_persist_merge1<RH1>(ap, k, (int)xcc * G + j, base, nstrips, split, lead_off, ubuf, sc);
        // publish: every wave drains its write-through stores, then ONE lane counts the item in
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) xk_count_in(sync + (XK_PS_X1CNT + k) * 16, sync + (XK_PS_X1FLAG + k) * 16, 1u, 8u * (unsigned)(G * msplit));
      }
      if (!ok) break;
      if (stamp) a.dbg[8 * k + 4] = wall_clock64();
      if (sd) sd[8 * k + 4] = wall_clock64();
      if (k + 1 < npanels) ok = xk_xcd_barrier(sync, xcc, (unsigned)NT, ++epoch, &s_ok);
      if (stamp) a.dbg[8 * k + 5] = wall_clock64();
      if (sd) sd[8 * k + 5] = wall_clock64();
    }
  } else {
    // ---- role L: the last merge level of every panel, next to the tile step of the following one
    const int lidx = (slot - NT) * 8 + (int)xcc;
    const bool stamp = a.dbg && lidx == 0 && threadIdx.x == 0;
    for (int k = 0; k < npanels && ok; ++k) {
      const int trail = max(0, a.C1 - 16 * k - 16);
      const int lsplit = max(1, (trail + a.lchalf - 1) / a.lchalf);
      if (lidx >= lsplit) continue;                      // (fewer columns, fewer workgroups: lsplit only shrinks)
      if (threadIdx.x == 0) s_ok = xk_spin_ge(sync + (XK_PS_X1FLAG + k) * 16, 1u, ab, 5u) ? 1u : 0u;
      __syncthreads();
      if (!s_ok) { ok = false; break; }
      if (stamp) a.dbg[8 * k + 6] = wall_clock64();
      for (int item = lidx; item < lsplit; item += NL) xk_persist_last(ap, k, item, ubuf, sc);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (threadIdx.x == 0) {
        const int mine_items = (lsplit - lidx + NL - 1) / NL;
        xk_count_in(sync + (XK_PS_L2CNT + k) * 16, sync + (XK_PS_L2FLAG + k) * 16, (unsigned)mine_items, (unsigned)lsplit);
      }
      if (stamp) a.dbg[8 * k + 7] = wall_clock64();
    }
  }
  if (!ok && threadIdx.x == 0) a.status[1] = (int)__hip_atomic_load(ab, XK_RLX_AGENT);
}
