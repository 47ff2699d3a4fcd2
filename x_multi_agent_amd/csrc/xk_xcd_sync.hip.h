// xk_xcd_sync.hip.h -- what workgroups of ONE launch use to hand data to each other on MI355X (8 XCDs, one L2 each; the L2s
// are not coherent with each other).
//
//   * inside an XCD: plain stores + s_waitcnt vmcnt(0) on the producer (the data is in the XCD's L2 once vmcnt reaches 0)
//     and L1-bypassing (sc1) loads on the consumer -- measured 2.8 us per hand-off including 24 KB out and 24 KB in per
//     workgroup, 0 stale words in 3e8 (tools/exp/xcd_sync_probe.hip);
//   * across XCDs: write-through (sc1) stores into buffers that are read with sc1 loads only and never reused inside a launch,
//     so that no L2 ever holds a stale or dirty copy of them;
//   * a workgroup learns the XCD it runs on from HW_REG_XCC_ID (placement is measured, never assumed: the dispatcher's
//     round-robin over the XCDs starts wherever the previous kernel left it);
//   * every spin is bounded and looks at an abort word: a launch whose workgroups are not all resident gives up, the host
//     redoes the update with the multi-launch schedule (xk_api.hip).
// These orderings are stronger than what the relaxed agent-scope atomics below promise in the HIP memory model; they are what
// gfx950 does, and tests/test_gpu_resident_caqr.py + tests/test_gpu_soak.py (random shapes against the C oracle) are the gate.
#pragma once
#include <hip/hip_runtime.h>

#ifndef XK_SPIN_SLEEP
#define XK_SPIN_SLEEP 8             // s_sleep argument between two polls of a flag (x 64 clocks)
#endif
// a poll gives up after XK_SPIN_TICKS (2 ms: the longest wait of a healthy launch is the last level's for the first roots, tens
// of microseconds; the first version's 0.2 s was six dropped camera frames)
#ifndef XK_SPIN_TICKS
#define XK_SPIN_TICKS 200000LL      // 100 MHz ticks
#endif
#define XK_CAQR_MAXP 32             // panels per launch (C1 <= 512)
// -DXK_SYNC_STRICT=1: every hand-off as the HIP memory model spells it -- the producer's counter update is an agent-scope RELEASE,
// every consumer thread runs an agent-scope ACQUIRE fence behind the poll.  On gfx950 an agent-scope release writes the XCD's L2
// back (buffer_wbl2 sc1) and an acquire invalidates it (buffer_inv sc1): with ~100 hand-offs per workgroup and launch that is what
// the whole design exists to avoid -- measured in DESIGN 3.2 -- so the default keeps relaxed counters and makes the DATA visible
// instead: write-through (sc1) stores for everything another XCD reads, plain stores + s_waitcnt vmcnt(0) inside an XCD (one L2),
// L1-bypassing (sc1) loads on the consumer.  That is what the hardware does, not what the language promises; the soak tests are
// the gate, and this switch is the reference the default can be checked against after a compiler or ROCm update.
#ifndef XK_SYNC_STRICT
#define XK_SYNC_STRICT 0
#endif
#if XK_SYNC_STRICT
#define XK_ARRIVE_ORDER __ATOMIC_RELEASE
#define XK_ACQUIRE_FENCE() __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent")
#else
#define XK_ARRIVE_ORDER __ATOMIC_RELAXED
#define XK_ACQUIRE_FENCE()
#endif
#define XK_RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT
// The completion marker a launch writes into pinned host memory behind its host-visible results (correction, status words).  Default:
// the results are relaxed system-scope stores, `s_waitcnt vmcnt(0)` waits for their acknowledgement, then the marker is a relaxed
// system-scope store -- what gfx950 does (a system-scope store is acknowledged once it has left for the host), not what the HIP
// memory model promises.  -DXK_SYNC_STRICT=1 (or -DXK_MARKER_RELEASE=1 alone) stores the marker with a system-scope RELEASE
// (buffer_wbl2 sc0 sc1 in front of it: a write-back of the XCD's L2, once per launch).  Either way only the words the marker sits
// behind are for the host; the posterior in HBM (Pout) is valid to later work THROUGH STREAM ORDER, not through the marker.
#ifndef XK_MARKER_RELEASE
#define XK_MARKER_RELEASE XK_SYNC_STRICT
#endif
#if XK_MARKER_RELEASE
#define XK_MARKER_ORDER __ATOMIC_RELEASE
#else
#define XK_MARKER_ORDER __ATOMIC_RELAXED
#endif

__device__ __forceinline__ double xk_ld_sc1(const double *p) {
  return __builtin_bit_cast(double, __hip_atomic_load(reinterpret_cast<const unsigned long long *>(p), XK_RLX_AGENT));
}
// XCD-local variant (experiment, XK_PIPE_LOCALLD=1): a workgroup-scope load (sc0) behind an explicit L1 invalidate -- it may be
// served by the XCD's L2 instead of going out to the fabric like a device-scope load of a line another CU has written
__device__ __forceinline__ double xk_ld_grp(const double *p) {
  return __builtin_bit_cast(double, __hip_atomic_load(reinterpret_cast<const unsigned long long *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
}
__device__ __forceinline__ void xk_inv_l1() { asm volatile("buffer_inv sc1" ::: "memory"); }
__device__ __forceinline__ void xk_st_sc1(double *p, double v) {
  __hip_atomic_store(reinterpret_cast<unsigned long long *>(p), __builtin_bit_cast(unsigned long long, v), XK_RLX_AGENT);
}
// a store the HOST is meant to see (pinned, fine-grained memory): system scope, written through
__device__ __forceinline__ void xk_st_sys(double *p, double v) {
  __hip_atomic_store(reinterpret_cast<unsigned long long *>(p), __builtin_bit_cast(unsigned long long, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ unsigned xk_xcc_id() {
  unsigned x;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
  return x & 7u;
}

// one lane polls one word (relaxed, L1-bypassing) until it reaches `target`
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

// DATA-POLLED hand-offs (round 5; relaxed build only).  The cross-XCD slabs (roots going up, pending strips coming down) are
// per-panel and never reused inside a launch, so a slot can say by itself whether it has been written: every slot holds a
// NOT-YET pattern -- a NaN payload no arithmetic produces -- until its producer stores the value (one write-through 8-byte store,
// untorn on gfx950), and the consumer's lanes load their slots until none of them is that pattern.  No counter, no drain of the
// producer's stores, no barriers around the poll: a hand-off costs a store's and a load's latency (~1.5 us) instead of 3.5-4.
// The launch re-arms the OTHER set of slabs for its successor (xk_caqr_pipe entry), the host arms both at creation and after a
// launch that gave up.  -DXK_SYNC_STRICT=1 keeps the counters and release / acquire pairs.
#define XK_NOTYET_BITS 0x7FF8BEEF7FF8BEEFull           // (both halves equal: hipMemsetD32Async arms a slab)
__device__ __forceinline__ bool xk_is_notyet(double v) { return __builtin_bit_cast(unsigned long long, v) == XK_NOTYET_BITS; }
__device__ __forceinline__ double xk_notyet() { return __builtin_bit_cast(double, (unsigned long long)XK_NOTYET_BITS); }
// One wave: the lanes with want = true load their R slots (stride `step` doubles) until every one of them has been written.
// Returns false if the wave gave up (bound of the spin, or somebody else's abort word).
template <int R>
__device__ __forceinline__ bool xk_poll_slots(double (&b)[R], const double *src, size_t step, bool want, unsigned *abort_, unsigned reason) {
  const long long t0 = wall_clock64();
  for (unsigned it = 0;; ++it) {
    bool ok = true;
    if (want) {
#pragma unroll
      for (int s = 0; s < R; ++s) b[s] = xk_ld_sc1(src + (size_t)s * step);
#pragma unroll
      for (int s = 0; s < R; ++s) ok = ok && !xk_is_notyet(b[s]);
    }
    if (__builtin_amdgcn_ballot_w64(!ok) == 0ull) return true;
    if ((it & 15u) == 15u) {
      if (__hip_atomic_load(abort_, XK_RLX_AGENT)) return false;
      if (wall_clock64() - t0 > XK_SPIN_TICKS) { __hip_atomic_store(abort_, reason, XK_RLX_AGENT); return false; }
    }
    __builtin_amdgcn_s_sleep(1);
  }
}
#ifndef XK_DATA_POLL
#define XK_DATA_POLL (!XK_SYNC_STRICT)
#endif

// Hides a pointer from loop-invariant code motion: the per-row addresses of a strip are then formed where they are used (one
// 64-bit add each) instead of being hoisted out of the panel loop, dozens of registers' worth, and spilled.
template <typename T> __device__ __forceinline__ T *xk_opaque(T *p) {
  asm volatile("" : "+v"(p));
  return p;
}
