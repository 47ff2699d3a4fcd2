// xk_caqr_resident.hip.h -- QR compression (VioUpdater::applyQRDecomposition, src/x/vio/vio_updater.cpp:487-512) in ONE
// launch with the row stack RESIDENT IN REGISTERS for the whole factorisation.
//
// What the single-launch experiment of xk_caqr_persist.hip.h taught: XCD-local hand-offs are cheap (2.8 us),
// but two 768-thread workgroups per CU (the geometry 400 64-row tiles force on 256 CUs) cost more than the boundaries
// save: skewed tile steps, 80-VGPR merges with three-fold redundant panel work, and a full pass over the 37 MB tile stack
// per panel that still has to drain before each barrier.  This kernel changes the geometry instead:
//
//   * ONE workgroup per CU.  The valid rows of the stack (~22 800 at the headline size) are re-cut into 248 FAT TILES of
//     <= 96 rows -- in-place QR does not care how rows are grouped -- one per workgroup, 4 lanes per column x 24 rows per
//     lane, and a thread keeps ITS column of ITS rows in registers from the first panel to the last (the column <-> thread
//     map is absolute: wave w holds columns 16 w .. 16 w + 15, i.e. panel w; finished waves just keep the barriers).
//     Rows are gathered ONCE through a row map (valid row -> physical row of the 64-row slots the per-feature kernel
//     writes); after that the only traffic of a panel is the 16-row pivot strip of every tile (5.6 MB per panel over the
//     chip instead of ~30 MB), XCD-local.
//   * XCD x = 31 tile workgroups (two first-level groups of 16 / 15 strips) + ONE workgroup for the last merge level,
//     which therefore has a CU of its own next to the tile steps.  First-level merges run on 16 lanes per column with
//     16 panel + 8..32 trailing columns per item (as few as keeps the items of a group within its workgroups), at most
//     one item per CU; the last level runs in two halves, the first beside the second half of the first level (the
//     strips of a merge are upper triangular in the panel columns: its steps 0..7 need rows 0..7 of the roots only).
//   * Steps with a one-reflector look-ahead (xk_linalg.hip.h: xk_caqr_steps_la, xk_caqr_form / xk_caqr_apply).
//   * Group leaders: the root strip leaves for the last level (X1) and comes back one panel later as the pending strip
//     (X2 -> first level -> Hq).  Meanwhile the leader needs another pivot strip in the registers of its part-0 lanes:
//     it swaps registers 0..15 of its part-0 and part-1 lanes (one quad-permute DPP move per register) and zeroes the
//     "away" half until the returned rows are loaded into it.
// Synchronisation primitives, bounded spins and the cross-XCD slabs are those of xk_caqr_persist.hip.h; the placement census
// does not wait for the whole grid, and the sync words are double-buffered (a launch zeroes the set its successor uses).
// DESIGN 3.2.1 has the anatomy, what made it fast and what was tried and rejected.
#pragma once
#include <hip/hip_runtime.h>

#include "xk_caqr_persist.hip.h"

#define XK_RES_THREADS 768          // 12 waves, one workgroup per CU: up to 168 VGPRs (a 1024-thread version spilled 271)
#define XK_RES_RPL 24               // rows per lane: fat tiles of 4 x 24 = 96 rows
#ifndef XK_RES_NPARK
#define XK_RES_NPARK 6              // tile registers parked in LDS across a merge call (see the kernel)
#endif
#ifndef XK_RES_CHAINWAVE
#define XK_RES_CHAINWAVE 0
#endif
#ifndef XK_RES_LOOKAHEAD
#define XK_RES_LOOKAHEAD 1          // one-reflector look-ahead steps (xk_caqr_steps_la); 0 = the plain steps, for A/B builds
#endif
#ifndef XK_RES_NL
#define XK_RES_NL 1                 // workgroups per XCD that run the last merge level (A/B builds: 2)
#endif
#define XK_RES_NT (32 - XK_RES_NL)  // tile workgroups per XCD

struct XkCaqrResidentArgs {
  const double *A;        // tiles [ntiles][64][C1P] row-major as the per-feature kernels wrote them (read once)
  const int *tile_rows;   // valid rows per 64-row slot (0 = rejected track)
  const int *rowmap;      // valid row g -> physical row of A (slot * 64 + row in slot), [R]
  int R, TR;              // valid rows in total, rows per fat tile (<= 96)
  int C1P, C1;
  double *Rout;           // [C1P][C1P] row-major
  double *S;              // [248][16][C1P] pivot strips (XCD-local hand-off tile step <-> first level)
  double *PB1;            // [248][16][16] their panel blocks
  double *Hq;             // [16][16][C1P] what the first level leaves of a group's pending strip (for its leader)
  double *X1, *X1P, *X2;  // cross-XCD slabs, as in xk_caqr_persist
  unsigned *sync;
  unsigned *sync_next;    // the other set of sync words: zeroed by this launch for the next one
  int *status;
  long long *dbg;
};
typedef const XkCaqrResidentArgs __attribute__((address_space(4))) *XkResidentArgsPtr;
__device__ __forceinline__ XkCaqrResidentArgs xk_resident_args(XkResidentArgsPtr ap) {
  XkCaqrResidentArgs a;
  __builtin_memcpy(&a, (const void *)ap, sizeof(a));
  return a;
}

// Hides a pointer from loop-invariant code motion: the sixteen per-row addresses of a strip are then formed where they
// are used (one 64-bit add each) instead of being hoisted out of the panel loop, 96 registers' worth, and spilled.
template <typename T> __device__ __forceinline__ T *xk_opaque(T *p) {
  asm volatile("" : "+v"(p));
  return p;
}

// The 16 steps of a merge in two halves (one-reflector look-ahead inside each, xk_caqr_msteps_la): reflectors 0..7, all
// applied on return, and reflectors 8..15.  Between the halves the caller may publish rows 0..7 of the root strip -- they
// are final -- and the first barrier of the second half, which every wave reaches with its stores drained, is where
// workgroup thread 0 counts the item in (cnt / flag / need).  The strips of a merge are upper triangular in the panel
// columns, so reflector j is zero in rows > j of every strip: the level above can run ITS steps 0..7 on rows 0..7 alone.
template <int RPL>
__device__ __forceinline__ void xk_res_msteps_a(double (&b)[RPL], int rel, bool live, int part, int nsteps, double *ubuf, double *sc) {
  xk_caqr_mform<0, RPL>(b, rel, part, ubuf, sc);
  __syncthreads();
#define XK_IT(K)                                                                                  \
  if (K < nsteps) { xk_caqr_apply<K - 1, 16, RPL>(b, rel, live, part, ubuf, sc); xk_caqr_mform<K, RPL>(b, rel, part, ubuf, sc); __syncthreads(); } \
  else if (K == nsteps) xk_caqr_apply<K - 1, 16, RPL>(b, rel, live, part, ubuf, sc);
  XK_IT(1) XK_IT(2) XK_IT(3) XK_IT(4) XK_IT(5) XK_IT(6) XK_IT(7)
  if (nsteps >= 8) xk_caqr_apply<7, 16, RPL>(b, rel, live, part, ubuf, sc);
}
template <int RPL>
__device__ __forceinline__ void xk_res_msteps_b(double (&b)[RPL], int rel, bool live, int part, int nsteps, double *ubuf, double *sc,
                                                unsigned *cnt, unsigned *flag, unsigned need) {
  if (8 < nsteps) xk_caqr_mform<8, RPL>(b, rel, part, ubuf, sc);
  if (cnt) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (cnt && threadIdx.x == 0) xk_count_in(cnt, flag, 1u, need);
  XK_IT(9) XK_IT(10) XK_IT(11) XK_IT(12) XK_IT(13) XK_IT(14) XK_IT(15)
#undef XK_IT
  if (nsteps == 16) xk_caqr_apply<15, 16, RPL>(b, rel, live, part, ubuf, sc);
}

#if XK_RES_CHAINWAVE
#include "xk_caqr_chainwave.hip.h"   // experiment, measured and rejected (see that file)
#endif

// first merge level of one group, 16 lanes per column (lane p = row p of every strip, register s = strip s, register
// NS = the pending strip): 16 panel + 32 trailing columns per workgroup.  Strips come from S / PB1, the root goes to
// X1 / X1P, the other strips back to S, the pending strip's rest to Hq.
template <int RPL>
__device__ __noinline__ bool xk_resident_merge1(XkResidentArgsPtr ap, int k, int gid, int base, int nstrips, int split, int MCH,
                                                double *ubuf, double *sc, unsigned *s_ok, unsigned need) {
  constexpr int NP = 16, NS = RPL - 2;
  const XkCaqrResidentArgs a = xk_resident_args(ap);
  const int c0 = 16 * k;
  const int cidx = (int)threadIdx.x / NP, part = threadIdx.x & (NP - 1);
  const bool panel = cidx < 16;
  const int col = panel ? c0 + cidx : c0 + 16 + split * MCH + (cidx - 16);
  const bool mine = col < a.C1 && (panel || cidx - 16 < MCH);
  const bool pend_ok = k > 0 && nstrips > 0;
  const size_t lane_off = panel ? (size_t)part * 16 + cidx : (size_t)part * a.C1P + col;
  const size_t strip_step = panel ? 256 : (size_t)16 * a.C1P;
  double *g0 = (panel ? a.PB1 + (size_t)base * 256 : a.S + (size_t)base * 16 * a.C1P) + lane_off;
  const double *pend_src = a.X2 + (((size_t)(k > 0 ? k - 1 : 0) * XK_PERSIST_MAXG + gid) * 16 + part) * a.C1P + col;
  double *pend_dst = a.Hq + ((size_t)gid * 16 + part) * a.C1P + col;
  double b[RPL];
#pragma unroll
  for (int s = 0; s < NS; ++s) b[s] = (mine && s < nstrips) ? xk_ld_sc1(g0 + (size_t)s * strip_step) : 0.0;
  // the pending strip is the one input that comes from the last level of panel k - 1: wait for it with the other sixteen
  // loads already in flight
  if (k > 0) {
    if (threadIdx.x == 0)
      *s_ok = xk_spin_ge(a.sync + (XK_PS_L2FLAG + k - 1) * 16, 1u, a.sync + XK_PS_ABORT * 16, 4u) ? 1u : 0u;
    __syncthreads();
    if (!*s_ok) return false;
  }
  b[NS] = (mine && pend_ok) ? xk_ld_sc1(pend_src) : 0.0;
  b[NS + 1] = 0.0;
  const int nsteps = (a.C1 - c0 < 16) ? a.C1 - c0 : 16;
  const bool mstamp = a.dbg && gid == 0 && split == 0 && threadIdx.x == 0;
  if (mstamp) {
    double sink = 0;
    for (int s = 0; s < RPL; ++s) sink += b[s];
    asm volatile("" ::"v"(sink));
    a.dbg[256 + 8 * k + 2] = wall_clock64(); a.dbg[256 + 8 * k + 4] = clock64();
  }
  unsigned *cnt_a = a.sync + (XK_PS_X1CNTA + k) * 16, *flag_a = a.sync + (XK_PS_X1FLAGA + k) * 16;
#if XK_RES_LOOKAHEAD
  if (panel) __builtin_amdgcn_s_setprio(3);
  xk_res_msteps_a<RPL>(b, cidx, mine, part, nsteps, ubuf, sc);
  // rows 0..7 of the root strip are final: out they go while steps 8..15 run (the last level starts on them)
  if (mine && part < 8) {
    if (panel) {
      if (split == 0) xk_st_sc1(a.X1P + ((size_t)k * XK_PERSIST_MAXG + gid) * 256 + part * 16 + cidx, (part > cidx) ? 0.0 : b[0]);
    } else {
      xk_st_sc1(a.X1 + (((size_t)k * XK_PERSIST_MAXG + gid) * 16 + part) * a.C1P + col, b[0]);
    }
  }
  xk_res_msteps_b<RPL>(b, cidx, mine, part, nsteps, ubuf, sc, cnt_a, flag_a, need);
  if (panel) __builtin_amdgcn_s_setprio(0);
  const int first_row = 8;
#else
#define XK_STEP(K) if (K < nsteps) xk_caqr_mstep<K, RPL>(b, cidx, mine, part, ubuf, sc);
  XK_STEP(0) XK_STEP(1) XK_STEP(2) XK_STEP(3) XK_STEP(4) XK_STEP(5) XK_STEP(6) XK_STEP(7)
  XK_STEP(8) XK_STEP(9) XK_STEP(10) XK_STEP(11) XK_STEP(12) XK_STEP(13) XK_STEP(14) XK_STEP(15)
#undef XK_STEP
  const int first_row = 0;
  (void)cnt_a; (void)flag_a;
#endif
  if (mstamp) { a.dbg[256 + 8 * k + 3] = wall_clock64(); a.dbg[256 + 8 * k + 5] = clock64(); }
  if (!mine) return true;
  if (panel) {
    if (split == 0 && part >= first_row)
      xk_st_sc1(a.X1P + ((size_t)k * XK_PERSIST_MAXG + gid) * 256 + part * 16 + cidx, (part > cidx) ? 0.0 : b[0]);
  } else {
    if (part >= first_row) xk_st_sc1(a.X1 + (((size_t)k * XK_PERSIST_MAXG + gid) * 16 + part) * a.C1P + col, b[0]);
#pragma unroll
    for (int s = 1; s < NS; ++s)
      if (s < nstrips) g0[(size_t)s * strip_step] = b[s];
    if (pend_ok) pend_dst[0] = b[NS];
  }
  return true;
}

// last merge level, 16 lanes per column (lane p = row p of every root strip): 16 roots -> 16 rows of R, the rest -> X2
__device__ __noinline__ bool xk_resident_last(XkResidentArgsPtr ap, int k, int split, int lchalf, double *ubuf, double *sc, unsigned *s_ok) {
  constexpr int NP = 16, RPL = 16;
  const XkCaqrResidentArgs a = xk_resident_args(ap);
  const int c0 = 16 * k;
  const int cidx = (int)threadIdx.x / NP, part = threadIdx.x & (NP - 1);
  const bool panel = cidx < 16;
  const int col = panel ? c0 + cidx : c0 + 16 + split * lchalf + (cidx - 16);
  const bool mine = col < a.C1 && (panel || cidx - 16 < lchalf);
  const size_t slab = (size_t)k * XK_PERSIST_MAXG;
  double b[RPL];
  // rows 0..7 of the roots first (published after step 7 of the first level), rows 8..15 when the first level is through:
  // steps 0..7 here overlap steps 8..15 there
  unsigned *ab = a.sync + XK_PS_ABORT * 16;
  long long *lst = (a.dbg && split == 0 && threadIdx.x == 0) ? a.dbg + 768 + 8 * k : nullptr;
  if (threadIdx.x == 0) *s_ok = xk_spin_ge(a.sync + (XK_PS_X1FLAGA + k) * 16, 1u, ab, 5u) ? 1u : 0u;
  __syncthreads();
  if (!*s_ok) return false;
  if (lst) lst[0] = wall_clock64();
#pragma unroll
  for (int s = 0; s < RPL; ++s)
    b[s] = (!mine || part >= 8) ? 0.0 : panel ? xk_ld_sc1(a.X1P + (slab + s) * 256 + part * 16 + cidx)
                                              : xk_ld_sc1(a.X1 + ((slab + s) * 16 + part) * a.C1P + col);
  const int nsteps = (a.C1 - c0 < 16) ? a.C1 - c0 : 16;
  if (panel) __builtin_amdgcn_s_setprio(3);
  xk_res_msteps_a<RPL>(b, cidx, mine, part, nsteps, ubuf, sc);
  if (lst) lst[1] = wall_clock64();
  // (Looking for the second flag two steps early and starting these loads into spare registers behind steps 6 and 7 was
  //  measured: the first half went from 8.8 to 10.5 us, the second from 9.0 to 4.5, the kernel from 0.414 to 0.428 ms.)
  // (a panel of <= 8 columns is the last one and has no trailing columns: rows 8..15 of its roots are structurally zero,
  //  there is no second half to wait for -- 4 us of the kernel's tail)
  if (nsteps > 8) {
    if (threadIdx.x == 0) *s_ok = xk_spin_ge(a.sync + (XK_PS_X1FLAG + k) * 16, 1u, ab, 5u) ? 1u : 0u;
    __syncthreads();
    if (!*s_ok) return false;
    if (lst) lst[2] = wall_clock64();
    if (mine && part >= 8) {
#pragma unroll
      for (int s = 0; s < RPL; ++s)
        b[s] = panel ? xk_ld_sc1(a.X1P + (slab + s) * 256 + part * 16 + cidx) : xk_ld_sc1(a.X1 + ((slab + s) * 16 + part) * a.C1P + col);
    }
    xk_res_msteps_b<RPL>(b, cidx, mine, part, nsteps, ubuf, sc, nullptr, nullptr, 0u);
  }
  if (panel) __builtin_amdgcn_s_setprio(0);
  if (lst) lst[3] = wall_clock64();
  if (!mine) return true;
  if (panel) {
    if (split == 0) {
      const double v = (part > cidx) ? 0.0 : b[0];
      if (c0 + part < a.C1) a.Rout[(size_t)(c0 + part) * a.C1P + col] = v;
    }
  } else {
    if (c0 + part < a.C1) a.Rout[(size_t)(c0 + part) * a.C1P + col] = b[0];
    b[0] = 0.0;
#pragma unroll
    for (int s = 0; s < RPL; ++s) xk_st_sc1(a.X2 + ((slab + s) * 16 + part) * a.C1P + col, b[s]);
  }
  return true;
}

__global__ __launch_bounds__(XK_RES_THREADS) void xk_caqr_resident(XkCaqrResidentArgs a) {
  constexpr int RPL = XK_RES_RPL, NT = XK_RES_NT, NL = XK_RES_NL, RM1 = 18, A1 = (NT + 1) / 2, G = 2;
  constexpr int LDS_T = 2 * 4 * (RPL + 2), LDS_M = 2 * 16 * (RM1 + 2), LDS_L = 2 * 16 * 18;
  constexpr int LDS_MAX = LDS_T > LDS_M ? (LDS_T > LDS_L ? LDS_T : LDS_L) : (LDS_M > LDS_L ? LDS_M : LDS_L);
  __shared__ __attribute__((aligned(16))) double ubuf[LDS_MAX];
  __shared__ __attribute__((aligned(16))) double sc[2 * 4];
  constexpr int NPARK = XK_RES_NPARK;
  __shared__ double park[(NPARK > 0 ? NPARK : 1) * XK_RES_THREADS];
  __shared__ unsigned s_slot, s_ok;
  unsigned *sync = a.sync, *ab = sync + XK_PS_ABORT * 16;
  const XkResidentArgsPtr ap = (XkResidentArgsPtr)__builtin_amdgcn_kernarg_segment_ptr();
  const unsigned xcc = xk_xcc_id();
  const long long t_entry = a.dbg ? wall_clock64() : 0;
  // Placement census: a workgroup reads the XCD it runs on and takes the next slot there (the dispatcher deals workgroups
  // round-robin over the XCDs, but where it starts depends on what ran before: block b -> XCD b % 8 does NOT hold).  An
  // XCD that receives a 33rd workgroup raises the abort word -- every spin looks at it -- and the host falls back to the
  // multi-launch schedule.  Nobody waits for the census to complete (the first version spun until all 256 had arrived:
  // 6.4 us before the first load); nothing needs every workgroup to be resident before the first group barrier, and
  // all spins are bounded.
  if (threadIdx.x == 0) {
    const unsigned sl = __hip_atomic_fetch_add(sync + (XK_PS_CENSUS + xcc) * 16, 1u, XK_RLX_AGENT);
    const bool bad = sl >= (unsigned)(NT + NL) || gridDim.x != 8u * (NT + NL);
    if (bad) { __hip_atomic_store(ab, 3u, XK_RLX_AGENT); a.status[1] = 3; }
    s_slot = sl;
    s_ok = bad ? 0u : 1u;
  }
  __syncthreads();
  if (!s_ok) return;
  const int slot = __builtin_amdgcn_readfirstlane((int)s_slot);
  const long long t_census = a.dbg ? wall_clock64() : 0;
  const int npanels = (a.C1 + 15) / 16;
  unsigned epoch = 0;
  bool ok = true;
  const int tid = threadIdx.x;
  if (slot < NT) {
    // ---- role T: one fat tile in registers + a share of the XCD's first-level merges
    const int j = (int)xcc * NT + slot;                      // my fat tile: valid rows [j TR, (j + 1) TR)
    const int cidx = tid >> 2, part = tid & 3;               // tile layout: 4 lanes per column (threads < 768)
    const bool tlane = true;                                 // (768 threads = 192 columns x 4 lanes)
    const int cabs = cidx;                                   // ABSOLUTE column of this thread, all panels
    const bool mine = tlane && cabs < a.C1;
    const bool leader = (slot % A1) == 0;
    const int grp = slot / A1;
    const int gsize = min(A1, NT - grp * A1), mygid = (int)xcc * G + grp;
    const bool stamp = a.dbg && xcc == 0 && slot == 1 && tid == 0;
    double b[RPL];
    {   // the one pass over the stack: gather my rows through the row map
      const int g0 = j * a.TR + part * RPL, gend = min((j + 1) * a.TR, a.R);
#pragma unroll
      for (int r = 0; r < RPL; ++r) {
        const int g = g0 + r;
        double v = 0.0;
        if (mine && g < gend) {
          const int pr = a.rowmap[g];
          // (the row and its track's verdict are fetched side by side, not one after the other: a rejected track's slot
          //  holds whatever the per-feature kernel left there, the select drops it)
          const double x = a.A[(size_t)pr * a.C1P + cabs];
          v = (a.tile_rows[pr >> 6] > 0) ? x : 0.0;
        }
        b[r] = v;
      }
    }
    double *myS = a.S + (size_t)j * 16 * a.C1P;
    if (stamp) { a.dbg[512] = t_entry; a.dbg[513] = t_census; }
    for (int k = 0; k < npanels && ok; ++k) {
      const int c0 = 16 * k, trail = max(0, a.C1 - c0 - 16);
      const int rel = tlane ? cabs - c0 : -1;
      if (stamp) a.dbg[8 * k + 0] = wall_clock64();
      if (leader && k > 0 && tlane) {
        // the strip that was my pivot strip is away at the last level: the rows part 1 kept become the new pivot strip
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const double sw = xk_dpp_quad<0xE1>(b[r]);         // quad_perm [1,0,2,3]: parts 0 and 1 trade places
          b[r] = (part == 0) ? sw : (part == 1) ? 0.0 : b[r];
        }
      }
      const int nsteps = (a.C1 - c0 < 16) ? a.C1 - c0 : 16;
      if (stamp) a.dbg[256 + 8 * k + 0] = wall_clock64();
#if XK_RES_CHAINWAVE
      // (wave 0's columns are finished from panel 1 on: it is the chain wave; panel 0 has no idle wave)
      if (k > 0) xk_cw_steps<RPL>(b, rel, mine, part, nsteps, (tid >> 6) == 0, tid & 63, ubuf, sc);
      else xk_caqr_steps_la<4, RPL>(b, rel, mine, part, nsteps, (rel >> 4) == 0, ubuf, sc);
#elif XK_RES_LOOKAHEAD
      xk_caqr_steps_la<4, RPL>(b, rel, mine, part, nsteps, (rel >> 4) == 0, ubuf, sc);
#else
      xk_caqr_steps<4, RPL>(b, rel, mine, part, nsteps, ubuf, sc);
#endif
      if (stamp) a.dbg[256 + 8 * k + 1] = wall_clock64();
      // hand the pivot strip over: rows 0..15 of the part-0 lanes.  (Storing row K right after step K -- it is final by then --
      // so that the stores drain behind the remaining steps was measured: tile phase 15.4 -> 21.5 us; the stores' data keeps
      // registers live across the steps and the compiler's waits land inside the chain.)
      if (mine && part == 0 && rel >= 0) {
        if (rel < 16) {
          double *pb = xk_opaque(a.PB1 + (size_t)j * 256 + rel);
#pragma unroll
          for (int r = 0; r < 16; ++r) pb[r * 16] = (r > rel) ? 0.0 : b[r];
        } else {
          double *ps = xk_opaque(myS + cabs);
#pragma unroll
          for (int r = 0; r < 16; ++r) ps[(size_t)r * a.C1P] = b[r];
        }
      }
      if (stamp) a.dbg[8 * k + 1] = wall_clock64();
      // the two first-level groups of an XCD are independent of each other until the last level: each has its own barrier
      // and runs its own merge items (a smaller barrier waits for fewer stragglers)
      ok = xk_flag_barrier(sync, XK_PS_GBARCNT, XK_PS_GBARGEN, (unsigned)mygid, (unsigned)gsize, ++epoch, &s_ok);
      if (!ok) break;
      if (stamp) a.dbg[8 * k + 2] = wall_clock64();
      // trailing columns per first-level item: as few as keeps G x msplit items within the XCD's 31 workgroups (a step of a
      // 48-column item takes 0.83 us, of a 28-column one 0.5: fewer waves share the LDS and the vector pipe)
      const int mch = min(32, max(8, 4 * ((trail + 4 * 15 - 1) / (4 * 15))));
      const int msplit = max(1, (trail + mch - 1) / mch);
      // The merge bodies are calls (inlined, they spill three times as much), so whatever of the fat tile stays live across
      // them must sit in the 64 callee-saved VGPRs of the 168: forty-eight tile registers plus the lane's indices do not
      // fit, and the register that lost was reloaded and stored back in every one of the 32 steps of a panel.  NPARK tile
      // registers wait in LDS instead while this workgroup runs a merge item.
      const bool has_item = slot - grp * A1 < msplit;
      if (has_item) {
#pragma unroll
        for (int q = 0; q < NPARK; ++q) park[q * XK_RES_THREADS + tid] = b[RPL - NPARK + q];
      }
      for (int split = slot - grp * A1; split < msplit; split += gsize) {
        const int jg = grp;
        const int base = (int)xcc * NT + jg * A1;
        const int nstrips = gsize;
        if (stamp) a.dbg[8 * k + 3] = wall_clock64();
        if (!xk_resident_merge1<RM1>(ap, k, (int)xcc * G + jg, base, nstrips, split, mch, ubuf, sc, &s_ok, 8u * (unsigned)(G * msplit))) { ok = false; break; }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
#if !XK_RES_LOOKAHEAD
        if (tid == 0) xk_count_in(sync + (XK_PS_X1CNTA + k) * 16, sync + (XK_PS_X1FLAGA + k) * 16, 1u, 8u * (unsigned)(G * msplit));
#endif
        if (tid == 0) xk_count_in(sync + (XK_PS_X1CNT + k) * 16, sync + (XK_PS_X1FLAG + k) * 16, 1u, 8u * (unsigned)(G * msplit));
      }
      if (!ok) break;
      if (has_item) {
#pragma unroll
        for (int q = 0; q < NPARK; ++q) b[RPL - NPARK + q] = park[q * XK_RES_THREADS + tid];
      }
      if (stamp) a.dbg[8 * k + 4] = wall_clock64();
      if (k + 1 < npanels) {
        ok = xk_flag_barrier(sync, XK_PS_GBARCNT, XK_PS_GBARGEN, (unsigned)mygid, (unsigned)gsize, ++epoch, &s_ok);
        if (!ok) break;
        // my strip comes back (trailing columns of the NEXT panels only: everything up to c0 + 15 is finished)
        if (mine && rel >= 16) {
          if (!leader) {
            if (part == 0) {
              const double *ps = xk_opaque(myS + cabs);
#pragma unroll
              for (int r = 0; r < 16; ++r) b[r] = xk_ld_sc1(ps + (size_t)r * a.C1P);
            }
          } else if (k > 0 && part == 1) {                   // what the first level left of the rows that were away
            const double *hq = xk_opaque(a.Hq + (size_t)((int)xcc * G + grp) * 16 * a.C1P + cabs);
#pragma unroll
            for (int r = 0; r < 16; ++r) b[r] = xk_ld_sc1(hq + (size_t)r * a.C1P);
          }
        }
      }
      if (stamp) a.dbg[8 * k + 5] = wall_clock64();
    }
    if (stamp) a.dbg[514] = wall_clock64();
  } else {
    // ---- role L: the last merge level of every panel, on a CU of its own
    const int lidx = (int)xcc * NL + (slot - NT);             // 8 NL of them
    // nothing to do until the first roots arrive: leave the other set of sync words zeroed for the next launch
    for (int i = lidx * XK_RES_THREADS + tid; i < XK_PS_WORDS * 16; i += 8 * NL * XK_RES_THREADS) a.sync_next[i] = 0u;
    const bool stamp = a.dbg && lidx == 0 && tid == 0;
    for (int k = 0; k < npanels && ok; ++k) {
      const int trail = max(0, a.C1 - 16 * k - 16);
      const int lchalf = max(4, 4 * ((trail + 32 * NL - 1) / (32 * NL)));   // trailing columns per workgroup: all 8 NL share the range
      const int lsplit = max(1, (trail + lchalf - 1) / lchalf);
      if (lidx >= lsplit) continue;
      if (stamp) a.dbg[8 * k + 6] = wall_clock64();
      if (!xk_resident_last(ap, k, lidx, lchalf, ubuf, sc, &s_ok)) { ok = false; break; }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) xk_count_in(sync + (XK_PS_L2CNT + k) * 16, sync + (XK_PS_L2FLAG + k) * 16, 1u, (unsigned)lsplit);
      if (stamp) a.dbg[8 * k + 7] = wall_clock64();
    }
    if (stamp) a.dbg[515] = wall_clock64();
  }
  if (!ok && tid == 0) a.status[1] = (int)__hip_atomic_load(ab, XK_RLX_AGENT);
}
