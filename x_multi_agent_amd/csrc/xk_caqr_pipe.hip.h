// xk_caqr_pipe.hip.h -- QR compression (VioUpdater::applyQRDecomposition, src/x/vio/vio_updater.cpp:487-512) in ONE launch,
// the row stack resident in registers, and the three levels of the CAQR tree on WORKGROUPS OF THEIR OWN that run as a pipeline.
//
// What xk_caqr_resident (round 2) could not get past: every tile workgroup ran its tile step, then a first-level merge item,
// then waited for its pivot strip -- tile step -> first level -> tile step is a dependency cycle, and with both halves on the
// same CU it costs their SUM (10 + 10 us of reflector steps + 4 hand-offs = 30 us per 16-column panel).  Here:
//
//   * XCD x = 23 TILE workgroups (a fat tile of <= 128 rows each: 4 lanes per column x 32 rows per lane, the column <-> thread
//     map absolute) + 8 FIRST-LEVEL workgroups (one group per XCD: 23 strips + the pending strip, 16 lanes per column,
//     16 panel + trail/8 trailing columns each) + 1 LAST-LEVEL workgroup (the 8 of them share the trailing columns by ABSOLUTE
//     column slices of C1P/8, so what they keep stays in their registers).
//   * The strips of a merge are upper triangular in the panel columns: reflector j of the level above is zero in rows > j of
//     every strip, so its steps 0..7 need rows 0..7 only.  A tile publishes rows 0..7 of its pivot strip after its step 7 and
//     rows 8..15 after step 15; the first level runs its steps 0..7 BESIDE the tile's steps 8..15, publishes rows 0..7 of the
//     root after them, and the last level follows in the same way.  The cycle tile -> first level -> tile is now
//     16 tile steps + 8 first-level steps + the hand-offs.
//   * The other cycle of the round-2 kernel (last level -> pending strip -> first level of the next panel -> last level) is
//     gone: the last level keeps what it leaves of a panel's roots for ONE MORE panel (two generations: 8 dense strips of the
//     previous panel + 8 triangular roots of this one), and only then sends them down as the pending strip of the first
//     level two panels later -- by then they have been through both panels' eliminations.  Row 0 of every merge (register 0
//     = the pending strip, zero when there is none) is the pivot strip, so no tile is special: every tile gets its own strip
//     back, no leaders, no holes.
// Hand-offs inside an XCD: plain stores + s_waitcnt vmcnt(0) + sc1 loads (the XCD's L2); across XCDs: write-through stores
// into per-panel slabs that are never reused inside a launch.  All spins are bounded and look at an abort word; a launch
// that gives up is redone by the multi-launch schedule (xk_api.hip).  DESIGN 3.2.2 has the anatomy and the numbers.
#pragma once
#include <hip/hip_runtime.h>

#include "xk_caqr_resident.hip.h"

#define XK_PIPE_THREADS 768
#define XK_PIPE_RPL 32              // rows per lane of a fat tile: 4 x 32 = 128 rows
#define XK_PIPE_NT 23               // tile workgroups per XCD
#define XK_PIPE_NM 8                // first-level workgroups per XCD   (NT + NM + 1 = 32 = CUs of an XCD)
#ifndef XK_PIPE_ARR
#define XK_PIPE_ARR 9               // the iteration whose barrier carries "rows 0..7 are out" (their stores drain meanwhile)
#endif

enum {
  XP_CENSUS = 0,                    // [8] workgroups per XCD
  XP_ABORT = 9,
  // per XCD, monotonic (an arrival counter + the generation word the waiters poll; generation = panel + 1)
  XP_TA_CNT = 16, XP_TA_GEN = 24,   // tiles: rows 0..7 of the pivot strips are out
  XP_TB_CNT = 32, XP_TB_GEN = 40,   // tiles: rows 8..15
  XP_MB_CNT = 48, XP_MB_GEN = 56,   // first level: the strips are back
  // per panel (XCDs run up to two panels apart)
  XP_X1A_CNT = 64, XP_X1A_FLAG = 64 + XK_PERSIST_MAXP,             // first-level items whose root rows 0..7 are out
  XP_X1B_CNT = 64 + 2 * XK_PERSIST_MAXP, XP_X1B_FLAG = 64 + 3 * XK_PERSIST_MAXP,   // rows 8..15
  XP_P_CNT = 64 + 4 * XK_PERSIST_MAXP, XP_P_FLAG = 64 + 5 * XK_PERSIST_MAXP,       // last-level workgroups whose pending strips are out
  XP_ON_CNT = 64 + 6 * XK_PERSIST_MAXP, XP_ON_FLAG = 64 + 7 * XK_PERSIST_MAXP,     // ... whose share of the next panel's columns is out
  XP_WORDS = 64 + 8 * XK_PERSIST_MAXP
};

struct XkCaqrPipeArgs {
  const double *A;        // tiles [ntiles][64][C1P] row-major as the per-feature kernels wrote them (read once)
  const int *tile_rows;   // valid rows per 64-row slot (0 = rejected track)
  const int *rowmap;      // valid row g -> physical row of A, [R]
  int R, TR;              // valid rows in total, rows per fat tile (<= 128)
  int C1P, C1;
  double *Rout;           // [C1P][C1P] row-major
  double *S;              // [8 NT][16][C1P] pivot strips (XCD-local)
  double *PB;             // [8 NT][16][16]  their panel blocks
  double *X1, *X1P;       // [panels][8][16][C1P] / [panels][8][16][16]: the root of XCD x (write-through)
  double *X2;             // [panels][8][16][C1P]: strips the last level sends down to XCD s (write-through)
  double *ON;             // [panels][8][16][16]: the NEXT panel's columns of the strips the last level keeps
  unsigned *sync, *sync_next;
  int *status;
  long long *dbg;
};
typedef const XkCaqrPipeArgs __attribute__((address_space(4))) *XkPipeArgsPtr;
__device__ __forceinline__ XkCaqrPipeArgs xk_pipe_args(XkPipeArgsPtr ap) {
  XkCaqrPipeArgs a;
  __builtin_memcpy(&a, (const void *)ap, sizeof(a));
  return a;
}

__device__ __forceinline__ void xk_pipe_arrive(unsigned *cnt, unsigned *gen, unsigned n, unsigned epoch) {
  const unsigned old = __hip_atomic_fetch_add(cnt, 1u, XK_RLX_AGENT);
  if (old == n * epoch - 1u) __hip_atomic_store(gen, epoch, XK_RLX_AGENT);
}
// Hides a per-lane constant from loop-invariant code motion: the sixteen unrolled steps of a panel derive 0/1 masks, LDS
// addresses and predicates from (part, column); hoisted out of the panel loop they are ~60 live registers that end up in
// scratch, and the reloads land on the owner's chain.  Laundered once per panel, they are re-derived where they are used.
__device__ __forceinline__ int xk_launder(int v) {
  asm volatile("" : "+v"(v));
  return v;
}
// Strips travel in blocks of 4 columns x 16 rows: element (column c, row r) of a strip sits at (c / 4) 64 + 4 r + c % 4.  A wave of
// the merge layout (4 columns x 16 rows) then moves one contiguous 512-byte block per strip; row-major strips cost it sixteen
// 32-byte pieces per instruction, and the strip loads / stores of a first-level workgroup were 4.6 + 4.5 us of its panel.
__device__ __forceinline__ size_t xk_blk(int c, int r) { return (size_t)(c >> 2) * 64 + (size_t)r * 4 + (size_t)(c & 3); }
// thread 0 polls, everybody learns the verdict
__device__ __forceinline__ bool xk_pipe_wait(unsigned *word, unsigned target, unsigned *ab, unsigned reason, unsigned *s_ok) {
  if (threadIdx.x == 0) *s_ok = xk_spin_ge(word, target, ab, reason) ? 1u : 0u;
  __syncthreads();
  const bool ok = *s_ok != 0u;
  __syncthreads();                  // (s_ok is rewritten by the next wait)
  return ok;
}

// the 16 steps of a tile's panel in two halves, one-reflector look-ahead (xk_caqr_form / xk_caqr_apply, tile layout)
template <int RPL>
__device__ __forceinline__ void xk_pipe_tsteps_a(double (&b)[RPL], int rel, bool live, int part, int nsteps, double *ubuf, double *sc) {
  xk_caqr_form<0, 4, RPL>(b, rel, part, ubuf, sc);
  __syncthreads();
#define XK_IT(K)                                                                                  \
  if (K < nsteps) { xk_caqr_apply<K - 1, 4, RPL>(b, rel, live, part, ubuf, sc); xk_caqr_form<K, 4, RPL>(b, rel, part, ubuf, sc); __syncthreads(); } \
  else if (K == nsteps) xk_caqr_apply<K - 1, 4, RPL>(b, rel, live, part, ubuf, sc);
  XK_IT(1) XK_IT(2) XK_IT(3) XK_IT(4) XK_IT(5) XK_IT(6) XK_IT(7)
#undef XK_IT
  if (nsteps >= 8) xk_caqr_apply<7, 4, RPL>(b, rel, live, part, ubuf, sc);
}
// second half; the barrier of iteration ARR is where thread 0 tells the first level that rows 0..7 are out (every wave has
// drained its stores by then).  Returns whether that happened (it does not for a panel of <= ARR columns).
template <int RPL, int ARR>
__device__ __forceinline__ bool xk_pipe_tsteps_b(double (&b)[RPL], int rel, bool live, int part, int nsteps, double *ubuf, double *sc,
                                                 unsigned *cnt, unsigned *gen, unsigned n, unsigned epoch) {
  if (8 < nsteps) xk_caqr_form<8, 4, RPL>(b, rel, part, ubuf, sc);
  __syncthreads();
#define XK_IT(K)                                                                                  \
  if (K < nsteps) {                                                                               \
    xk_caqr_apply<K - 1, 4, RPL>(b, rel, live, part, ubuf, sc); xk_caqr_form<K, 4, RPL>(b, rel, part, ubuf, sc); \
    if (K == ARR) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                \
    __syncthreads();                                                                              \
    if (K == ARR && threadIdx.x == 0) xk_pipe_arrive(cnt, gen, n, epoch);                         \
  } else if (K == nsteps) xk_caqr_apply<K - 1, 4, RPL>(b, rel, live, part, ubuf, sc);
  XK_IT(9) XK_IT(10) XK_IT(11) XK_IT(12) XK_IT(13) XK_IT(14) XK_IT(15)
#undef XK_IT
  if (nsteps == 16) xk_caqr_apply<15, 4, RPL>(b, rel, live, part, ubuf, sc);
  return nsteps > ARR;
}

// ---- role T: one fat tile in registers for the whole factorisation
template <int RPL>
__device__ __noinline__ bool xk_pipe_tile(XkPipeArgsPtr ap, int xcc, int slot, long long t_entry, double *ubuf, double *sc, unsigned *s_ok) {
  constexpr int NT = XK_PIPE_NT, NM = XK_PIPE_NM;
  const XkCaqrPipeArgs a = xk_pipe_args(ap);
  unsigned *sync = a.sync, *ab = sync + XP_ABORT * 16;
  const int tid = threadIdx.x;
  const int j = xcc * NT + slot;                           // my fat tile: valid rows [j TR, (j + 1) TR)
  const int cabs = tid >> 2, part_ = tid & 3;              // ABSOLUTE column of this thread, all panels
  const bool mine = cabs < a.C1;
  const int npanels = (a.C1 + 15) / 16;
  const bool stamp = a.dbg && xcc == 0 && slot == 1 && tid == 0;
  double b[RPL];
  {   // the one pass over the stack: gather my rows through the row map
    const int g0 = j * a.TR + part_ * RPL, gend = min(min((j + 1) * a.TR, g0 + RPL), a.R);
#pragma unroll
    for (int r = 0; r < RPL; ++r) {
      const int g = g0 + r;
      double v = 0.0;
      if (mine && g < gend) {
        const int pr = a.rowmap[g];
        const double x = a.A[(size_t)pr * a.C1P + cabs];
        v = (a.tile_rows[pr >> 6] > 0) ? x : 0.0;
      }
      b[r] = v;
    }
  }
  double *myS = a.S + (size_t)j * 16 * a.C1P;
  double *myPB = a.PB + (size_t)j * 256;
  if (stamp) { a.dbg[1536] = t_entry; a.dbg[1537] = wall_clock64(); }
  for (int k = 0; k < npanels; ++k) {
    const int c0 = 16 * k;
    const int part = xk_launder(part_);
    const int rel = xk_launder(cabs) - c0;
    const int nsteps = (a.C1 - c0 < 16) ? a.C1 - c0 : 16;
    const bool hot = (rel >> 4) == 0;
    const bool pub = mine && part == 0 && rel >= 0;
    if (stamp) a.dbg[16 * k + 0] = wall_clock64();
    if (hot) __builtin_amdgcn_s_setprio(3);
    xk_pipe_tsteps_a<RPL>(b, rel, mine, part, nsteps, ubuf, sc);
    bool told = false;
    if (nsteps > 8) {
      // rows 0..7 of the pivot strip are final: out they go, the first level starts on them while steps 8..15 run here
      if (pub) {
        if (rel < 16) {
          double *pb = xk_opaque(myPB + xk_blk(rel, 0));
#pragma unroll
          for (int r = 0; r < 8; ++r) pb[r * 4] = (r > rel) ? 0.0 : b[r];
        } else {
          double *ps = xk_opaque(myS + xk_blk(cabs, 0));
#pragma unroll
          for (int r = 0; r < 8; ++r) ps[r * 4] = b[r];
        }
      }
      if (stamp) a.dbg[16 * k + 1] = wall_clock64();
      told = xk_pipe_tsteps_b<RPL, XK_PIPE_ARR>(b, rel, mine, part, nsteps, ubuf, sc, sync + (XP_TA_CNT + xcc) * 16, sync + (XP_TA_GEN + xcc) * 16,
                                                (unsigned)NT, (unsigned)(k + 1));
    }
    if (hot) __builtin_amdgcn_s_setprio(0);
    if (stamp) a.dbg[16 * k + 2] = wall_clock64();
    if (pub) {
      const int r0 = (nsteps > 8) ? 8 : 0;
      if (rel < 16) {
        double *pb = xk_opaque(myPB + xk_blk(rel, 0));
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (r >= r0) pb[r * 4] = (r > rel) ? 0.0 : b[r];
      } else {
        double *ps = xk_opaque(myS + xk_blk(cabs, 0));
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (r >= r0) ps[r * 4] = b[r];
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      if (!told) xk_pipe_arrive(sync + (XP_TA_CNT + xcc) * 16, sync + (XP_TA_GEN + xcc) * 16, (unsigned)NT, (unsigned)(k + 1));
      xk_pipe_arrive(sync + (XP_TB_CNT + xcc) * 16, sync + (XP_TB_GEN + xcc) * 16, (unsigned)NT, (unsigned)(k + 1));
    }
    if (stamp) a.dbg[16 * k + 3] = wall_clock64();
    if (k + 1 == npanels) break;
    // my strip comes back from the first level (trailing columns of the NEXT panels only: everything up to c0 + 15 is finished)
    if (!xk_pipe_wait(sync + (XP_MB_GEN + xcc) * 16, (unsigned)(k + 1), ab, 2u, s_ok)) return false;
    if (stamp) a.dbg[16 * k + 4] = wall_clock64();
    if (mine && rel >= 16 && part == 0) {
      const double *ps = xk_opaque(myS + xk_blk(cabs, 0));
#pragma unroll
      for (int r = 0; r < 16; ++r) b[r] = xk_ld_sc1(ps + r * 4);
    }
    if (stamp) {
      double sink = 0;
      for (int r = 0; r < 16; ++r) sink += b[r];
      asm volatile("" ::"v"(sink));
      a.dbg[16 * k + 5] = wall_clock64();
    }
  }
  (void)NM;
  if (stamp) a.dbg[1538] = wall_clock64();
  return true;
}

// ---- role M: the first merge level of this XCD's 23 strips (+ the pending strip), 16 panel + <= 32 trailing columns per workgroup
// 16 lanes per column: lane p = row p of every strip, register 0 = the pending strip = the pivot strip, register 1 + t = tile t
__device__ __noinline__ bool xk_pipe_first(XkPipeArgsPtr ap, int xcc, int item, double *ubuf, double *sc, unsigned *s_ok) {
  constexpr int NT = XK_PIPE_NT, NM = XK_PIPE_NM, RM = NT + 1, NP = 16;
  static_assert(RM % 2 == 0, "register count of the merge layout must be even");
  const XkCaqrPipeArgs a = xk_pipe_args(ap);
  unsigned *sync = a.sync, *ab = sync + XP_ABORT * 16;
  const int tid = threadIdx.x;
  const int cidx_ = tid / NP, part_ = tid & (NP - 1);
  const bool panel = cidx_ < 16;
  const int npanels = (a.C1 + 15) / 16;
  const int base = xcc * NT;
  const size_t SS = (size_t)16 * a.C1P;                   // doubles per strip
  const bool stamp = a.dbg && xcc == 0 && item == 0 && tid == 0;
  for (int k = 0; k < npanels; ++k) {
    const int c0 = 16 * k, trail = max(0, a.C1 - c0 - 16);
    const int cidx = xk_launder(cidx_), part = xk_launder(part_);
    const int mch = min(32, 4 * ((trail + 4 * NM - 1) / (4 * NM)));     // trailing columns per item (whole quarter-waves)
    const bool active = item == 0 || item * mch < trail;
    const int col = panel ? c0 + cidx : c0 + 16 + item * mch + (cidx - 16);
    const bool mine = active && col < a.C1 && (panel || cidx - 16 < mch);
    const int nsteps = (a.C1 - c0 < 16) ? a.C1 - c0 : 16;
    const size_t slab = (size_t)k * 8 + xcc;
    double b[RM];
    // the pending strip: what the last level left of the roots of panel k - 2 (XCD 0 gets none)
    b[0] = 0.0;
    if (k >= 1 && xcc != 0) {
      if (!xk_pipe_wait(sync + (XP_P_FLAG + k - 1) * 16, 1u, ab, 4u, s_ok)) return false;
      if (mine) b[0] = xk_ld_sc1(a.X2 + ((size_t)(k - 1) * 8 + xcc) * SS + xk_blk(col, part));
    }
    if (!xk_pipe_wait(sync + (XP_TA_GEN + xcc) * 16, (unsigned)(k + 1), ab, 6u, s_ok)) return false;
    if (stamp) a.dbg[512 + 16 * k + 0] = wall_clock64();
    const size_t lane_off = panel ? xk_blk(cidx, part) : xk_blk(col, part);
    const size_t strip_step = panel ? 256 : SS;
    double *g0 = (panel ? a.PB + (size_t)base * 256 : a.S + (size_t)base * SS) + lane_off;
    if (active) {
      {
        double *g = xk_opaque(g0);
#pragma unroll
        for (int s = 1; s < RM; ++s) b[s] = (mine && part < 8) ? xk_ld_sc1(g + (size_t)(s - 1) * strip_step) : 0.0;
      }
      if (stamp) {
        double sink = 0;
        for (int s = 0; s < RM; ++s) sink += b[s];
        asm volatile("" ::"v"(sink));
        a.dbg[512 + 16 * k + 1] = wall_clock64();
      }
      if (panel) __builtin_amdgcn_s_setprio(3);
      xk_res_msteps_a<RM>(b, cidx, mine, part, nsteps, ubuf, sc);
      if (panel) __builtin_amdgcn_s_setprio(0);
      // rows 0..7 of the root are final: out they go (write-through: the last level sits on other XCDs)
      if (mine && part < 8) {
        if (panel) {
          if (item == 0) xk_st_sc1(a.X1P + slab * 256 + xk_blk(cidx, part), (part > cidx) ? 0.0 : b[0]);
        } else {
          xk_st_sc1(a.X1 + slab * SS + xk_blk(col, part), b[0]);
        }
      }
      if (stamp) a.dbg[512 + 16 * k + 2] = wall_clock64();
    }
    if (!xk_pipe_wait(sync + (XP_TB_GEN + xcc) * 16, (unsigned)(k + 1), ab, 7u, s_ok)) return false;
    if (stamp) a.dbg[512 + 16 * k + 3] = wall_clock64();
    unsigned *cnt_a = sync + (XP_X1A_CNT + k) * 16, *flag_a = sync + (XP_X1A_FLAG + k) * 16;
    if (active) {
      if (mine && part >= 8) {
        double *g = xk_opaque(g0);
#pragma unroll
        for (int s = 1; s < RM; ++s) b[s] = xk_ld_sc1(g + (size_t)(s - 1) * strip_step);
      }
      if (panel) __builtin_amdgcn_s_setprio(3);
      // (its first barrier, behind the loads above, is where rows 0..7 of the root are counted in)
      xk_res_msteps_b<RM>(b, cidx, mine, part, nsteps, ubuf, sc, cnt_a, flag_a, 8u * NM);
      if (panel) __builtin_amdgcn_s_setprio(0);
      if (stamp) a.dbg[512 + 16 * k + 4] = wall_clock64();
      if (mine) {
        if (!panel) {
          // the tiles' strips first (the tiles wait for them), then the rest of the root
          double *g = xk_opaque(g0);
#pragma unroll
          for (int s = 1; s < RM; ++s) g[(size_t)(s - 1) * strip_step] = b[s];
          if (part >= 8) xk_st_sc1(a.X1 + slab * SS + xk_blk(col, part), b[0]);
        } else if (item == 0 && part >= 8) {
          xk_st_sc1(a.X1P + slab * 256 + xk_blk(cidx, part), (part > cidx) ? 0.0 : b[0]);
        }
      }
    } else if (tid == 0) {
      xk_count_in(cnt_a, flag_a, 1u, 8u * NM);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      xk_pipe_arrive(sync + (XP_MB_CNT + xcc) * 16, sync + (XP_MB_GEN + xcc) * 16, (unsigned)NM, (unsigned)(k + 1));
      xk_count_in(sync + (XP_X1B_CNT + k) * 16, sync + (XP_X1B_FLAG + k) * 16, 1u, 8u * NM);
    }
    if (stamp) a.dbg[512 + 16 * k + 5] = wall_clock64();
  }
  return true;
}

// ---- role L: the last merge level, one workgroup per XCD: the 8 of them share the trailing columns of the panel and factor
// its 16 columns redundantly.  Register s = the root of XCD s (register 0 is the pivot strip); what is left of registers
// 1..7 goes down to XCD s as the pending strip of its first level in the NEXT panel.  (That is a dependency loop -- last
// level -> first level -> last level -- but with the first level on CUs of its own it is shorter than the tiles' loop:
// rows 8..15 of the roots leave the first level together with the tiles' strips, 8 steps later the pending strips are out,
// and the next first level does not start before its tiles are half way through their panel.)
__device__ __noinline__ bool xk_pipe_last(XkPipeArgsPtr ap, int lidx, double *ubuf, double *sc, unsigned *s_ok) {
  constexpr int NP = 16, RL = 8;
  const XkCaqrPipeArgs a = xk_pipe_args(ap);
  unsigned *sync = a.sync, *ab = sync + XP_ABORT * 16;
  const int tid = threadIdx.x;
  const int cidx_ = tid / NP, part_ = tid & (NP - 1);
  const bool panel = cidx_ < 16;
  const int npanels = (a.C1 + 15) / 16;
  const bool stamp = a.dbg && lidx == 0 && tid == 0;
  const size_t SS = (size_t)16 * a.C1P;                   // doubles per strip
  for (int k = 0; k < npanels; ++k) {
    const int c0 = 16 * k, trail = max(0, a.C1 - c0 - 16);
    const int lchalf = max(4, 4 * ((trail + 31) / 32));   // trailing columns per workgroup: all 8 share the range
    const int lsplit = max(1, (trail + lchalf - 1) / lchalf);
    if (lidx >= lsplit) continue;
    const int cidx = xk_launder(cidx_), part = xk_launder(part_);
    const int col = panel ? c0 + cidx : c0 + 16 + lidx * lchalf + (cidx - 16);
    const bool mine = col < a.C1 && (panel || cidx - 16 < lchalf);
    const int nsteps = (a.C1 - c0 < 16) ? a.C1 - c0 : 16;
    const double *src = panel ? a.X1P + (size_t)k * 8 * 256 + xk_blk(cidx, part) : a.X1 + (size_t)k * 8 * SS + xk_blk(col, part);
    const size_t sstep = panel ? 256 : SS;
    double b[RL];
    if (!xk_pipe_wait(sync + (XP_X1A_FLAG + k) * 16, 1u, ab, 5u, s_ok)) return false;
    if (stamp) a.dbg[1024 + 16 * k + 0] = wall_clock64();
#pragma unroll
    for (int s = 0; s < RL; ++s) b[s] = (mine && part < 8) ? xk_ld_sc1(src + s * sstep) : 0.0;
    if (panel) __builtin_amdgcn_s_setprio(3);
    xk_res_msteps_a<RL>(b, panel ? cidx : 16, mine, part, nsteps, ubuf, sc);
    if (panel) __builtin_amdgcn_s_setprio(0);
    if (stamp) a.dbg[1024 + 16 * k + 1] = wall_clock64();
    if (nsteps > 8) {
      if (!xk_pipe_wait(sync + (XP_X1B_FLAG + k) * 16, 1u, ab, 5u, s_ok)) return false;
      if (stamp) a.dbg[1024 + 16 * k + 2] = wall_clock64();
      if (mine && part >= 8) {
#pragma unroll
        for (int s = 0; s < RL; ++s) b[s] = xk_ld_sc1(src + s * sstep);
      }
      if (panel) __builtin_amdgcn_s_setprio(3);
      xk_res_msteps_b<RL>(b, panel ? cidx : 16, mine, part, nsteps, ubuf, sc, nullptr, nullptr, 0u);
      if (panel) __builtin_amdgcn_s_setprio(0);
    }
    if (stamp) a.dbg[1024 + 16 * k + 3] = wall_clock64();
    if (mine) {
      if (panel) {
        if (lidx == 0 && c0 + part < a.C1) a.Rout[(size_t)(c0 + part) * a.C1P + col] = (part > cidx) ? 0.0 : b[0];
      } else {
        if (k + 1 < npanels) {
          double *dst = a.X2 + (size_t)k * 8 * SS + xk_blk(col, part);
#pragma unroll
          for (int s = 1; s < RL; ++s) xk_st_sc1(dst + s * SS, b[s]);
        }
        if (c0 + part < a.C1) a.Rout[(size_t)(c0 + part) * a.C1P + col] = b[0];
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0 && k + 1 < npanels) xk_count_in(sync + (XP_P_CNT + k) * 16, sync + (XP_P_FLAG + k) * 16, 1u, (unsigned)lsplit);
    if (stamp) a.dbg[1024 + 16 * k + 4] = wall_clock64();
  }
  if (stamp) a.dbg[1539] = wall_clock64();
  return true;
}

__global__ __launch_bounds__(XK_PIPE_THREADS) void xk_caqr_pipe(XkCaqrPipeArgs a) {
  constexpr int RPL = XK_PIPE_RPL, NT = XK_PIPE_NT, NM = XK_PIPE_NM, RM = NT + 1;
  constexpr int LDS_T = 2 * 4 * (RPL + 2), LDS_M = 2 * 16 * (RM + 2), LDS_L = 2 * 16 * 10;
  constexpr int LDS_MAX = LDS_T > LDS_M ? (LDS_T > LDS_L ? LDS_T : LDS_L) : (LDS_M > LDS_L ? LDS_M : LDS_L);
  __shared__ __attribute__((aligned(16))) double ubuf[LDS_MAX];
  __shared__ __attribute__((aligned(16))) double sc[2 * 4];
  __shared__ unsigned s_slot, s_ok;
  unsigned *sync = a.sync, *ab = sync + XP_ABORT * 16;
  const XkPipeArgsPtr ap = (XkPipeArgsPtr)__builtin_amdgcn_kernarg_segment_ptr();
  const unsigned xcc = xk_xcc_id();
  const long long t_entry = a.dbg ? wall_clock64() : 0;
  // placement census as in xk_caqr_resident: a workgroup takes the next slot of the XCD it finds itself on; a 33rd arrival
  // on one XCD raises the abort word
  if (threadIdx.x == 0) {
    const unsigned sl = __hip_atomic_fetch_add(sync + (XP_CENSUS + xcc) * 16, 1u, XK_RLX_AGENT);
    const bool bad = sl >= 32u || gridDim.x != 256u;
    if (bad) { __hip_atomic_store(ab, 3u, XK_RLX_AGENT); a.status[1] = 3; }
    s_slot = sl;
    s_ok = bad ? 0u : 1u;
  }
  __syncthreads();
  if (!s_ok) return;
  const int slot = __builtin_amdgcn_readfirstlane((int)s_slot);
  __syncthreads();
  bool ok;
  if (slot < NT) ok = xk_pipe_tile<RPL>(ap, (int)xcc, slot, t_entry, ubuf, sc, &s_ok);
  else if (slot < NT + NM) ok = xk_pipe_first(ap, (int)xcc, slot - NT, ubuf, sc, &s_ok);
  else {
    // nothing to do until the first roots arrive: leave the other set of sync words zeroed for the next launch
    for (int i = (int)xcc * XK_PIPE_THREADS + threadIdx.x; i < XP_WORDS * 16; i += 8 * XK_PIPE_THREADS) a.sync_next[i] = 0u;
    ok = xk_pipe_last(ap, (int)xcc, ubuf, sc, &s_ok);
  }
  if (!ok && threadIdx.x == 0) a.status[1] = (int)__hip_atomic_load(ab, XK_RLX_AGENT);
}
