// xk_caqr_resident_blocked.hip.h -- EXPERIMENT (not part of libxk.so unless built with -DXK_CAQR_BLOCKED_EXPERIMENT):
// the register-resident kernel with the blocked tile step of xk_caqr_blocked.hip.h selected by a template flag.
// Measured on MI355X (tools/exp/qr_variants.sh, same box): QR 0.504 ms blocked vs 0.476 ms reflector-by-reflector --
// the single panel wave's 16-step DPP chain (7.7 us) plus Gram matrix and T (1.4 + 6 us with a back substitution that
// waits for one LDS read per term) is longer than the 16 LDS-broadcast steps it replaces, and merely having both variants
// in one templated kernel made the plain one 18 % slower (0.566 ms).  Kept for the record and for the next attempt
// (block forward substitution on the matrix cores instead of T^-1).
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
//     16 panel + 32 trailing columns per workgroup (768 threads, 168 VGPRs), at most one item per CU.
//   * Group leaders: the root strip leaves for the last level (X1) and comes back one panel later as the pending strip
//     (X2 -> first level -> Hq).  Meanwhile the leader needs another pivot strip in the registers of its part-0 lanes:
//     it swaps registers 0..15 of its part-0 and part-1 lanes (one quad-permute DPP move per register) and zeroes the
//     "away" half until the returned rows are loaded into it.
// Synchronisation, placement census, bounded spins and the cross-XCD slabs are those of xk_caqr_persist.hip.h.
#pragma once
#define xk_caqr_resident xk_caqr_resident_blk
#define xk_resident_merge1 xk_resident_merge1_blk
#define xk_resident_last xk_resident_last_blk
#define XkCaqrResidentArgs XkCaqrResidentBlkArgs
#define XkResidentArgsPtr XkResidentBlkArgsPtr
#define xk_resident_args xk_resident_blk_args
#include <hip/hip_runtime.h>

#include "xk_caqr_blocked.hip.h"
#include "xk_caqr_persist.hip.h"

#define XK_RES_THREADS 768          // 12 waves, one workgroup per CU: up to 168 VGPRs (a 1024-thread version spilled 271)
#define XK_RES_RPL 24               // rows per lane: fat tiles of 4 x 24 = 96 rows
#define XK_RES_NT 31                // tile workgroups per XCD (the 32nd runs the last merge level)

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
  int *status;
  long long *dbg;
};
typedef const XkCaqrResidentArgs __attribute__((address_space(4))) *XkResidentArgsPtr;
__device__ __forceinline__ XkCaqrResidentArgs xk_resident_args(XkResidentArgsPtr ap) {
  XkCaqrResidentArgs a;
  __builtin_memcpy(&a, (const void *)ap, sizeof(a));
  return a;
}

// first merge level of one group, 16 lanes per column (lane p = row p of every strip, register s = strip s, register
// NS = the pending strip): 16 panel + 32 trailing columns per workgroup.  Strips come from S / PB1, the root goes to
// X1 / X1P, the other strips back to S, the pending strip's rest to Hq.
template <int RPL>
__device__ __noinline__ void xk_resident_merge1(XkResidentArgsPtr ap, int k, int gid, int base, int nstrips, int split, int MCH,
                                                double *ubuf, double *sc) {
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
#define XK_STEP(K) if (K < nsteps) xk_caqr_mstep<K, RPL>(b, cidx, mine, part, ubuf, sc);
  XK_STEP(0) XK_STEP(1) XK_STEP(2) XK_STEP(3) XK_STEP(4) XK_STEP(5) XK_STEP(6) XK_STEP(7)
  XK_STEP(8) XK_STEP(9) XK_STEP(10) XK_STEP(11) XK_STEP(12) XK_STEP(13) XK_STEP(14) XK_STEP(15)
#undef XK_STEP
  if (mstamp) { a.dbg[256 + 8 * k + 3] = wall_clock64(); a.dbg[256 + 8 * k + 5] = clock64(); }
  if (!mine) return;
  if (panel) {
    if (split == 0) xk_st_sc1(a.X1P + ((size_t)k * XK_PERSIST_MAXG + gid) * 256 + part * 16 + cidx, (part > cidx) ? 0.0 : b[0]);
  } else {
    xk_st_sc1(a.X1 + (((size_t)k * XK_PERSIST_MAXG + gid) * 16 + part) * a.C1P + col, b[0]);
#pragma unroll
    for (int s = 1; s < NS; ++s)
      if (s < nstrips) g0[(size_t)s * strip_step] = b[s];
    if (pend_ok) pend_dst[0] = b[NS];
  }
}

// last merge level, 16 lanes per column (lane p = row p of every root strip): 16 roots -> 16 rows of R, the rest -> X2
__device__ __noinline__ void xk_resident_last(XkResidentArgsPtr ap, int k, int split, int lchalf, double *ubuf, double *sc) {
  constexpr int NP = 16, RPL = 16;
  const XkCaqrResidentArgs a = xk_resident_args(ap);
  const int c0 = 16 * k;
  const int cidx = (int)threadIdx.x / NP, part = threadIdx.x & (NP - 1);
  const bool panel = cidx < 16;
  const int col = panel ? c0 + cidx : c0 + 16 + split * lchalf + (cidx - 16);
  const bool mine = col < a.C1 && (panel || cidx - 16 < lchalf);
  const size_t slab = (size_t)k * XK_PERSIST_MAXG;
  double b[RPL];
#pragma unroll
  for (int s = 0; s < RPL; ++s)
    b[s] = !mine ? 0.0 : panel ? xk_ld_sc1(a.X1P + (slab + s) * 256 + part * 16 + cidx)
                               : xk_ld_sc1(a.X1 + ((slab + s) * 16 + part) * a.C1P + col);
  const int nsteps = (a.C1 - c0 < 16) ? a.C1 - c0 : 16;
#define XK_STEP(K) if (K < nsteps) xk_caqr_mstep<K, RPL>(b, cidx, mine, part, ubuf, sc);
  XK_STEP(0) XK_STEP(1) XK_STEP(2) XK_STEP(3) XK_STEP(4) XK_STEP(5) XK_STEP(6) XK_STEP(7)
  XK_STEP(8) XK_STEP(9) XK_STEP(10) XK_STEP(11) XK_STEP(12) XK_STEP(13) XK_STEP(14) XK_STEP(15)
#undef XK_STEP
  if (!mine) return;
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
}

// BLK = false: the reflector-by-reflector tile step of xk_linalg.hip.h (rows blocked over the 4 lanes of a column);
// BLK = true:  the blocked step of xk_caqr_blocked.hip.h (rows interleaved, lane = part * 16 + column within the wave).
template <bool BLK>
__global__ __launch_bounds__(XK_RES_THREADS) void xk_caqr_resident(XkCaqrResidentArgs a) {
  constexpr int RPL = XK_RES_RPL, NT = XK_RES_NT, RM1 = 18, A1 = 16, G = 2;
  constexpr int LDS_T = 2 * 4 * (RPL + 2), LDS_M = 2 * 16 * (RM1 + 2), LDS_L = 2 * 16 * 18;
  constexpr int LDS_MAX = LDS_T > LDS_M ? (LDS_T > LDS_L ? LDS_T : LDS_L) : (LDS_M > LDS_L ? LDS_M : LDS_L);
  __shared__ __attribute__((aligned(16))) double ubuf[LDS_MAX];
  __shared__ __attribute__((aligned(16))) double sc[2 * 4];
  __shared__ __attribute__((aligned(16))) double blds[BLK ? XK_BLK_LDS : 2];
  __shared__ unsigned s_slot, s_nx, s_ok;
  unsigned *sync = a.sync, *ab = sync + XK_PS_ABORT * 16;
  const XkResidentArgsPtr ap = (XkResidentArgsPtr)__builtin_amdgcn_kernarg_segment_ptr();
  const unsigned xcc = xk_xcc_id();
  if (threadIdx.x == 0) {
    s_slot = __hip_atomic_fetch_add(sync + (XK_PS_CENSUS + xcc) * 16, 1u, XK_RLX_AGENT);
    __hip_atomic_fetch_add(sync + XK_PS_TOTAL * 16, 1u, XK_RLX_AGENT);
    bool ok = xk_spin_ge(sync + XK_PS_TOTAL * 16, gridDim.x, ab, 1u);
    const unsigned nx = __hip_atomic_load(sync + (XK_PS_CENSUS + xcc) * 16, XK_RLX_AGENT);
    if (ok && (nx * 8u != gridDim.x || nx != NT + 1)) { __hip_atomic_store(ab, 3u, XK_RLX_AGENT); ok = false; }
    s_nx = nx;
    s_ok = ok ? 1u : 0u;
  }
  __syncthreads();
  if (!s_ok) {
    if (threadIdx.x == 0 && blockIdx.x == 0) a.status[1] = (int)__hip_atomic_load(ab, XK_RLX_AGENT);
    return;
  }
  // (read from LDS, so the compiler takes it for a per-lane value: every pointer derived from it became 64-bit VGPR
  //  arithmetic, hoisted out of the panel loop and spilled -- ~100 dwords per lane, 0.39 GB of scratch traffic per update)
  const int slot = __builtin_amdgcn_readfirstlane((int)s_slot);
  const int npanels = (a.C1 + 15) / 16;
  unsigned epoch = 0;
  bool ok = true;
  const int tid = threadIdx.x;
  if (slot < NT) {
    // ---- role T: one fat tile in registers + a share of the XCD's first-level merges
    const int j = (int)xcc * NT + slot;                      // my fat tile: valid rows [j TR, (j + 1) TR)
    // tile layout: 4 lanes per column, 768 threads = 192 columns; wave w holds columns 16 w .. 16 w + 15 either way
    const int wv = tid >> 6, lane = tid & 63;
    const int part = BLK ? lane >> 4 : tid & 3;
    const int cidx = BLK ? 16 * wv + (lane & 15) : tid >> 2;
    const bool tlane = true;
    const int cabs = cidx;                                   // ABSOLUTE column of this thread, all panels
    const bool mine = tlane && cabs < a.C1;
    const bool leader = (slot % A1) == 0;
    const int grp = slot / A1;
    const int gsize = min(A1, NT - grp * A1), mygid = (int)xcc * G + grp;
    const bool stamp = a.dbg && xcc == 0 && slot == 1 && tid == 0;
    double b[RPL];
    {   // the one pass over the stack: gather my rows through the row map
      const int g0 = j * a.TR + (BLK ? part : part * RPL), gend = min((j + 1) * a.TR, a.R);
#pragma unroll
      for (int r = 0; r < RPL; ++r) {
        const int g = g0 + (BLK ? 4 * r : r);                // register r = row 4 r + part (blocked) / 24 part + r
        double v = 0.0;
        if (mine && g < gend) {
          const int pr = a.rowmap[g];
          if (a.tile_rows[pr >> 6] > 0) v = a.A[(size_t)pr * a.C1P + cabs];
        }
        b[r] = v;
      }
    }
    double *myS = a.S + (size_t)j * 16 * a.C1P;
    for (int k = 0; k < npanels && ok; ++k) {
      const int c0 = 16 * k, trail = max(0, a.C1 - c0 - 16);
      const int rel = tlane ? cabs - c0 : -1;
      if (stamp) a.dbg[8 * k + 0] = wall_clock64();
      const int nsteps = (a.C1 - c0 < 16) ? a.C1 - c0 : 16;
      if (BLK) {
        if (leader && k > 0) {
          // the strip that was my pivot strip (rows 0..15 = registers 0..3) is away at the last level: rows 16..31 take over
#pragma unroll
          for (int q = 0; q < 4; ++q) { b[q] = b[q + 4]; b[q + 4] = 0.0; }
        }
        if (stamp) a.dbg[256 + 8 * k + 0] = wall_clock64();
        if (wv == k) xk_blk_panel_wave(b, lane, nsteps, blds, (a.dbg && xcc == 0 && slot == 1) ? a.dbg + 512 + 8 * k : nullptr);
        else if (wv > k) xk_blk_trailing_wave(b, lane, blds);
        else { __syncthreads(); __syncthreads(); }
        if (stamp) a.dbg[256 + 8 * k + 1] = wall_clock64();
        if (mine && rel >= 0) {                              // hand the pivot strip over: rows 0..15 = registers 0..3
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int row = 4 * q + part;
            if (rel < 16) a.PB1[((size_t)j * 16 + row) * 16 + rel] = (row > rel) ? 0.0 : b[q];
            else myS[(size_t)row * a.C1P + cabs] = b[q];
          }
        }
      } else {
      if (leader && k > 0 && tlane) {
        // the strip that was my pivot strip is away at the last level: the rows part 1 kept become the new pivot strip
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const double sw = xk_dpp_quad<0xE1>(b[r]);         // quad_perm [1,0,2,3]: parts 0 and 1 trade places
          b[r] = (part == 0) ? sw : (part == 1) ? 0.0 : b[r];
        }
      }
      if (stamp) a.dbg[256 + 8 * k + 0] = wall_clock64();
      xk_caqr_steps<4, RPL>(b, rel, mine, part, nsteps, ubuf, sc);
      // hand the pivot strip over: rows 0..15 of the part-0 lanes.  (Storing row K right after step K -- it is final by
      // then -- so that the stores drain behind the remaining steps was measured: the tile phase goes from 15.4 to 21.5 us,
      // the store data keeps the steps' registers live and the compiler's waits land inside the chain.)
      if (mine && part == 0 && rel >= 0) {
        if (rel < 16) {
#pragma unroll
          for (int r = 0; r < 16; ++r) a.PB1[((size_t)j * 16 + r) * 16 + rel] = (r > rel) ? 0.0 : b[r];
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) myS[(size_t)r * a.C1P + cabs] = b[r];
        }
      }
      if (stamp) a.dbg[256 + 8 * k + 1] = wall_clock64();
      }
      if (stamp) a.dbg[8 * k + 1] = wall_clock64();
      // the two first-level groups of an XCD are independent of each other: each has its own barrier and runs its own
      // merge items (a smaller barrier waits for fewer stragglers)
      ok = xk_flag_barrier(sync, XK_PS_GBARCNT, XK_PS_GBARGEN, (unsigned)mygid, (unsigned)gsize, ++epoch, &s_ok);
      if (!ok) break;
      if (stamp) a.dbg[8 * k + 2] = wall_clock64();
      // trailing columns per first-level item: as few as keeps G x msplit items within the XCD's 31 workgroups (a step of a
      // 48-column item takes 0.83 us, of a 28-column one 0.5: fewer waves share the LDS and the vector pipe)
      const int mch = min(32, max(8, 4 * ((trail + 4 * 15 - 1) / (4 * 15))));   // <= 15 items per group of 15 / 16 workgroups
      const int msplit = max(1, (trail + mch - 1) / mch);
      for (int split = slot - grp * A1; split < msplit; split += gsize) {
        const int jg = grp;
        const int base = (int)xcc * NT + jg * A1;
        const int nstrips = gsize;
        if (k > 0) {
          if (tid == 0) s_ok = xk_spin_ge(sync + (XK_PS_L2FLAG + k - 1) * 16, 1u, ab, 4u) ? 1u : 0u;
          __syncthreads();
          if (!s_ok) { ok = false; break; }
        }
        if (stamp) a.dbg[8 * k + 3] = wall_clock64();
        xk_resident_merge1<RM1>(ap, k, (int)xcc * G + jg, base, nstrips, split, mch, ubuf, sc);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) xk_count_in(sync + (XK_PS_X1CNT + k) * 16, sync + (XK_PS_X1FLAG + k) * 16, 1u, 8u * (unsigned)(G * msplit));   // all 16 groups
      }
      if (!ok) break;
      if (stamp) a.dbg[8 * k + 4] = wall_clock64();
      if (k + 1 < npanels) {
        ok = xk_flag_barrier(sync, XK_PS_GBARCNT, XK_PS_GBARGEN, (unsigned)mygid, (unsigned)gsize, ++epoch, &s_ok);
        if (!ok) break;
        // my strip comes back (trailing columns of the NEXT panels only: everything up to c0 + 15 is finished)
        if (mine && rel >= 16) {
          const double *hq = a.Hq + (size_t)((int)xcc * G + grp) * 16 * a.C1P;
          if (BLK) {
            if (!leader) {
#pragma unroll
              for (int q = 0; q < 4; ++q) b[q] = xk_ld_sc1(myS + (size_t)(4 * q + part) * a.C1P + cabs);
            } else if (k > 0) {                              // what the first level left of the rows that were away
#pragma unroll
              for (int q = 0; q < 4; ++q) b[4 + q] = xk_ld_sc1(hq + (size_t)(4 * q + part) * a.C1P + cabs);
            }
          } else if (!leader) {
            if (part == 0) {
#pragma unroll
              for (int r = 0; r < 16; ++r) b[r] = xk_ld_sc1(myS + (size_t)r * a.C1P + cabs);
            }
          } else if (k > 0 && part == 1) {                   // what the first level left of the rows that were away
#pragma unroll
            for (int r = 0; r < 16; ++r) b[r] = xk_ld_sc1(hq + (size_t)r * a.C1P + cabs);
          }
        }
      }
      if (stamp) a.dbg[8 * k + 5] = wall_clock64();
    }
  } else {
    // ---- role L: the last merge level of every panel, on a CU of its own
    const int lidx = (int)xcc;                               // 8 of them
    const bool stamp = a.dbg && lidx == 0 && tid == 0;
    for (int k = 0; k < npanels && ok; ++k) {
      const int trail = max(0, a.C1 - 16 * k - 16);
      const int lchalf = max(4, 4 * ((trail + 31) / 32));    // trailing columns per workgroup: all 8 share the range
      const int lsplit = max(1, (trail + lchalf - 1) / lchalf);
      if (lidx >= lsplit) continue;
      if (tid == 0) s_ok = xk_spin_ge(sync + (XK_PS_X1FLAG + k) * 16, 1u, ab, 5u) ? 1u : 0u;
      __syncthreads();
      if (!s_ok) { ok = false; break; }
      if (stamp) a.dbg[8 * k + 6] = wall_clock64();
      xk_resident_last(ap, k, lidx, lchalf, ubuf, sc);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) xk_count_in(sync + (XK_PS_L2CNT + k) * 16, sync + (XK_PS_L2FLAG + k) * 16, 1u, (unsigned)lsplit);
      if (stamp) a.dbg[8 * k + 7] = wall_clock64();
    }
  }
  if (!ok && tid == 0) a.status[1] = (int)__hip_atomic_load(ab, XK_RLX_AGENT);
}
