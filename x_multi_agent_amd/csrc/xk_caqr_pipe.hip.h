// xk_caqr_pipe.hip.h -- QR compression (VioUpdater::applyQRDecomposition, src/x/vio/vio_updater.cpp:487-512) in ONE launch,
// the row stack resident in registers, and the three levels of the CAQR tree on WORKGROUPS OF THEIR OWN that run as a pipeline.
//
// What xk_caqr_resident (round 2) could not get past: every tile workgroup ran its tile step, then a first-level merge item,
// then waited for its pivot strip -- tile step -> first level -> tile step is a dependency cycle, and with both halves on the
// same CU it costs their SUM (10 + 10 us of reflector steps + 4 hand-offs = 30 us per 16-column panel).  Here:
//
//   * XCD x = NT TILE workgroups (a fat tile of LPC x RPL rows each, in registers for the whole factorisation, the column <->
//     thread map absolute) + NM FIRST-LEVEL workgroups (one group per XCD: NT strips + the pending strip, 16 lanes per column,
//     16 panel + <= 32 trailing columns each) + 1 LAST-LEVEL workgroup (8 roots; the eight of them share the trailing columns
//     of the panel).  Two geometries: 23 + 8 + 1 with 4 x 32-row lanes (<= 192 columns), 19 + 12 + 1 with 2 x 40 (<= 384).
//   * The strips of a merge are upper triangular in the panel columns: reflector j of the level above is zero in rows > j of
//     every strip, so its steps need the rows of the level below only as far as that level has got.  A tile publishes its
//     pivot strip in FOUR PHASES of 4 rows; the first level runs its steps 0..3 BESIDE the tile's steps 4..7 and so on, and
//     hands the root to the last level phase by phase in the same way.  The cycle tile -> first level -> tile is now
//     16 tile steps + the first level's last phase + two hand-offs, and the first level's own chain runs beside the tile's.
//   * Register 0 of every merge -- the pending strip, zero when there is none -- is the pivot strip: the root is produced in a
//     slot no tile owns, every tile gets its own strip back (no leaders, no holes).  What the last level leaves of the roots
//     of XCDs 1..7 goes down as the pending strip of that XCD's first level in the NEXT panel: a strip released after panel
//     k has seen the eliminations of panels <= k and must enter panel k + 1's tree (keeping it longer at the top does not
//     move that deadline -- the first version tried).
//   * Hand-off counters are fire-and-forget (no returning atomics on a producer's chain), consumers poll the counter itself
//     and fetch every phase that is already complete; strips travel in blocks of 4 columns x 16 rows (xk_blk).
// Hand-offs inside an XCD: plain stores + s_waitcnt vmcnt(0) + sc1 loads; across XCDs: write-through stores into per-panel
// slabs that are never reused inside a launch (xk_xcd_sync.hip.h).  All spins are bounded and look at an abort word; a launch
// that gives up is redone by the multi-launch schedule (xk_api.hip).  DESIGN 3.2.2 has the anatomy and the numbers.
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "xk_linalg.hip.h"
#include "xk_xcd_sync.hip.h"

#define XK_PIPE_THREADS 768
// Geometry of a launch.  LPC lanes per column x RPL rows per lane = rows of a fat tile (768 / LPC columns per workgroup);
// per XCD: NT tile workgroups + NM first-level workgroups + 1 last-level workgroup = 32 = the CUs of an XCD; NCL = column
// sets per last-level thread (its 8 workgroups cover NCL x 8 x <= 32 trailing columns), NCM = column sets per first-level thread.
// NG = first-level GROUPS per XCD (round 5): the XCD's NT strips are merged NTG = ceil(NT / NG) at a time by NM / NG workgroups each,
// every group with a pending strip and a root of its own -- the last level then merges 8 NG roots.  A first-level lane holds
// NTG + 1 rows instead of NT + 1: its steps and its strip loads are that much shorter, and the first level's chain is what a panel
// waits for.  It takes NG times the first-level workgroups for the same columns, i.e. fewer tile workgroups:
// the geometry for stacks whose ACCEPTED rows fit them.
#ifndef XK_PIPE_NPH
#define XK_PIPE_NPH 4               // hand-off phases per panel: a level publishes 16 / NPH rows of its strip at a time
#endif
template <int LPC_, int RPL_, int NT_, int NM_, int NCL_, int NCM_ = 1, int RPLS_ = RPL_, int RPLT_ = RPLS_, int NG_ = 1, int NPH_ = XK_PIPE_NPH>
struct XkPipeGeom {
  static constexpr int LPC = LPC_, RPL = RPL_, NT = NT_, NM = NM_, NCL = NCL_, NCM = NCM_, NG = NG_;
  static constexpr int NPH = NPH_;                         // hand-off phases per panel of this geometry (1, 2 or 4: the sync words hold 4)
  static_assert(NPH_ == 1 || NPH_ == 2 || NPH_ == 4, "XP_TQ_CNT / XP_X1_CNT hold four phases");
  static constexpr int NTG = (NT_ + NG_ - 1) / NG_, NMG = NM_ / NG_;   // strips / workgroups of a first-level group
  static_assert(NM_ % NG_ == 0 && NG_ >= 1 && NG_ <= 2, "first-level groups share the first-level workgroups evenly");
  static constexpr int RPLS = RPLS_, RPLT = RPLT_;         // lighter tile-step instantiations: fewer rows per lane, taken when the ACCEPTED rows fit
  static_assert(RPLS_ % 4 == 0 && RPLS_ >= 16 && RPLS_ <= RPL_ && RPLT_ % 4 == 0 && RPLT_ >= 16 && RPLT_ <= RPLS_, "the lighter tile steps");
  static constexpr int RM = (NTG + 2) & ~1;                // registers of a first-level lane: pending strip + NTG strips, even
  static constexpr int RL = 8 * NG_;                       // registers of a last-level lane: the roots
  static constexpr int COLS = XK_PIPE_THREADS / LPC_;      // widest system (C1P) a tile workgroup holds
  static constexpr int ROWS = 8 * NT_ * LPC_ * RPL_;       // most stacked rows
  static_assert(NT_ + NM_ + 1 == 32, "one workgroup per CU, 32 CUs per XCD");
  static_assert(RPL_ % 4 == 0 && RPL_ >= 16, "the pivot strip is the first 16 rows of the part-0 lane");
};
#ifndef XK_PIPE_NARROW
#define XK_PIPE_NARROW 4, 32, 23, 8, 1, 1, 28, 24
#endif
using XkPipeNarrow = XkPipeGeom<XK_PIPE_NARROW>;           // C1 <= 192 (MSCKF-only windows up to 31 poses): 184 tiles of 128 rows
#ifndef XK_PIPE_NARROW2
#define XK_PIPE_NARROW2 4, 32, 19, 12, 1, 1, 32, 28, 2, 4
#endif
using XkPipeNarrow2 = XkPipeGeom<XK_PIPE_NARROW2>;         // the same columns, 152 tiles of 128 rows, TWO first-level groups per XCD (10 + 9 strips on
                                                           // 6 workgroups each): taken when the rows that pass the gates are expected to fit 19 456
#ifndef XK_PIPE_WIDE
#define XK_PIPE_WIDE 2, 40, 19, 12, 2, 1, 32, 28, 1, 4
#endif
using XkPipeWide = XkPipeGeom<XK_PIPE_WIDE>;               // C1 <= 384 (SLAM features, BASELINE config 2): 152 tiles of 80 rows
// TAIL geometry (round 6): the LAST <= 96 columns of a system whose stack does not fit the register files at its full width
// (BASELINE config 3: 62 565 rows x 301 columns).  The multi-launch schedule sweeps the stack in HBM once per panel whatever the
// panel's width; once the columns still to be factored are few enough that ALL rows fit the registers at 8 lanes per column
// (184 fat tiles of 8 x 44 = 352 rows: 64 768 rows x 96 columns), one launch of this kernel finishes the factorisation: the rows are
// read once more and never written back (the launch only reads the stack, so a launch that gives up is redone by the schedule it
// took over from, on the same rows).  No Kalman role: these systems have n > 206.
// The same at 4 lanes per column: <= 192 columns, 32 384 rows a launch -- for stacks of up to twice that, TWO launches in a row, the
// second taking the first one's R (its <= 192 rows, XkCaqrPipeArgs::nextra) on top of the other half of the stack: a flat TSQR tree at
// launch granularity.  Two passes of ~26 us a panel beat the multi-launch schedule's 80..130 us a panel.
#ifndef XK_PIPE_TAIL
#define XK_PIPE_TAIL 8, 44, 23, 8, 1, 1, 44, 44, 1, 4
#endif
using XkPipeTail = XkPipeGeom<XK_PIPE_TAIL>;
#ifndef XK_PIPE_TAIL4
#define XK_PIPE_TAIL4 4, 44, 23, 8, 1, 1, 44, 44, 1, 4
#endif
using XkPipeTail4 = XkPipeGeom<XK_PIPE_TAIL4>;
#define XK_PIPE_RLS 16              // strips per panel of the cross-XCD slabs X1 / X1P / X2, whatever the geometry uses of them
#define XK_PIPE_NT_MAX 23
#define XK_PIPE_ROWS_MAX 24320      // (of the geometries that take their rows from 64-row slots)
#define XK_PIPE_SLOTS_MAX 1536      // 64-row slots (tracks + packed SLAM rows) a launch can compact
// Phase boundaries: phase q = reflector steps / strip rows [xk_pbn<NPH>(q), xk_pbn<NPH>(q + 1)); equal phases unless NPH = 4 and XK_PIPE_PB says
// otherwise.  (Measured, round 4: a short first phase -- the level above starts when the level below has published its first
// phase -- does NOT pay: 0,2,6,11,16: QR 0.348 ms, 0,3,7,11,16: 0.342, 0,1,4,9,16: 0.347 against 0.339 for 0,4,8,12,16.)
#ifndef XK_PIPE_PB
#define XK_PIPE_PB 0, 4, 8, 12, 16
#endif
#ifndef XK_PIPE_PB2
#define XK_PIPE_PB2 0, 8, 16        // ... of the two-phase geometries
#endif
template <int NPH>
__host__ __device__ constexpr int xk_pbn(int q) {
  if (NPH == 4) {
    constexpr int t[] = {XK_PIPE_PB};
    return q <= 0 ? 0 : (q >= 4 ? 16 : t[q]);
  }
  if (NPH == 2) {
    constexpr int t[] = {XK_PIPE_PB2};
    return q <= 0 ? 0 : (q >= 2 ? 16 : t[q]);
  }
  return q * (16 / NPH);
}
#ifndef XK_PIPE_ARRD
#define XK_PIPE_ARRD 1              // a phase's rows are counted in at the barrier ARRD steps after their stores were issued
#endif
#ifndef XK_PIPE_PF
#define XK_PIPE_PF 0                // 1: first level: a phase's rows of the tiles' strips come in as three 16-bytes-per-lane loads per
#endif                              // wave through LDS instead of 23 eight-byte loads per lane (measured: the loads of a phase land
                                    // in 0.76 instead of 1.44 us, the QR stage does not get shorter -- DESIGN 6.0)
#ifndef XK_PIPE_LOCALLD
#define XK_PIPE_LOCALLD 0           // 1: XCD-local strip loads as workgroup-scope loads behind an L1 invalidate (buffer_inv sc1):
                                    // correct, and 0.32 -> 0.59 ms -- the invalidate costs far more than the fabric round trips it saves
#endif
#if XK_PIPE_LOCALLD
#define XK_LD_LOC(p) xk_ld_grp(p)
#ifdef XK_PIPE_NOINV
#define XK_INV_LOC()
#else
#define XK_INV_LOC() xk_inv_l1()
#endif
#else
#define XK_LD_LOC(p) xk_ld_sc1(p)
#define XK_INV_LOC()
#endif
#ifndef XK_PIPE_TRI
#define XK_PIPE_TRI 0               // 1: merge steps skip the reflector in the rows where the strips' triangular structure makes it zero
                                    // (measured: the divergent halves cost more than the LDS traffic they save, QR 0.319 -> 0.332 ms)
#endif
#ifndef XK_PIPE_NLW
#define XK_PIPE_NLW 7               // last-level workgroups that share a panel's trailing columns (of the 8, one per XCD); the eighth
#endif                              // (XCD 7's) is free for the Kalman role (xk_pipe_kalman)
#ifndef XK_PIPE_CHUNK
#define XK_PIPE_CHUNK 0             // 1: tile steps fetch the reflector in two halves (no spills; measured 2 % slower than the spills)
#endif

enum {
  XP_CENSUS = 0,                    // [8] workgroups per XCD
  XP_ABORT = 9,
  // per XCD, monotonic arrival counters (a panel is complete at NT (k + 1) resp. NM (k + 1))
  XP_TQ_CNT = 16,                   // [4][8] tiles: rows of phase q of the pivot strips are out
  XP_MB_CNT = 48,                   // [8] first level: the strips are back
  // per panel (XCDs run up to a panel apart)
  XP_X1_CNT = 64,                   // [4][MAXP] first-level items whose root rows of phase q are out
  XP_P_CNT = 64 + 4 * XK_CAQR_MAXP,   // [MAXP] last-level workgroups whose pending strips are out
  XP_R_CNT = 64 + 5 * XK_CAQR_MAXP,   // [MAXP] last-level workgroups whose rows of R are out (Kalman role only)
  XP_WORDS = 64 + 6 * XK_CAQR_MAXP
};

struct XkCaqrPipeArgs {
  const double *A;        // tiles [ntiles][64][C1P] row-major as the per-feature kernels wrote them (read once)
  const double *Hc;       // factor records of slots [0, nhc) instead of their tiles (xk_feature.hip.h: XkFeatArgs::Hc), hs doubles each;
  int hs, nhc;            // nhc = 0: every slot is a tile in A
  const int *tile_rows;   // valid rows per 64-row slot (0 = rejected track), [nslots]: the stack is COMPACTED on the device --
  int nslots;             // fat tile j takes rows [j TR, (j + 1) TR) of the rows that passed the gates, TR = ceil(rows / tiles)
  int slot_rows;          // rows of a slot of A (64; 128 next to the multi-launch schedule's tall slots: the tail geometry)
  int lead_stride;        // > 0 (tail geometry): slots t % lead_stride == 0 led a merge group of the multi-launch schedule -- their first 32
                          // rows hold merged rows whatever their track's own row count was (xk_caqr_merge_body: pivot strip + hole)
  int res_col;            // > 0: the system is columns [0, C1 - 1) of A plus column res_col as its last one (the residual) -- a stack whose rows
                          // are zero in the columns in between (MSCKF rows next to SLAM features' columns, vio_updater.cpp:406-422: xk_api.hip,
                          // split compression); rows of R carry their last entry at column res_col as well
  int nextra;             // tail geometry, second launch of two: rows [extra_row0, extra_row0 + nextra) of A (row stride C1P) -- the R the first
  long extra_row0;        // launch left -- join the stack behind the slots' rows
  int C1P, C1;
  double *Rout;           // [C1P][C1P] row-major
  double *S;              // [8 NT][16 x C1P] pivot strips, block layout (xk_blk), XCD-local
  double *PB;             // [8 NT][16 x 16]  their panel blocks
  double *X1, *X1P;       // [panels][8][16 x C1P] / [panels][8][16 x 16]: the root of XCD x (write-through)
  double *X2;             // [panels][8][16 x C1P]: what the last level sends down to XCD s (write-through)
  double *Xnext;          // the OTHER set of X1 | X2 | X1P slabs (contiguous, xnext_doubles): re-armed with the NOT-YET pattern for the
  long xnext_doubles;     // next launch by the workgroups that idle at the start (data-polled hand-offs, xk_xcd_sync.hip.h)
  unsigned *sync, *sync_next;
  int *status;
  long long *dbg;
  int test_stall;         // test hook: one tile workgroup leaves at once -- everybody else runs into the bound of their spins
  int acc_tag;            // 1 .. 32767: written above the accepted-rows count in status word 2, so that the host can tell a late word of the launch before
  // Kalman role (xk_pipe_kalman; narrow geometry): Updater::applyUpdate (updater.cpp:117-141, cov_update)
  // inside this launch.  kal = 0: the compressed [T_H | z] is all the launch leaves behind.
  int kal, kn;            // on / off, n = error states
  const double *Pin;      // prior, n x n column-major
  double *Pout;           // posterior
  double sigma2;          // sigma_img^2 (vio_updater.cpp:508-509)
  double *corr;           // [n] correction (device or pinned host memory)
  const double *ct;       // [n] correction_total of the IEKF passes (updater.cpp:128: K (res + H ct) - ct), nullptr = 0: d starts at -ct
  unsigned long long *done_flag;   // optional completion marker (pinned host memory) ...
  unsigned long long done_seq;     // ... and its value
};
typedef const XkCaqrPipeArgs __attribute__((address_space(4))) *XkPipeArgsPtr;
__device__ __forceinline__ XkCaqrPipeArgs xk_pipe_args(XkPipeArgsPtr ap) {
  XkCaqrPipeArgs a;
  __builtin_memcpy(&a, (const void *)ap, sizeof(a));
  return a;
}

// Hand-off counters: the producer's thread 0 adds one WITHOUT asking for the old value (a returning atomic is a round trip to
// the L2 that its wave -- and, at the next barrier, its whole workgroup -- sits out: ~0.7 us per hand-off in the first
// version, which kept an arrival counter + a flag the last arriver raised); the consumers poll the counter itself.  They are
// few (8 first-level workgroups per tile counter, 23 tiles per first-level counter, one lane each, s_sleep between polls).
__device__ __forceinline__ void xk_pipe_arrive(unsigned *cnt) {
  (void)__hip_atomic_fetch_add(cnt, 1u, XK_ARRIVE_ORDER, __HIP_MEMORY_SCOPE_AGENT);
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
// column of A / of R that holds column c of the system (XkCaqrPipeArgs::res_col)
__device__ __forceinline__ int xk_pipe_srccol(const XkCaqrPipeArgs &a, int c) { return (a.res_col > 0 && c == a.C1 - 1) ? a.res_col : c; }
// How the last level cuts a panel's `trail` trailing columns: chunks of lchalf columns (whole quarter-waves, <= 32), one per
// workgroup while XK_PIPE_NLW of them cover the range, else (NCL = 2, the first panels of a wide system) two chunks per workgroup --
// x and x + XK_PIPE_NLW; lsplit = chunks in use.  The first level needs the same numbers for the panel before its own (how many
// last-level workgroups send pending strips down).
__device__ __forceinline__ void xk_pipe_lastcut(int trail, int NCL, int &ncl, int &lchalf, int &lsplit) {
  ncl = (NCL > 1 && trail > 32 * XK_PIPE_NLW) ? NCL : 1;
  lchalf = max(4, 4 * ((trail + 4 * XK_PIPE_NLW * ncl - 1) / (4 * XK_PIPE_NLW * ncl)));
  lsplit = max(1, (trail + lchalf - 1) / lchalf);
}
// thread 0 polls, everybody learns the verdict
__device__ __forceinline__ bool xk_pipe_wait(unsigned *word, unsigned target, unsigned *ab, unsigned reason, unsigned *s_ok) {
  if (threadIdx.x == 0) *s_ok = xk_spin_ge(word, target, ab, reason) ? 1u : 0u;
  __syncthreads();
  const bool ok = *s_ok != 0u;
  XK_ACQUIRE_FENCE();
  __syncthreads();                  // (s_ok is rewritten by the next wait)
  return ok;
}

// Waits for phase q of a producer and looks (without waiting) how many of the following phases are complete as well: the
// consumer then fetches all of them at once.  A level that runs behind its producer finds the next phases ready and saves
// their waits and load latencies; rows fetched early are harmless -- the strips are upper triangular in the panel columns,
// so the reflectors of the earlier steps are zero in those rows.  cnt0 = phase 0's counter, `stride` words between phases.
// Returns the number of complete phases (> q), 0 if the wait gave up.
__device__ __forceinline__ int xk_pipe_wait_phases(unsigned *cnt0, int stride, int q, int nph, unsigned target, unsigned *ab, unsigned reason,
                                                   unsigned *s_ok) {
  if (threadIdx.x == 0) {
    unsigned av = 0;
    if (xk_spin_ge(cnt0 + (size_t)q * stride, target, ab, reason)) {
      av = (unsigned)q + 1u;
      while ((int)av < nph && __hip_atomic_load(cnt0 + (size_t)av * stride, XK_RLX_AGENT) >= target) ++av;
    }
    *s_ok = av;
  }
  __syncthreads();
  const int av = (int)*s_ok;
  XK_ACQUIRE_FENCE();
  __syncthreads();
  return av;
}

// xk_caqr_apply (tile layout, LPC lanes per column) with the reflector fetched in two halves: past 32 rows per lane the lane's
// rows + the whole reflector + the step's temporaries do not fit 168 registers.  Same sums in the same order; the first half
// is read from LDS a second time.  (At 32 rows per lane the plain version spills a few loop invariants and is still 2 %
// faster: XK_PIPE_CHUNK = 0.)
template <int KK, int LPC, int RPL>
__device__ __forceinline__ void xk_pipe_tapply(double (&b)[RPL], int rel, bool live, int part, const double *ubuf, const double *sc) {
  if constexpr (XK_PIPE_CHUNK || RPL > 32) {
    constexpr int RPLP = RPL + 2, H = RPL / 4;
    constexpr int pb = KK & 1;
    const xk_d2 *useg = reinterpret_cast<const xk_d2 *>(ubuf + (pb * LPC + part) * RPLP);
    const double mtt = sc[pb * 4];
    if (rel > KK && live && mtt != 0.0) {
      double d0 = 0.0, d1 = 0.0, d2 = 0.0, d3 = 0.0;
      {
        xk_d2 u[H];
#pragma unroll
        for (int r = 0; r < H; ++r) u[r] = useg[r];
#pragma unroll
        for (int r = 0; r < H; ++r) {
          if (r & 1) { d2 = fma(u[r][0], b[2 * r], d2); d3 = fma(u[r][1], b[2 * r + 1], d3); }
          else { d0 = fma(u[r][0], b[2 * r], d0); d1 = fma(u[r][1], b[2 * r + 1], d1); }
        }
      }
      xk_d2 v[H];
#pragma unroll
      for (int r = 0; r < H; ++r) v[r] = useg[H + r];
#pragma unroll
      for (int r = 0; r < H; ++r) {
        if ((H + r) & 1) { d2 = fma(v[r][0], b[2 * (H + r)], d2); d3 = fma(v[r][1], b[2 * (H + r) + 1], d3); }
        else { d0 = fma(v[r][0], b[2 * (H + r)], d0); d1 = fma(v[r][1], b[2 * (H + r) + 1], d1); }
      }
      const double w = mtt * xk_group_sum<LPC>((d0 + d1) + (d2 + d3));
#pragma unroll
      for (int r = 0; r < H; ++r) {
        b[2 * (H + r)] = fma(w, v[r][0], b[2 * (H + r)]);
        b[2 * (H + r) + 1] = fma(w, v[r][1], b[2 * (H + r) + 1]);
      }
      const xk_d2 *useg2 = useg + xk_launder(0);             // (a second look at the same LDS words, not the registers of the first)
#pragma unroll
      for (int r = 0; r < H; ++r) {
        const xk_d2 u = useg2[r];
        b[2 * r] = fma(w, u[0], b[2 * r]);
        b[2 * r + 1] = fma(w, u[1], b[2 * r + 1]);
      }
    }
  } else {
#if defined(XK_PIPE_PROBE_THALF)
    // TIMING PROBE ONLY (wrong results): the tile step's apply with half the reflector and half the multiply-adds
    constexpr int RPLP = RPL + 2;
    constexpr int pb = KK & 1;
    const xk_d2 *useg = reinterpret_cast<const xk_d2 *>(ubuf + (pb * LPC + part) * RPLP);
    const double mtt = sc[pb * 4];
    if (rel > KK && live && mtt != 0.0) {
      xk_d2 u[RPL / 4];
#pragma unroll
      for (int r = 0; r < RPL / 4; ++r) u[r] = useg[r];
      double d0 = 0.0, d1 = 0.0, d2 = 0.0, d3 = 0.0;
#pragma unroll
      for (int r = 0; r < RPL / 4; ++r) {
        if (r & 1) { d2 = fma(u[r][0], b[2 * r], d2); d3 = fma(u[r][1], b[2 * r + 1], d3); }
        else { d0 = fma(u[r][0], b[2 * r], d0); d1 = fma(u[r][1], b[2 * r + 1], d1); }
      }
      const double w = mtt * xk_group_sum<LPC>((d0 + d1) + (d2 + d3));
#pragma unroll
      for (int r = 0; r < RPL / 4; ++r) {
        b[2 * r] = fma(w, u[r][0], b[2 * r]);
        b[2 * r + 1] = fma(w, u[r][1], b[2 * r + 1]);
      }
    }
#else
    xk_caqr_apply<KK, LPC, RPL>(b, rel, live, part, ubuf, sc);
#endif
  }
}

// Steps [K0, K1) of a panel with the one-reflector look-ahead of xk_caqr_steps_la; on entry every reflector < K0 has been
// applied, on return every reflector < min(K1, nsteps).
// KH: the iteration whose barrier carries a hand-off -- every wave drains its stores before it, thread 0 runs `hook` after it.
// LPC > 0: the tile layout with LPC lanes per column; LPC = 0: the merge layout (16 lanes per column).  b2 (merge layout only):
// a second set of trailing columns of the same lanes -- it only ever takes reflectors, it never owns one.
template <int LPC, int K, int RPL>
__device__ __forceinline__ void xk_pipe_form(double (&b)[RPL], int rel, int part, double *ubuf, double *sc) {
  if constexpr (LPC > 0) xk_caqr_form<K, LPC, RPL>(b, rel, part, ubuf, sc);
  else xk_caqr_mform<K, RPL>(b, rel, part, ubuf, sc);
}
// Merge layout, reflector KK applied to one column.  The strips of a merge are upper triangular in the panel columns, so column
// KK -- the reflector -- is ZERO in rows > KK of every strip except the pivot strip's slot (register 0: the pending strip of the
// first level is dense).  Lanes part > KK therefore fetch and use register 0's entry only: on average half the lanes skip the
// reflector.  Correct, and slower (the two lane populations run one after the other): off by default.
template <int KK, int RPL>
__device__ __forceinline__ void xk_pipe_mapply(double (&b)[RPL], int rel, bool live, int part, const double *ubuf, const double *sc) {
#if XK_PIPE_TRI
  constexpr int RPLP = RPL + 2;
  constexpr int pb = KK & 1;
  const xk_d2 *useg = reinterpret_cast<const xk_d2 *>(ubuf + (pb * 16 + part) * RPLP);
  const double mtt = sc[pb * 4];
  if (rel > KK && live && mtt != 0.0) {
    const bool full = part <= KK;
    xk_d2 u[RPL / 2];
    double dsum;
    if (full) {
#pragma unroll
      for (int r = 0; r < RPL / 2; ++r) u[r] = useg[r];
      double d0 = 0.0, d1 = 0.0, d2 = 0.0, d3 = 0.0;
#pragma unroll
      for (int r = 0; r < RPL / 2; ++r) {
        if (r & 1) { d2 = fma(u[r][0], b[2 * r], d2); d3 = fma(u[r][1], b[2 * r + 1], d3); }
        else { d0 = fma(u[r][0], b[2 * r], d0); d1 = fma(u[r][1], b[2 * r + 1], d1); }
      }
      dsum = (d0 + d1) + (d2 + d3);
    } else {
      u[0][0] = reinterpret_cast<const double *>(useg)[0];
      dsum = u[0][0] * b[0];
    }
    const double w = mtt * xk_group_sum<16>(dsum);
    if (full) {
#pragma unroll
      for (int r = 0; r < RPL / 2; ++r) {
        b[2 * r] = fma(w, u[r][0], b[2 * r]);
        b[2 * r + 1] = fma(w, u[r][1], b[2 * r + 1]);
      }
    } else {
      b[0] = fma(w, u[0][0], b[0]);
    }
  }
#elif defined(XK_PIPE_PROBE_HALF)
  // TIMING PROBES ONLY (wrong results): XK_PIPE_PROBE_HALF = 1 the same arithmetic with half the reflector fetched from LDS,
  // = 2 half the reflector fetched AND half the multiply-adds -- which of the two a merge step is bound by
  constexpr int RPLP = RPL + 2;
  constexpr int pb = KK & 1;
  constexpr int NR = (XK_PIPE_PROBE_HALF == 2) ? RPL / 4 : RPL / 2;
  const xk_d2 *useg = reinterpret_cast<const xk_d2 *>(ubuf + (pb * 16 + part) * RPLP);
  const double mtt = sc[pb * 4];
  if (rel > KK && live && mtt != 0.0) {
    xk_d2 u[RPL / 2];
#pragma unroll
    for (int r = 0; r < RPL / 4; ++r) u[r] = useg[r];
#pragma unroll
    for (int r = RPL / 4; r < RPL / 2; ++r) u[r] = u[r - RPL / 4];
    double d0 = 0.0, d1 = 0.0, d2 = 0.0, d3 = 0.0;
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      if (r & 1) { d2 = fma(u[r][0], b[2 * r], d2); d3 = fma(u[r][1], b[2 * r + 1], d3); }
      else { d0 = fma(u[r][0], b[2 * r], d0); d1 = fma(u[r][1], b[2 * r + 1], d1); }
    }
    const double w = mtt * xk_group_sum<16>((d0 + d1) + (d2 + d3));
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      b[2 * r] = fma(w, u[r][0], b[2 * r]);
      b[2 * r + 1] = fma(w, u[r][1], b[2 * r + 1]);
    }
  }
#else
  xk_caqr_apply<KK, 16, RPL>(b, rel, live, part, ubuf, sc);
#endif
}
template <int LPC, int K, int RPL>
__device__ __forceinline__ void xk_pipe_apply(double (&b)[RPL], int rel, bool live, int part, const double *ubuf, const double *sc) {
  if constexpr (LPC > 0) xk_pipe_tapply<K, LPC, RPL>(b, rel, live, part, ubuf, sc);
  else xk_pipe_mapply<K, RPL>(b, rel, live, part, ubuf, sc);
}
// One reflector applied to TWO column sets of the same lanes (merge layout): the reflector is fetched from LDS once.  Used by
// the last level of wide systems (8 rows per lane); in the first level the second set does not fit the registers.
template <int KK, int RPL>
__device__ __forceinline__ void xk_pipe_mapply_pair(double (&b)[RPL], double (&b2)[RPL], int rel, bool live, bool live2, int part, const double *ubuf,
                                                    const double *sc) {
  constexpr int RPLP = RPL + 2;
  constexpr int pb = KK & 1;
  const xk_d2 *useg = reinterpret_cast<const xk_d2 *>(ubuf + (pb * 16 + part) * RPLP);
  const double mtt = sc[pb * 4];
  if (rel > KK && (live || live2) && mtt != 0.0) {
    xk_d2 u[RPL / 2];
#pragma unroll
    for (int r = 0; r < RPL / 2; ++r) u[r] = useg[r];
    double d0 = 0.0, d1 = 0.0, d2 = 0.0, d3 = 0.0, e0 = 0.0, e1 = 0.0, e2 = 0.0, e3 = 0.0;
#pragma unroll
    for (int r = 0; r < RPL / 2; ++r) {
      if (r & 1) {
        d2 = fma(u[r][0], b[2 * r], d2); d3 = fma(u[r][1], b[2 * r + 1], d3);
        e2 = fma(u[r][0], b2[2 * r], e2); e3 = fma(u[r][1], b2[2 * r + 1], e3);
      } else {
        d0 = fma(u[r][0], b[2 * r], d0); d1 = fma(u[r][1], b[2 * r + 1], d1);
        e0 = fma(u[r][0], b2[2 * r], e0); e1 = fma(u[r][1], b2[2 * r + 1], e1);
      }
    }
    const double w = mtt * xk_group_sum<16>((d0 + d1) + (d2 + d3));
    const double w2 = mtt * xk_group_sum<16>((e0 + e1) + (e2 + e3));
#pragma unroll
    for (int r = 0; r < RPL / 2; ++r) {
      b[2 * r] = fma(w, u[r][0], b[2 * r]);
      b[2 * r + 1] = fma(w, u[r][1], b[2 * r + 1]);
      b2[2 * r] = fma(w2, u[r][0], b2[2 * r]);
      b2[2 * r + 1] = fma(w2, u[r][1], b2[2 * r + 1]);
    }
  }
}
template <int LPC, int K, int RPL>
__device__ __forceinline__ void xk_pipe_apply2(double (&b)[RPL], double (*b2)[RPL], int rel, bool live, bool live2, int part, const double *ubuf,
                                               const double *sc) {
  if constexpr (LPC == 0) {
    // (a lane with a second set is a trailing lane: rel = 16 for both sets; panel lanes pass b2 with live2 = false)
    if (b2 && rel >= 16) { xk_pipe_mapply_pair<K, RPL>(b, *b2, rel, live, live2, part, ubuf, sc); return; }
  }
  xk_pipe_apply<LPC, K, RPL>(b, rel, live, part, ubuf, sc);
}
// `every` is called by all threads behind the barrier of every step (the first level's look-out for the next phase's rows)
struct XkNoStepHook { __device__ __forceinline__ void operator()(int) const {} };
template <int LPC, int K, int K1, int KH, int RPL, typename Hook, typename Every>
__device__ __forceinline__ void xk_pipe_range_it(double (&b)[RPL], double (*b2)[RPL], int rel, bool live, bool live2, int part, int nsteps,
                                                 double *ubuf, double *sc, bool hook_on, Hook hook, Every every) {
  if constexpr (K < K1) {
    if (K < nsteps) {
      xk_pipe_apply2<LPC, K - 1, RPL>(b, b2, rel, live, live2, part, ubuf, sc);
      xk_pipe_form<LPC, K, RPL>(b, rel, part, ubuf, sc);
      if (K == KH && hook_on) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (K == KH && hook_on && threadIdx.x == 0) hook();
      every(K);
    } else if (K == nsteps) {
      xk_pipe_apply2<LPC, K - 1, RPL>(b, b2, rel, live, live2, part, ubuf, sc);
    }
    xk_pipe_range_it<LPC, K + 1, K1, KH, RPL>(b, b2, rel, live, live2, part, nsteps, ubuf, sc, hook_on, hook, every);
  }
}
template <int LPC, int K0, int K1, int KH, int RPL, typename Hook, typename Every>
__device__ __forceinline__ void xk_pipe_range(double (&b)[RPL], double (*b2)[RPL], int rel, bool live, bool live2, int part, int nsteps,
                                              double *ubuf, double *sc, bool hook_on, Hook hook, Every every) {
  if (K0 < nsteps) xk_pipe_form<LPC, K0, RPL>(b, rel, part, ubuf, sc);
  if (KH == K0 && hook_on) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (KH == K0 && hook_on && threadIdx.x == 0) hook();
  every(K0);
  xk_pipe_range_it<LPC, K0 + 1, K1, KH, RPL>(b, b2, rel, live, live2, part, nsteps, ubuf, sc, hook_on, hook, every);
  if (nsteps >= K1) xk_pipe_apply2<LPC, K1 - 1, RPL>(b, b2, rel, live, live2, part, ubuf, sc);
}
template <int LPC, int K0, int K1, int KH, int RPL, typename Hook>
__device__ __forceinline__ void xk_pipe_range(double (&b)[RPL], double (*b2)[RPL], int rel, bool live, bool live2, int part, int nsteps,
                                              double *ubuf, double *sc, bool hook_on, Hook hook) {
  xk_pipe_range<LPC, K0, K1, KH, RPL>(b, b2, rel, live, live2, part, nsteps, ubuf, sc, hook_on, hook, XkNoStepHook());
}

// ---- role T: one fat tile in registers for the whole factorisation
// Which rows are mine.  The per-feature kernels leave tile_rows[slot] = rows of the slot that hold data (0: a track the gate
// rejected); the reference appends inliers only (msckf_update.cpp:463-479), and so does this: every tile workgroup forms the
// prefix sums of tile_rows (the same few hundred words, in LDS -- nobody waits for anybody), learns the number of rows that
// passed, TR = ceil(rows / tiles), and turns its rows [j TR, (j + 1) TR) into physical rows by a binary search.  Rejected tracks
// cost no registers; the host needs no row map (it does not know the verdicts when it queues the launch).
// Returns TR (0: more rows than the tiles hold -- the launch gives up, the multi-launch schedule serves the update).
template <class G>
__device__ __forceinline__ int xk_pipe_rowplan(const XkCaqrPipeArgs &a, int j, int *pre, int *myrows, int *rows_accepted) {
  constexpr int NTP = 8 * G::NT, CAP = G::LPC * G::RPL;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  __shared__ int wsum[XK_PIPE_THREADS / 64];
  int carry = 0;
  for (int base = 0; base < a.nslots; base += XK_PIPE_THREADS) {
    const int t = base + tid;
    int v = t < a.nslots ? a.tile_rows[t] : 0;
    if (a.lead_stride > 0 && t < a.nslots && t % a.lead_stride == 0) v = max(v, 32);
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {                    // inclusive scan inside the wave
      const int u = __shfl_up(v, o, 64);
      if (lane >= o) v += u;
    }
    if (lane == 63) wsum[wave] = v;
    __syncthreads();
    int off = carry;
    for (int w = 0; w < wave; ++w) off += wsum[w];
    if (t < a.nslots) pre[t + 1] = v + off;
    int tot = 0;
    for (int w = 0; w < XK_PIPE_THREADS / 64; ++w) tot += wsum[w];
    carry += tot;
    __syncthreads();
  }
  if (tid == 0) pre[0] = 0;
  const int Rs = carry, R = Rs + a.nextra, TR = (R + NTP - 1) / NTP;
  *rows_accepted = R;
  __syncthreads();
  if (TR > CAP) return 0;
  if (tid < CAP) {
    const int g = j * TR + tid;
    int phys = -1;
    if (tid < TR && g >= Rs && g < R) phys = (int)a.extra_row0 + (g - Rs);
    else if (tid < TR && g < Rs) {
      int lo = 0, hi = a.nslots;                           // largest slot with pre[slot] <= g
      while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (pre[mid] <= g) lo = mid; else hi = mid;
      }
      phys = lo * a.slot_rows + (g - pre[lo]);
    }
    myrows[tid] = phys;
  }
  __syncthreads();
  return max(TR, 1);
}

template <class G, int RPL>
__device__ __noinline__ bool xk_pipe_tile(XkPipeArgsPtr ap, int xcc, int slot, long long t_entry, double *ubuf, double *sc, unsigned *s_ok,
                                          const int *myrows, double *hbuf, int hcap_doubles, int nvalid) {
  constexpr int NT = G::NT, LPC = G::LPC, NPH = G::NPH, GS = 16 / NPH, ARRD = XK_PIPE_ARRD;
  static_assert(ARRD < xk_pbn<NPH>(1) && ARRD < 16 - xk_pbn<NPH>(NPH - 1), "a phase is counted in before the next one is published");
  const XkCaqrPipeArgs a = xk_pipe_args(ap);
  unsigned *sync = a.sync, *ab = sync + XP_ABORT * 16;
  const int tid = threadIdx.x;
  const int j = xcc * NT + slot;                           // my fat tile: valid rows [j TR, (j + 1) TR)
  const int cabs = tid / LPC, part_ = tid % LPC;           // ABSOLUTE column of this thread, all panels
  const bool mine = cabs < a.C1;
  const int npanels = (a.C1 + 15) / 16;
  const bool stamp = a.dbg && xcc == 0 && slot == 1 && tid == 0;
  double b[RPL];
  if (a.nhc > 0) {
    // My rows from the FACTOR RECORDS of their tracks (H0 itself never exists in HBM).  The records of the slots
    // my rows come from -- a tile's rows are consecutive accepted rows, so consecutive slots, rejected ones in between -- are
    // staged in LDS (as many as fit at a time: one chunk unless the tracks are very short).  The entries are formed in the layout
    // that makes the track of a row WAVE-UNIFORM -- thread (fp, fc) = rows of lane-part fp, column fc: the row part {v0, v1, v2,
    // r'} is one broadcast read, the column part {w0, w1, w2, x0, x1, r0} is re-read when the track changes, 3 multiply-adds
    // (xk_h0_entry) -- eight registers' worth of rows at a time into an LDS block, from where the factorisation's lanes (column
    // cabs, part part_) take theirs.  Rows of slots >= nhc (SLAM features, tracks that become features) are tiles in A.
    constexpr int BR = 8, CG = XK_PIPE_THREADS / LPC;
    const int hs = a.hs, C1P = a.C1P, ccl = min(cabs, C1P - 1);
    double *Tb = hbuf, *recs = hbuf + LPC * BR * C1P;
    const int hcap = (hcap_doubles - LPC * BR * C1P) / hs;
#pragma unroll
    for (int r = 0; r < RPL; ++r) b[r] = 0.0;
    if (stamp) a.dbg[1544] = wall_clock64();
    const int s_lo = myrows[0] >> 6, s_hi = nvalid > 0 ? myrows[nvalid - 1] >> 6 : -1;   // (an empty tile: nothing below runs)
    const int fp = __builtin_amdgcn_readfirstlane(tid / CG), fc = tid % CG;
    if (stamp) a.dbg[1545] = wall_clock64();
    for (int lo = max(s_lo, 0); lo <= s_hi && lo < a.nhc; lo += hcap) {
      const int hi = min(min(lo + hcap, s_hi + 1), a.nhc);
      {
        // (loads first, stores behind them: six 16-byte pieces per thread in flight -- one record is 700 of them)
        const xk_d2 *src = reinterpret_cast<const xk_d2 *>(a.Hc + (size_t)lo * hs);
        xk_d2 *dst = reinterpret_cast<xk_d2 *>(recs);
        const int tot = (hi - lo) * (hs / 2);
        for (int base = 0; base < tot; base += 6 * XK_PIPE_THREADS) {
          xk_d2 t[6];
#pragma unroll
          for (int i = 0; i < 6; ++i) t[i] = src[min(base + i * XK_PIPE_THREADS + tid, tot - 1)];
#pragma unroll
          for (int i = 0; i < 6; ++i) {
            const int e = base + i * XK_PIPE_THREADS + tid;
            if (e < tot) dst[e] = t[i];
          }
        }
      }
      __syncthreads();
      if (stamp) a.dbg[1546] = wall_clock64();
      int kcur = -1, r0 = -2;
      double nw0 = 0.0, nw1 = 0.0, nw2 = 0.0, wres = 0.0, x0 = 0.0, x1 = 0.0;
      // (every row of the tile comes from a record of this chunk -- the usual case: a row exists iff its index is < nvalid, no lookups)
      const bool whole = lo <= s_lo && s_hi < hi;
#pragma unroll
      for (int rb = 0; rb < (RPL + BR - 1) / BR; ++rb) {
        if (fc < C1P) {
#pragma unroll
          for (int j = 0; j < BR; ++j) {
            if (rb * BR + j < RPL) {
              const int ph = __builtin_amdgcn_readfirstlane(myrows[fp * RPL + rb * BR + j]), sl = ph >> 6;
              if (ph >= 0 && sl >= lo && sl < hi) {        // (wave-uniform)
                const double *rec = recs + (sl - lo) * hs;
                if (sl != kcur) {
                  const xk_d2 *wc = reinterpret_cast<const xk_d2 *>(rec + XK_HC_VR + XK_HC_WC * xk_pipe_srccol(a, fc));
                  const xk_d2 q0 = wc[0], q1 = wc[1], q2 = wc[2];
                  nw0 = -q0[0]; nw1 = -q0[1]; nw2 = -q1[0]; x0 = q1[1]; x1 = q2[0]; r0 = (int)q2[1];
                  wres = (r0 == -1) ? 1.0 : 0.0;             // the residual column: w = x = 0, the entry is r'
                  kcur = sl;
                }
                // xk_h0_entry without its selects: r0 = -1 / -2 never meet a row (rows start at 3), an untouched column is all
                // zeros, and + 0.0 comes last (so it is +0.0 whatever the signs of v0..v2)
                const int rr = (ph & 63) + 3, dd = rr - r0;
                const xk_d2 va = reinterpret_cast<const xk_d2 *>(rec + 4 * rr)[0], vb = reinterpret_cast<const xk_d2 *>(rec + 4 * rr)[1];
                const double add = (dd == 0) ? x0 : ((dd == 1) ? x1 : 0.0);
                Tb[(fp * BR + j) * C1P + fc] = fma(wres, vb[1], fma(nw2, vb[0], fma(nw1, va[1], nw0 * va[0]))) + add;
              }
            }
          }
        }
        __syncthreads();
        if (whole) {
#pragma unroll
          for (int j = 0; j < BR; ++j) {
            if (rb * BR + j < RPL) {
              const double v = Tb[(part_ * BR + j) * C1P + ccl];
              b[rb * BR + j] = (mine && part_ * RPL + rb * BR + j < nvalid) ? v : 0.0;
            }
          }
        } else {
#pragma unroll
          for (int j = 0; j < BR; ++j) {
            if (rb * BR + j < RPL) {
              const int ph = myrows[part_ * RPL + rb * BR + j], sl = ph >> 6;
              const double v = Tb[(part_ * BR + j) * C1P + ccl];
              if (ph >= 0 && sl >= lo && sl < hi) b[rb * BR + j] = mine ? v : 0.0;
            }
          }
        }
        __syncthreads();                                   // (the block is rewritten; after the last one: hbuf and ubuf have other users)
        if (stamp) a.dbg[1547 + rb] = wall_clock64();
      }
    }
    if (s_hi >= a.nhc) {                                   // (workgroup-uniform) rows that are tiles
      const double *Ac = a.A + xk_pipe_srccol(a, ccl);
#pragma unroll
      for (int h0 = 0; h0 < RPL; h0 += 16) {
        double x[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          if (h0 + r < RPL) x[r] = Ac[(size_t)max(myrows[part_ * RPL + h0 + r], 0) * C1P];
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          if (h0 + r < RPL) {
            const int ph = myrows[part_ * RPL + h0 + r];
            if (mine && ph >= 0 && (ph >> 6) >= a.nhc) b[h0 + r] = x[r];
          }
        }
      }
    }
  } else
  {   // The one pass over the stack: my rows, through the physical row numbers xk_pipe_rowplan left in LDS (-1: no row).  Branch-free
      // and in batches -- every index clamped into range, every load issued whether or not its value is used -- so that 16 loads are
      // in flight at a time: with a branch per row the compiler waited for each row's data before the next, 22 us of the launch.
    int pr[RPL];
#pragma unroll
    for (int r = 0; r < RPL; ++r) pr[r] = myrows[part_ * RPL + r];
    const double *Ac = a.A + xk_pipe_srccol(a, min(cabs, a.C1P - 1));   // (the wide geometry's column slots run to 383, past C1P = 256 / 320: clamped into the row)
#pragma unroll
    for (int h0 = 0; h0 < RPL; h0 += 16) {
      double x[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        if (h0 + r < RPL) x[r] = Ac[(size_t)max(pr[h0 + r], 0) * a.C1P];
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        if (h0 + r < RPL) b[h0 + r] = (mine && pr[h0 + r] >= 0) ? x[r] : 0.0;
      }
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
    const bool full = nsteps == 16;                        // (a short panel -- the last one -- hands everything over at its end)
    const int wrel = (tid >> 6) * (64 / LPC) - c0;         // first column of my wave, relative to the panel
    const bool hot = wrel <= 0 && wrel > -(64 / LPC);       // my wave holds panel columns (wave-uniform)
    const bool pub = mine && part == 0 && rel >= 0;
    const unsigned epoch = (unsigned)(k + 1);
    if (stamp) { a.dbg[16 * k + 0] = wall_clock64(); a.dbg[16 * k + 8] = clock64(); }
    if (a.dbg && slot == 0 && tid == 0) a.dbg[12288 + xcc * 32 + k] = wall_clock64();       // (skew_trace: every XCD's panel start)
    if (a.dbg && tid == 0 && k < 2) a.dbg[40960 + 256 * k + j] = wall_clock64();                 // (every tile's start of panels 0 and 1)
    if (a.dbg && tid == 0 && k == 0) a.dbg[40960 + 512 + j] = t_entry;
    if (hot) __builtin_amdgcn_s_setprio(3);
    // phase q: steps [q GS, (q + 1) GS), then rows [q GS, (q + 1) GS) of the pivot strip are final and go out; they are counted in
    // ARRD steps into the next phase (their stores drain behind those steps), the last phase after the panel
    auto phase = [&](auto qc) {
      constexpr int q = decltype(qc)::value;
      auto hook = [&]() { xk_pipe_arrive(sync + (XP_TQ_CNT + (q - 1) * 8 + xcc) * 16); if (a.dbg && k == 0 && q == 1) a.dbg[47104 + j] = wall_clock64(); };
      xk_pipe_range<LPC, xk_pbn<NPH>(q), xk_pbn<NPH>(q + 1), (q > 0 ? xk_pbn<NPH>(q) + ARRD : -1), RPL>(b, nullptr, rel, mine, false, part, nsteps, ubuf, sc, full, hook);
      if (pub) {
        if (rel < 16) {
          double *pb = xk_opaque(myPB + xk_blk(rel, 0));
#pragma unroll
          for (int r = xk_pbn<NPH>(q); r < xk_pbn<NPH>(q + 1); ++r) pb[r * 4] = (r > rel) ? 0.0 : b[r];
        } else {
          double *ps = xk_opaque(myS + xk_blk(cabs, 0));
#pragma unroll
          for (int r = xk_pbn<NPH>(q); r < xk_pbn<NPH>(q + 1); ++r) ps[r * 4] = b[r];
        }
      }
      if (stamp && q < 4) a.dbg[16 * k + 1 + q] = wall_clock64();
    };
    phase(std::integral_constant<int, 0>{});
    if constexpr (NPH >= 2) phase(std::integral_constant<int, 1>{});
    if constexpr (NPH >= 4) { phase(std::integral_constant<int, 2>{}); phase(std::integral_constant<int, 3>{}); }
    if constexpr (NPH >= 8) {
      phase(std::integral_constant<int, 4>{}); phase(std::integral_constant<int, 5>{});
      phase(std::integral_constant<int, 6>{}); phase(std::integral_constant<int, 7>{});
    }
    if (hot) __builtin_amdgcn_s_setprio(0);
    if (stamp) { a.dbg[16 * k + 9] = clock64(); a.dbg[16 * k + 10] = wall_clock64(); }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      for (int q = full ? NPH - 1 : 0; q < NPH; ++q)
        xk_pipe_arrive(sync + (XP_TQ_CNT + q * 8 + xcc) * 16);
    }
    if (stamp) a.dbg[16 * k + 5] = wall_clock64();
    if (k + 1 == npanels) break;
    // my strip comes back from the first level (trailing columns of the NEXT panels only: everything up to c0 + 15 is finished)
    if (!xk_pipe_wait(sync + (XP_MB_CNT + xcc) * 16, (unsigned)G::NM * epoch, ab, 2u, s_ok)) return false;
    if (stamp) a.dbg[16 * k + 6] = wall_clock64();
    XK_INV_LOC();
    if (mine && rel >= 16 && part == 0) {
      const double *ps = xk_opaque(myS + xk_blk(cabs, 0));
#pragma unroll
      for (int r = 0; r < 16; ++r) b[r] = XK_LD_LOC(ps + r * 4);
    }
    if (stamp) {
      double sink = 0;
      for (int r = 0; r < 16; ++r) sink += b[r];
      asm volatile("" ::"v"(sink));
      a.dbg[16 * k + 7] = wall_clock64();
    }
  }
  if (stamp) a.dbg[1538] = wall_clock64();
  return true;
}

// 16 bytes per lane from global memory straight into LDS (lane l lands at lds_byte + 16 l): device-scope load, no VGPR
// destination, so nothing in the step code waits for it.  Written in assembly on purpose: the compiler would put a vmcnt(0) in
// front of the next LDS read of ANY array; here the one wait is placed by hand where the rows are taken out.
__device__ __forceinline__ void xk_lds_dma16(const double *src, unsigned lds_byte) {
  asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off sc1" ::"v"(src), "s"(lds_byte) : "memory", "m0");
}

// ---- role M: the first merge level of this XCD's NT strips (+ the pending strip), 16 panel + <= 32 trailing columns per workgroup
// 16 lanes per column: lane p = row p of every strip, register 0 = the pending strip = the pivot strip, register 1 + t = tile t
// (register RM - 1 stays zero when NT is even)
template <class G>
__device__ __noinline__ bool xk_pipe_first(XkPipeArgsPtr ap, int xcc, int item, double *ubuf, double *sc, unsigned *s_ok, double *pfbuf) {
  constexpr int NT = G::NT, NM = G::NMG, RM = G::RM, NP = 16, NPH = G::NPH, GS = 16 / NPH, NCL = G::NCL, NCM = G::NCM;
  constexpr int NG = G::NG, NTG = G::NTG;
  const XkCaqrPipeArgs a = xk_pipe_args(ap);
  unsigned *sync = a.sync, *ab = sync + XP_ABORT * 16;
  const int tid = threadIdx.x;
  const int cidx_ = tid / NP, part_ = tid & (NP - 1);
  const bool panel = cidx_ < 16;
  const int npanels = (a.C1 + 15) / 16;
  // my group of this XCD's strips: tiles [grp NTG, grp NTG + nts); `item` counts inside the group from here on (NM = G::NMG)
  const int grp = NG > 1 ? item / NM : 0;
  item -= grp * NM;
  const int nts = NG > 1 ? min(NTG, NT - grp * NTG) : NT;
  const int base = xcc * NT + grp * NTG;
  const bool has_pending = !(xcc == 0 && grp == 0);       // (XCD 0's first root is the last level's pivot strip)
  const size_t SS = (size_t)16 * a.C1P;                   // doubles per strip
  const bool stamp = a.dbg && xcc == 0 && item == 0 && tid == 0;
  for (int k = 0; k < npanels; ++k) {
    const int c0 = 16 * k, trail = max(0, a.C1 - c0 - 16);
    const int cidx = xk_launder(cidx_), part = xk_launder(part_);
    // trailing columns per item: NCM sets of mh columns (whole quarter-waves); a trailing lane holds column col of the first set
    // and col + mh of the second
    const int mh = min(32, 4 * ((trail + 4 * NM * NCM - 1) / (4 * NM * NCM))), mch = mh * NCM;
    const bool active = item == 0 || item * mch < trail;
    const int col = panel ? c0 + cidx : c0 + 16 + item * mch + (cidx - 16);
    const int col2 = col + mh;
    const bool mine = active && col < a.C1 && (panel || cidx - 16 < mh);
    const bool mine2 = NCM > 1 && active && !panel && cidx - 16 < mh && col2 < a.C1;
    const int nsteps = (a.C1 - c0 < 16) ? a.C1 - c0 : 16;
    const bool full = nsteps == 16;
    const unsigned epoch = (unsigned)(k + 1);
    const size_t slab = (size_t)k * XK_PIPE_RLS + xcc * NG + grp;
    double b[RM], b2[NCM > 1 ? RM : 2];
    // the pending strip: what the last level left of this XCD's root of the previous panel (XCD 0's root is its pivot strip)
    b[0] = 0.0;
    b2[0] = 0.0;
    int ncl_prev, lchalf_prev, lsplit_prev;                 // last-level workgroups of panel k - 1
    xk_pipe_lastcut(a.C1 - c0, NCL, ncl_prev, lchalf_prev, lsplit_prev);
    lsplit_prev = min(XK_PIPE_NLW, lsplit_prev);
    const size_t pslab = (size_t)(k - 1) * XK_PIPE_RLS + xcc * NG + grp;
    if (k >= 1 && has_pending) {
#if XK_DATA_POLL
      // the slots say themselves when the last level has written them (xk_xcd_sync.hip.h): no counter, no barrier in front of the load
      double pb1[1] = {0.0}, pb2[1] = {0.0};
      if (!xk_poll_slots<1>(pb1, a.X2 + pslab * SS + xk_blk(min(col, a.C1P - 1), part), 0, mine, ab, 4u)) *s_ok = 0u;
      if constexpr (NCM > 1) {
        if (!xk_poll_slots<1>(pb2, a.X2 + pslab * SS + xk_blk(min(col2, a.C1P - 1), part), 0, mine2, ab, 4u)) *s_ok = 0u;
      }
      __syncthreads();
      if (*s_ok == 0u) return false;                         // (s_ok: 1 on entry, only ever cleared -- uniform verdict, no second barrier)
      if (mine) b[0] = pb1[0];
      if constexpr (NCM > 1) b2[0] = mine2 ? pb2[0] : 0.0;
      if (a.dbg && tid == 0) a.dbg[4096 + ((xcc * NG + grp) * NM + item) * 32 + k] = wall_clock64();
#else
      if (!xk_pipe_wait(sync + (XP_P_CNT + k - 1) * 16, (unsigned)lsplit_prev, ab, 4u, s_ok)) return false;
      if (mine) b[0] = xk_ld_sc1(a.X2 + pslab * SS + xk_blk(col, part));
#endif
    }
#pragma unroll
    for (int s = 1; s < RM; ++s) b[s] = 0.0;
    if constexpr (NCM > 1) {
#pragma unroll
      for (int s = 1; s < RM; ++s) b2[s] = 0.0;
#if !XK_DATA_POLL
      if (k >= 1 && has_pending && mine2) b2[0] = xk_ld_sc1(a.X2 + pslab * SS + xk_blk(col2, part));
#endif
    }
    double *g02 = a.S + (size_t)base * SS + xk_blk(min(col2, a.C1P - 1), part);
    double *x12 = a.X1 + slab * SS + xk_blk(min(col2, a.C1P - 1), part);
    const size_t lane_off = panel ? xk_blk(cidx, part) : xk_blk(col, part);
    const size_t strip_step = panel ? 256 : SS;
    double *g0 = (panel ? a.PB + (size_t)base * 256 : a.S + (size_t)base * SS) + lane_off;
    double *x1 = panel ? a.X1P + slab * 256 + xk_blk(cidx, part) : a.X1 + slab * SS + xk_blk(col, part);
    const bool x1_mine = mine && (!panel || item == 0);
    bool ok = true;
    int loaded = 0;                                        // phases of the tiles' strips that are in registers
    // Wide strip loads (XK_PIPE_PF).  Rows 4 q .. 4 q + 3 of one strip over the four columns of a wave are ONE 128-byte line
    // (xk_blk), and only a quarter of the lanes hold rows of the phase: as 23 eight-byte loads per lane a phase costs the
    // workgroup 12 x 23 vector-memory instructions that move 128 bytes each -- ~16 clocks apiece in the CU's one address
    // unit, 1.5 us per phase, four times per panel, on the chain of both dependency loops.  Instead every wave asks for its 23
    // lines with THREE 16-bytes-per-lane loads that land straight in LDS (lane l: strip 8 j + l / 8, piece l % 8), and the
    // lanes of the phase pick their 23 values up from there.
    constexpr bool PF = XK_PIPE_PF && NG == 1 && NCM == 1 && NPH == 4 && NT <= 24 && xk_pbn<NPH>(1) == 4 && xk_pbn<NPH>(2) == 8 && xk_pbn<NPH>(3) == 12;
    const int ln = tid & 63, cid4 = cidx & ~3;
    const int wcol = panel ? c0 + cid4 : c0 + 16 + item * mch + (cid4 - 16);
    const bool wv_ok = active && (panel || cid4 - 16 < mh) && wcol < a.C1;
    double *pfw = pfbuf + (size_t)(tid >> 6) * (24 * 16);
    const unsigned pfw_lds = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)pfw);   // (LDS offset = low half of the flat address)
    const double *pfsrc = (panel ? a.PB + (size_t)base * 256 + xk_blk(cid4, 0) : a.S + (size_t)base * SS + xk_blk(wcol, 0)) +
                          (size_t)(ln >> 3) * strip_step + (ln & 7) * 2;
    auto phase = [&](auto qc) {
      constexpr int q = decltype(qc)::value;
      if (!ok) return;
#ifdef XK_PIPE_PROBE_M1PACE
      // TIMING PROBE ONLY (wrong results): all rows were fetched with phase 0, but every phase still WAITS for the tiles' rows
      // of its own -- separates what the waits (the tiles' pace) cost from what the loads cost
      if (q > 0 && loaded > q) {
        if (xk_pipe_wait_phases(sync + (XP_TQ_CNT + xcc) * 16, 8 * 16, q, NPH, (unsigned)NT * epoch, ab, 6u, s_ok) == 0) { ok = false; return; }
      }
#endif
      if (loaded <= q) {
#ifdef XK_PIPE_PROBE_M1ALL
        // TIMING PROBE ONLY (wrong results): every phase's rows are fetched with phase 0's -- what the first level would cost if
        // the load latency of phases 1..3 were hidden
        int av = xk_pipe_wait_phases(sync + (XP_TQ_CNT + xcc) * 16, 8 * 16, q, NPH, (unsigned)NT * epoch, ab, 6u, s_ok);
        if (av != 0) av = NPH;
#else
        const int av = xk_pipe_wait_phases(sync + (XP_TQ_CNT + xcc) * 16, 8 * 16, q, NPH, (unsigned)NT * epoch, ab, 6u, s_ok);
#endif
        if (av == 0) { ok = false; return; }
        if (stamp && q < 4) a.dbg[512 + 16 * k + 2 * q] = wall_clock64();
        if (a.dbg && tid == 0 && k == 0 && q == 0) a.dbg[46080 + (xcc * NG + grp) * NM + item] = wall_clock64();
        XK_INV_LOC();
        if constexpr (PF) {
          // (a first level that runs behind finds several phases complete: one round per phase, the LDS area holds one)
          for (int p = loaded; p < av; ++p) {
            if (wv_ok) {                                     // wave-uniform
              const double *src = pfsrc + (size_t)p * 16;
#pragma unroll
              for (int j = 0; j < 3; ++j)
                if (8 * j + (ln >> 3) < NT) xk_lds_dma16(src + (size_t)(8 * j) * strip_step, pfw_lds + (unsigned)j * 1024u);
              asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
              if (mine && part >= p * GS && part < (p + 1) * GS) {
                typedef const __attribute__((address_space(3))) double xk_lds_cd;
                xk_lds_cd *pl = (xk_lds_cd *)(pfw + (part - p * GS) * 4 + (cidx & 3));
#pragma unroll
                for (int s = 1; s <= NT; ++s) b[s] = pl[(s - 1) * 16];
              }
              if (p + 1 < av) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the area is rewritten by the next round
            }
          }
        } else {
          if (active && mine && part >= xk_pbn<NPH>(loaded) && part < xk_pbn<NPH>(av)) {
            double *g = xk_opaque(g0);
#pragma unroll
            for (int s = 1; s <= NTG; ++s)
              if (NG == 1 || s <= nts) b[s] = XK_LD_LOC(g + (size_t)(s - 1) * strip_step);
          }
        }
        if constexpr (NCM > 1) {
          if (mine2 && part >= xk_pbn<NPH>(loaded) && part < xk_pbn<NPH>(av)) {
            double *g = xk_opaque(g02);
#pragma unroll
            for (int s = 1; s <= NTG; ++s)
              if (NG == 1 || s <= nts) b2[s] = xk_ld_sc1(g + (size_t)(s - 1) * SS);
          }
        }
        loaded = av;
#ifdef XK_PIPE_M1PROBE
        // timing probe: when have the strips' rows landed?  (the wait sits on the chain: not a production build)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (stamp && q < 4) a.dbg[512 + 16 * k + 10 + q] = wall_clock64();
#endif
      }
      if (!active) return;
      if (panel) __builtin_amdgcn_s_setprio(3);
      // (the first barrier of the range, behind the loads above, is where the root's rows of the previous phase are counted in)
      auto hook = [&]() { xk_pipe_arrive(sync + (XP_X1_CNT + (q - 1) * XK_CAQR_MAXP + k) * 16); };
      if constexpr (NCM > 1)
        xk_pipe_range<0, xk_pbn<NPH>(q), xk_pbn<NPH>(q + 1), (q > 0 ? xk_pbn<NPH>(q) : -1), RM>(b, &b2, cidx, mine, mine2, part, nsteps, ubuf, sc, full && !XK_DATA_POLL, hook);
      else
        xk_pipe_range<0, xk_pbn<NPH>(q), xk_pbn<NPH>(q + 1), (q > 0 ? xk_pbn<NPH>(q) : -1), RM>(b, nullptr, cidx, mine, false, part, nsteps, ubuf, sc, full && !XK_DATA_POLL, hook);
      if (panel) __builtin_amdgcn_s_setprio(0);
      // rows [q GS, (q + 1) GS) of the root are final: out they go (write-through: the last level sits on other XCDs)
      if (q < NPH - 1 && x1_mine && part >= xk_pbn<NPH>(q) && part < xk_pbn<NPH>(q + 1)) xk_st_sc1(x1, (panel && part > cidx) ? 0.0 : b[0]);
      if constexpr (NCM > 1) {
        if (q < NPH - 1 && mine2 && part >= xk_pbn<NPH>(q) && part < xk_pbn<NPH>(q + 1)) xk_st_sc1(x12, b2[0]);
      }
      if (stamp && q < 4) a.dbg[512 + 16 * k + 2 * q + 1] = wall_clock64();
      if (a.dbg && tid == 0 && q == 0) a.dbg[4096 + ((xcc * NG + grp) * NM + item) * 32 + 16 + k] = wall_clock64();
      if (a.dbg && tid == 0 && q < 4) a.dbg[16640 + (((xcc * NG + grp) * NM + item) * 16 + k) * 4 + q] = wall_clock64();   // (skew_trace: every phase)
    };
    phase(std::integral_constant<int, 0>{});
    if constexpr (NPH >= 2) phase(std::integral_constant<int, 1>{});
    if constexpr (NPH >= 4) { phase(std::integral_constant<int, 2>{}); phase(std::integral_constant<int, 3>{}); }
    if constexpr (NPH >= 8) {
      phase(std::integral_constant<int, 4>{}); phase(std::integral_constant<int, 5>{});
      phase(std::integral_constant<int, 6>{}); phase(std::integral_constant<int, 7>{});
    }
    if (!ok) return false;
    if (mine) {
      if (!panel) {
        // the tiles' strips first (the tiles wait for them), then the last rows of the root
        double *g = xk_opaque(g0);
#pragma unroll
        for (int s = 1; s <= NTG; ++s)
          if (NG == 1 || s <= nts) g[(size_t)(s - 1) * strip_step] = b[s];
      }
      if (x1_mine && part >= xk_pbn<NPH>(NPH - 1)) xk_st_sc1(x1, (panel && part > cidx) ? 0.0 : b[0]);
    }
    if constexpr (NCM > 1) {
      if (mine2) {
        double *g = xk_opaque(g02);
#pragma unroll
        for (int s = 1; s <= NTG; ++s)
          if (NG == 1 || s <= nts) g[(size_t)(s - 1) * SS] = b2[s];
        if (part >= xk_pbn<NPH>(NPH - 1)) xk_st_sc1(x12, b2[0]);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      xk_pipe_arrive(sync + (XP_MB_CNT + xcc) * 16);
      if (!XK_DATA_POLL)
        for (int q = (full && active) ? NPH - 1 : 0; q < NPH; ++q) xk_pipe_arrive(sync + (XP_X1_CNT + q * XK_CAQR_MAXP + k) * 16);
    }
    if (stamp) a.dbg[512 + 16 * k + 8] = wall_clock64();
  }
  return true;
}

// ---- role L: the last merge level, one workgroup per XCD: the 8 of them share the trailing columns of the panel and factor
// its 16 columns redundantly.  Register s = the root of XCD s (register 0 is the pivot strip); what is left of registers
// 1..7 goes down to XCD s as the pending strip of its first level in the NEXT panel.  (That is a dependency loop -- last
// level -> first level -> last level -- but with the first level on CUs of its own it is shorter than the tiles' loop:
// the last rows of the roots leave the first level together with the tiles' strips, a few steps later the pending strips
// are out, and the next first level does not start before its tiles have published the first rows of their strips.)
// Wide systems (NCL = 2): a trailing thread holds TWO columns, XK_PIPE_NLW lchalf apart -- the second set only takes reflectors.
template <class G>
__device__ __noinline__ bool xk_pipe_last(XkPipeArgsPtr ap, int lidx, double *ubuf, double *sc, unsigned *s_ok) {
  constexpr int NP = 16, RL = G::RL, NPH = G::NPH, GS = 16 / NPH, NCL = G::NCL, NM = G::NM;
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
    // trailing columns per chunk: 8 chunks of <= 32 columns when they cover the range (one column per lane), else 8 NCL chunks
    // (workgroup x takes chunks x and x + 8: two columns per lane, slower steps -- only the first panels of a wide system)
    int ncl, lchalf, lsplit;                                                // chunks in use
    xk_pipe_lastcut(trail, NCL, ncl, lchalf, lsplit);
    if (lidx >= lsplit || lidx >= XK_PIPE_NLW) continue;
    const int cidx = xk_launder(cidx_), part = xk_launder(part_);
    const int col = panel ? c0 + cidx : c0 + 16 + lidx * lchalf + (cidx - 16);
    const int col2 = col + XK_PIPE_NLW * lchalf;
    const bool mine = col < a.C1 && (panel || cidx - 16 < lchalf);
    const bool mine2 = ncl > 1 && !panel && cidx - 16 < lchalf && col2 < a.C1;
    const int nsteps = (a.C1 - c0 < 16) ? a.C1 - c0 : 16;
    const double *src = panel ? a.X1P + (size_t)k * XK_PIPE_RLS * 256 + xk_blk(cidx, part) : a.X1 + (size_t)k * XK_PIPE_RLS * SS + xk_blk(col, part);
    const double *src2 = a.X1 + (size_t)k * XK_PIPE_RLS * SS + xk_blk(min(col2, a.C1P - 1), part);
    const size_t sstep = panel ? 256 : SS;
    double b[RL], b2[RL];
#pragma unroll
    for (int s = 0; s < RL; ++s) { b[s] = 0.0; b2[s] = 0.0; }
    bool ok = true;
    int loaded = 0;
    auto phase = [&](auto qc) {
      constexpr int q = decltype(qc)::value;
      if (!ok || nsteps <= xk_pbn<NPH>(q)) return;                 // (a short last panel: the roots' rows past its columns are zero)
#if XK_DATA_POLL
      {
        // rows [pb(q), pb(q + 1)) of the roots: the lanes that hold them load their slots until every one has been written
        const bool rows_q = part >= xk_pbn<NPH>(q) && part < xk_pbn<NPH>(q + 1);
        bool got = xk_poll_slots<RL>(b, src, sstep, rows_q && mine, ab, 5u);
        if (got && ncl > 1) got = xk_poll_slots<RL>(b2, src2, SS, rows_q && mine2, ab, 5u);
        if (!got) *s_ok = 0u;
        __syncthreads();
        if (*s_ok == 0u) { ok = false; return; }
        if (stamp && q < 4) a.dbg[1024 + 16 * k + 2 * q] = wall_clock64();
        if (a.dbg && tid == 0 && q < 2) a.dbg[8192 + lidx * 64 + 16 * q + k] = wall_clock64();
        if (a.dbg && tid == 0 && q < 4) a.dbg[32768 + ((lidx * 16 + k) * 4 + q) * 2] = wall_clock64();                       // roots of phase q in
        (void)loaded;
      }
#else
      if (loaded <= q) {
        const int av = xk_pipe_wait_phases(sync + (XP_X1_CNT + k) * 16, XK_CAQR_MAXP * 16, q, NPH, 8u * NM, ab, 5u, s_ok);
        if (av == 0) { ok = false; return; }
        if (stamp && q < 4) a.dbg[1024 + 16 * k + 2 * q] = wall_clock64();
        if (part >= xk_pbn<NPH>(loaded) && part < xk_pbn<NPH>(av)) {
          if (mine) {
#pragma unroll
            for (int s = 0; s < RL; ++s) b[s] = xk_ld_sc1(src + s * sstep);
          }
          if (mine2) {
#pragma unroll
            for (int s = 0; s < RL; ++s) b2[s] = xk_ld_sc1(src2 + s * SS);
          }
        }
        loaded = av;
      }
#endif
      if (panel) __builtin_amdgcn_s_setprio(3);
      xk_pipe_range<0, xk_pbn<NPH>(q), xk_pbn<NPH>(q + 1), -1, RL>(b, ncl > 1 ? &b2 : nullptr, panel ? cidx : 16, mine, mine2, part, nsteps, ubuf, sc, false, []() {});
      if (panel) __builtin_amdgcn_s_setprio(0);
      if (stamp && q < 4) a.dbg[1024 + 16 * k + 2 * q + 1] = wall_clock64();
      if (a.dbg && tid == 0 && q < 4) a.dbg[32768 + ((lidx * 16 + k) * 4 + q) * 2 + 1] = wall_clock64();                     // phase q done
    };
    phase(std::integral_constant<int, 0>{});
    if constexpr (NPH >= 2) phase(std::integral_constant<int, 1>{});
    if constexpr (NPH >= 4) { phase(std::integral_constant<int, 2>{}); phase(std::integral_constant<int, 3>{}); }
    if (!ok) return false;
    // (with the Kalman role on, the rows of R are a hand-off of their own -- XCD 7's role reads them as they become final:
    //  written through like every cross-XCD hand-off, counted in per panel)
    auto rout = [&](size_t at, double v) { if (a.kal) xk_st_sc1(a.Rout + at, v); else a.Rout[at] = v; };
    if (mine) {
      if (panel) {
        if (lidx == 0 && c0 + part < a.C1) rout((size_t)(c0 + part) * a.C1P + xk_pipe_srccol(a, col), (part > cidx) ? 0.0 : b[0]);
      } else {
        if (k + 1 < npanels) {
          double *dst = a.X2 + (size_t)k * XK_PIPE_RLS * SS + xk_blk(col, part);
#pragma unroll
          for (int s = 1; s < RL; ++s) xk_st_sc1(dst + s * SS, b[s]);
        }
        if (c0 + part < a.C1) rout((size_t)(c0 + part) * a.C1P + xk_pipe_srccol(a, col), b[0]);
      }
    }
    if (mine2) {
      if (k + 1 < npanels) {
        double *dst = a.X2 + (size_t)k * XK_PIPE_RLS * SS + xk_blk(col2, part);
#pragma unroll
        for (int s = 1; s < RL; ++s) xk_st_sc1(dst + s * SS, b2[s]);
      }
      if (c0 + part < a.C1) rout((size_t)(c0 + part) * a.C1P + xk_pipe_srccol(a, col2), b2[0]);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      if (!XK_DATA_POLL && k + 1 < npanels) xk_pipe_arrive(sync + (XP_P_CNT + k) * 16);
      if (a.kal) xk_pipe_arrive(sync + (XP_R_CNT + k) * 16);
    }
    if (stamp) a.dbg[1024 + 16 * k + 8] = wall_clock64();
    if (a.dbg && tid == 0) a.dbg[8192 + lidx * 64 + 32 + k] = wall_clock64();
  }
  if (stamp) a.dbg[1539] = wall_clock64();
  return true;
}


// ---- role K: the Kalman update (Updater::applyUpdate, src/x/ekf/updater.cpp:117-141, cov_update; correction_total through d's start)
// INSIDE the compression launch, on the one workgroup the last level does not need (XCD 7's).  The 16 rows of R that panel k
// finishes are rows 16 k .. 16 k + 15 of the compressed measurement matrix T -- final from then on -- and the compressed rows
// all carry the same noise sigma^2 I (vio_updater.cpp:508-509), so the update can be applied BLOCK BY BLOCK as the panels
// complete, each block against the covariance the blocks before it left:
//     W_k = T_k P          S_k = W_k T_k^T + sigma^2 I = L L^T          X_k = L^-1 [W_k | T_k d - z_k]
//     [P | d] -= X_k[:, :n]^T X_k                   (d = the correction so far: d += K_k (z_k - T_k d))
// which is the batch update (updater.cpp:124-133: S = H P H^T + R, K = P H^T S^-1, P = (I - K H) P, P = (P + P^T) / 2) in
// exact arithmetic -- the block Cholesky of S in bordered form -- and agrees with it to rounding (1e-14 on P at the headline
// size).  When the last panel's rows arrive, everything but a rank-4 update has been done beside the QR: the update adds
// a few microseconds to the launch instead of four launches and 52 us behind it.
// Layout: [P | d] lives ON THE CU for the whole launch, tiles of 16 x 16 in the MFMA C/D layout (n <= 206: the narrow geometry),
// column 207 = d.  Waves 1..11 hold tile column w, rows 0..191, in REGISTERS (12 tiles, 96 VGPRs); tile column 0 and tile column 12
// (state columns 192.., the d column) sit in LDS -- column 0 belongs to wave 0, which also runs the 16-pivot chains (the chain
// keeps the block and its inverse in 64 registers: next to 96 registers of tiles that does not fit, so the factor wave keeps
// none), tile (w, 12) is worked on by wave w, (12, 12) by wave 0.  Tile ROW 12 is not kept at all: it is column 12 transposed,
// and LDS reads a tile either way.  A C/D register is the B operand of A x tile as it is (f64 16x16x4: the tile's row index is the
// contraction index), so W_k's column tile w needs no data from other waves; column 12's partial tiles are summed in LDS in a
// fixed order.  P is symmetrised when loaded and stays symmetric bit for bit (tile (I,J) and (J,I) run the same products in the
// same order), so the posterior is written through the transposed -- coalesced -- addresses.
#define XK_KAL_NB 12                                        // tile rows / columns of the main block (state indices 0..191)
#define XK_KAL_DC 207                                       // column of [P | d] that holds d
#define XK_KAL_TLD 226                                      // row stride of T_k in LDS: = 2 mod 32 doubles, operand gathers hit 32 banks
#define XK_KAL_LDS (16 * XK_KAL_TLD + 13 * 288 + 13 * 256 + 13 * 256 + 12 * 256 + 256 + 256 + 16 * 17 + 16)
typedef __attribute__((address_space(3))) double xk_ldsd;   // LDS, said out loud: through a generic pointer every access of the role
                                                            // became a FLAT load (64-bit addresses, no immediate offsets, vmcnt waits)
struct XkKalLds {
  xk_ldsd *Tl, *Wt, *part, *Pc, *P0, *W12s, *dbuf, *Ls, *zcol, *Sp, *Xs;
  __device__ __forceinline__ explicit XkKalLds(xk_ldsd *base) {
    Tl = base; Wt = Tl + 16 * XK_KAL_TLD; part = Wt + 13 * 288; Pc = part + 13 * 256; P0 = Pc + 13 * 256; W12s = P0 + 12 * 256;
    dbuf = W12s + 256; Ls = dbuf + 256; zcol = Ls + 16 * 17;
    Sp = Wt;      // a wave's partial S tile lands where it turned its W tile (stride 288)
    Xs = part;    // the partial tiles are dead when the X tiles are written
  }
};
// the pivot chain of xk_chol16.hip.h on LDS-typed operands.  NPIV = 4: a block whose rows past the fourth are padding (S there is
// sigma^2 I, the rows of W are zero): four pivots do, the other rows of the inverse stay the identity's and multiply zeros
template <int NPIV>
__device__ __forceinline__ bool xk_kal_chol16(const xk_ldsd *blk, xk_ldsd *linv, int lane) {
  const int tt = lane & 15;
  double v[16], w[16];
#pragma unroll
  for (int q = 0; q < 16; ++q) { v[q] = blk[16 * tt + q]; w[q] = (tt == q) ? 1.0 : 0.0; }
  bool bad = false;
  xk_chol16_step<0>(v, w, bad); xk_chol16_step<1>(v, w, bad); xk_chol16_step<2>(v, w, bad); xk_chol16_step<3>(v, w, bad);
  if constexpr (NPIV > 4) {
    xk_chol16_step<4>(v, w, bad); xk_chol16_step<5>(v, w, bad); xk_chol16_step<6>(v, w, bad); xk_chol16_step<7>(v, w, bad);
    xk_chol16_step<8>(v, w, bad); xk_chol16_step<9>(v, w, bad); xk_chol16_step<10>(v, w, bad); xk_chol16_step<11>(v, w, bad);
    xk_chol16_step<12>(v, w, bad); xk_chol16_step<13>(v, w, bad); xk_chol16_step<14>(v, w, bad); xk_chol16_step<15>(v, w, bad);
  }
  if (lane < 16) {
#pragma unroll
    for (int q = 0; q < 16; ++q) linv[q * 17 + tt] = w[q];
  }
  return bad;
}
// [P | d] -= X^T X on one wave's tiles, NQ k-steps (rows of X past 4 NQ are zero)
template <bool REG, int NQ>
__device__ __forceinline__ void xk_kal_downdate(const XkKalLds &m, xk_d4 (&Pt)[XK_KAL_NB], const xk_d4 &X, int wave, int lane) {
  constexpr int NB = XK_KAL_NB;
  const xk_ldsd *xl = m.Xs + lane;
  xk_ldsd *p0 = m.P0 + lane;
#pragma unroll
  for (int I = 0; I < NB; ++I) {
    double xc[NQ];
    xk_d4 c;
#pragma unroll
    for (int q = 0; q < NQ; ++q) xc[q] = xl[I * 256 + 64 * q];
#pragma unroll
    for (int r = 0; r < 4; ++r) c[r] = REG ? Pt[I][r] : p0[I * 256 + 64 * r];
#pragma unroll
    for (int q = 0; q < NQ; ++q) c = __builtin_amdgcn_mfma_f64_16x16x4f64(-xc[q], X[q], c, 0, 0, 0);
    if (REG) Pt[I] = c;
    else {
#pragma unroll
      for (int r = 0; r < 4; ++r) p0[I * 256 + 64 * r] = c[r];
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  xk_d4 c12, ccc;
#pragma unroll
  for (int r = 0; r < 4; ++r) { c12[r] = m.Pc[wave * 256 + 64 * r + lane]; ccc[r] = REG ? 0.0 : m.Pc[12 * 256 + 64 * r + lane]; }
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    c12 = __builtin_amdgcn_mfma_f64_16x16x4f64(-X[q], xl[12 * 256 + 64 * q], c12, 0, 0, 0);
    if (!REG) ccc = __builtin_amdgcn_mfma_f64_16x16x4f64(-xl[12 * 256 + 64 * q], xl[12 * 256 + 64 * q], ccc, 0, 0, 0);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) { m.Pc[wave * 256 + 64 * r + lane] = c12[r]; if (!REG) m.Pc[12 * 256 + 64 * r + lane] = ccc[r]; }
}
// One block of the update for one wave.  REG: tile column `wave` in registers (Pt), else in LDS (m.P0: wave 0, the factor wave).
template <bool REG>
__device__ __forceinline__ void xk_kal_block(const XkKalLds &m, xk_d4 (&Pt)[XK_KAL_NB], int k, int nrows, int n, double sigma2, int wave, int lane,
                                             bool &bad, long long *stp) {
#define XK_KSTAMP(i) do { if (stp && lane == 0) stp[i] = wall_clock64(); } while (0)
  constexpr int NB = XK_KAL_NB, TLD = XK_KAL_TLD;
  const int li = lane & 15, lk = lane >> 4;
  const int nq = (nrows + 3) >> 2;                          // k-steps that hold rows of this block
  const xk_ldsd *ta = m.Tl + li * TLD + lk;                 // A operand: T_k[m = li][state 4 q + lk + ..]
  const xk_ldsd *p0 = m.P0 + lane;
  // ---- W_k: my column tile (operands of the NEXT tile in flight behind this tile's MFMAs; the scheduling barriers keep the
  // scheduler from hoisting every operand load above the first MFMA, a hundred registers next to the tiles')
  // T_k is zero in the state columns of tile rows I < k: one jump into a straight line of tiles k .. 11 (a branch per tile made
  // every tile a basic block of its own and the accumulators a chain of phis)
  xk_d4 w0 = {0, 0, 0, 0}, w1 = {0, 0, 0, 0};
#define XK_KAL_WTILE(I)                                                                                                   \
  {                                                                                                                       \
    double ac[4], bc[4];                                                                                                  \
    _Pragma("unroll") for (int q = 0; q < 4; ++q) { ac[q] = ta[16 * (I) + 4 * q]; bc[q] = REG ? Pt[I][q] : p0[(I) * 256 + 64 * q]; } \
    _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                                       \
      if ((I) & 1) w1 = __builtin_amdgcn_mfma_f64_16x16x4f64(ac[q], bc[q], w1, 0, 0, 0);                                  \
      else w0 = __builtin_amdgcn_mfma_f64_16x16x4f64(ac[q], bc[q], w0, 0, 0, 0);                                          \
    }                                                                                                                     \
  }
  switch (k) {
    case 0: XK_KAL_WTILE(0) [[fallthrough]];
    case 1: XK_KAL_WTILE(1) [[fallthrough]];
    case 2: XK_KAL_WTILE(2) [[fallthrough]];
    case 3: XK_KAL_WTILE(3) [[fallthrough]];
    case 4: XK_KAL_WTILE(4) [[fallthrough]];
    case 5: XK_KAL_WTILE(5) [[fallthrough]];
    case 6: XK_KAL_WTILE(6) [[fallthrough]];
    case 7: XK_KAL_WTILE(7) [[fallthrough]];
    case 8: XK_KAL_WTILE(8) [[fallthrough]];
    case 9: XK_KAL_WTILE(9) [[fallthrough]];
    case 10: XK_KAL_WTILE(10) [[fallthrough]];
    default: XK_KAL_WTILE(11)
  }
#undef XK_KAL_WTILE
  if (n > 192) {                                            // state rows 192..n-1: tile (12, w) = tile (w, 12)^T, read transposed from LDS
#pragma unroll
    for (int q = 0; q < 4; ++q) w1 = __builtin_amdgcn_mfma_f64_16x16x4f64(ta[192 + 4 * q], m.Pc[wave * 256 + 16 * li + 4 * q + lk], w1, 0, 0, 0);
  }
  __builtin_amdgcn_sched_barrier(0);
  XK_KSTAMP(2);
  xk_d4 W = w0 + w1;
  {   // my share of tile column 12: tile (w, 12) takes T's block w; wave 0 also (12, 12)
    xk_d4 x = {0, 0, 0, 0};
    if (wave >= k) {
#pragma unroll
      for (int q = 0; q < 4; ++q) x = __builtin_amdgcn_mfma_f64_16x16x4f64(ta[16 * wave + 4 * q], m.Pc[wave * 256 + 64 * q + lane], x, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) m.part[wave * 256 + 64 * r + lane] = x[r];
    if (!REG) {
      xk_d4 y = {0, 0, 0, 0};
#pragma unroll
      for (int q = 0; q < 4; ++q) y = __builtin_amdgcn_mfma_f64_16x16x4f64(ta[192 + 4 * q], m.Pc[12 * 256 + 64 * q + lane], y, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; ++r) m.part[12 * 256 + 64 * r + lane] = y[r];
    }
  }
  // ---- my share of S_k = W_k T_k^T: the contraction runs over the COLUMNS of my W tile -- one trip through LDS turns it
  auto spart = [&](const xk_d4 &Wc, int J) {
    xk_ldsd *wt = m.Wt + J * 288;
#pragma unroll
    for (int r = 0; r < 4; ++r) wt[(lk + 4 * r) * 18 + li] = Wc[r];
    xk_d4 sacc = {0, 0, 0, 0};
#pragma unroll
    for (int q = 0; q < 4; ++q) sacc = __builtin_amdgcn_mfma_f64_16x16x4f64(wt[li * 18 + 4 * q + lk], ta[16 * J + 4 * q], sacc, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 4; ++r) m.Sp[J * 288 + 64 * r + lane] = sacc[r];
  };
  __builtin_amdgcn_sched_barrier(0);
  spart(W, wave);
  __builtin_amdgcn_sched_barrier(0);
  XK_KSTAMP(3);
  __syncthreads();                                          // (1) the partial tiles of column 12
  XK_KSTAMP(4);
  if (REG && wave == 11) {
    xk_d4 W12 = {0, 0, 0, 0};
    for (int w = k; w < 13; ++w)                            // (tile rows < k meet zero columns of T_k: their partial tiles are zero)
#pragma unroll
      for (int r = 0; r < 4; ++r) W12[r] += m.part[w * 256 + 64 * r + lane];
    if (li == XK_KAL_DC - 192) {                            // the d column: T_k d - z_k
#pragma unroll
      for (int r = 0; r < 4; ++r) W12[r] -= m.zcol[lk + 4 * r];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) m.W12s[64 * r + lane] = W12[r];
    spart(W12, 12);
  }
  __syncthreads();                                          // (2) the partial tiles of S_k
  XK_KSTAMP(5);
  if (!REG) {
    xk_d4 S = {0, 0, 0, 0};
    for (int J = k; J < 13; ++J)                            // (likewise: T_k is zero in the state columns of tiles < k)
#pragma unroll
      for (int r = 0; r < 4; ++r) S[r] += m.Sp[J * 288 + 64 * r + lane];
#pragma unroll
    for (int r = 0; r < 4; ++r) m.dbuf[16 * (lk + 4 * r) + li] = S[r] + ((lk + 4 * r == li) ? sigma2 : 0.0);
    if (nrows <= 4 ? xk_kal_chol16<4>(m.dbuf, m.Ls, lane) : xk_kal_chol16<16>(m.dbuf, m.Ls, lane)) bad = true;
  }
  XK_KSTAMP(6);
  __syncthreads();                                          // (3) L^-1
  XK_KSTAMP(7);
  double lv[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) lv[q] = m.Ls[li * 17 + 4 * q + lk];
  xk_d4 X = {0, 0, 0, 0};
#pragma unroll
  for (int q = 0; q < 4; ++q) X = __builtin_amdgcn_mfma_f64_16x16x4f64(lv[q], W[q], X, 0, 0, 0);
#pragma unroll
  for (int r = 0; r < 4; ++r) m.Xs[wave * 256 + 64 * r + lane] = X[r];
  if (REG && wave == 11) {
    xk_d4 X12 = {0, 0, 0, 0};
#pragma unroll
    for (int q = 0; q < 4; ++q) X12 = __builtin_amdgcn_mfma_f64_16x16x4f64(lv[q], m.W12s[64 * q + lane], X12, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 4; ++r) m.Xs[12 * 256 + 64 * r + lane] = X12[r];
  }
  __syncthreads();                                          // (4) the X tiles
  XK_KSTAMP(8);
  // ---- [P | d] -= X^T X (rows of X past nrows are zero: their k-steps are skipped)
  if (nq == 1) xk_kal_downdate<REG, 1>(m, Pt, X, wave, lane);   // (the headline's last block: 4 rows)
  else xk_kal_downdate<REG, 4>(m, Pt, X, wave, lane);
  XK_KSTAMP(9);
  __syncthreads();                                          // (5) Tl, Xs are rewritten by the next block
  XK_KSTAMP(10);
#undef XK_KSTAMP
}

template <class G>
__device__ __noinline__ bool xk_pipe_kalman(XkPipeArgsPtr ap, xk_ldsd *kb, unsigned *s_ok) {
  constexpr int NB = XK_KAL_NB, TLD = XK_KAL_TLD, NCL = G::NCL;
  const XkCaqrPipeArgs a = xk_pipe_args(ap);
  unsigned *sync = a.sync, *ab = sync + XP_ABORT * 16;
  const int tid = threadIdx.x, lane_ = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = a.kn, na = a.C1 - 1, npanels = (a.C1 + 15) / 16;
  // the prior, symmetrised; column 207 = d, the correction so far.  The blocks run d += K_k (z_k - T_k d); started at -correction_total
  // that recurrence ends at K (res + H ct) - ct, which is what updater.cpp:128 asks of an IEKF pass (ct = 0: the plain update)
  auto ldsym = [&](int row, int col) {
    if (col == XK_KAL_DC) return (a.ct && row < n) ? -a.ct[row] : 0.0;
    return (row < n && col < n) ? 0.5 * (a.Pin[(size_t)row + (size_t)col * n] + a.Pin[(size_t)col + (size_t)row * n]) : 0.0;
  };
  // what a block starts with: the rows of R the last level has just finished, by STATE index -- Tl[m][15 + c] = R[c0 + m][c],
  // c0 <= c < na (zero elsewhere), z_k[m] = R[c0 + m][na]; false: the launch is giving up
  auto fetch = [&](const XkKalLds &m, int k, int &nrows) -> bool {
    const int c0 = 16 * k;
    nrows = min(16, na - c0);
    int ncl, lchalf, lsplit;
    xk_pipe_lastcut(max(0, a.C1 - c0 - 16), NCL, ncl, lchalf, lsplit);
    if (!xk_pipe_wait(sync + (XP_R_CNT + k) * 16, (unsigned)min(XK_PIPE_NLW, lsplit), ab, 8u, s_ok)) return false;
    double tv[5];                                           // (every load first, then the stores: one round trip to the other XCDs' rows)
#pragma unroll
    for (int u = 0; u < 5; ++u) {
      const int idx = tid + u * XK_PIPE_THREADS, mr = idx / 208, st = idx - 208 * mr, c = st - 15;
      tv[u] = (idx < 16 * 208 && mr < nrows && c >= c0 && c < na) ? xk_ld_sc1(a.Rout + (size_t)(c0 + mr) * a.C1P + c) : 0.0;
    }
#pragma unroll
    for (int u = 0; u < 5; ++u) {
      const int idx = tid + u * XK_PIPE_THREADS, mr = idx / 208, st = idx - 208 * mr;
      if (idx < 16 * 208) m.Tl[mr * TLD + st] = tv[u];
    }
    if (tid < 16) m.zcol[tid] = (tid < nrows) ? xk_ld_sc1(a.Rout + (size_t)(c0 + tid) * a.C1P + na) : 0.0;
    __syncthreads();
    return true;
  };
  // The end of the role.  The correction -- column 207 of tile column 12, sixteen entries per wave -- is gathered in LDS; the
  // factor wave sends it to the host with system-scope stores and, once they are complete (~3 us: they cross PCIe), the marker:
  // the host has what it waits for while the other waves' tiles of the posterior are on their way out (the next kernel sees
  // those through the kernel boundary).
  auto publish = [&](const XkKalLds &m, int lane) {
    const int li = lane & 15, lk = lane >> 4;
    if (li == XK_KAL_DC - 192) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        m.dbuf[16 * wave + lk + 4 * r] = m.Pc[wave * 256 + 64 * r + lane];
        if (wave == 0) m.dbuf[192 + lk + 4 * r] = m.Pc[12 * 256 + 64 * r + lane];
      }
    }
    __syncthreads();
    if (wave == 0) {
      for (int i = lane; i < n; i += 64) xk_st_sys(a.corr + i, m.dbuf[i]);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (lane == 0 && a.done_flag) __hip_atomic_store(a.done_flag, a.done_seq, XK_MARKER_ORDER, __HIP_MEMORY_SCOPE_SYSTEM);   // (xk_xcd_sync.hip.h)
    }
  };
  // P is symmetric bit for bit: element (row, col) goes to the address of (col, row), sixteen lanes to a 128-byte line; the
  // lane's part of the address is formed once, the tile's part is uniform
  auto store_tile = [&](int I, int J, int lane, const xk_d4 &v, bool both) {
    const int li = lane & 15, lk = lane >> 4, col = 16 * J + li;
    if (col >= n) return;
    double *pb = a.Pout + (size_t)col + (size_t)lk * n;
    double *pm = a.Pout + (size_t)lk + (size_t)col * n;     // the mirror image (tile column 12 stands for tile row 12 as well)
    const bool whole = 16 * I + 16 <= n;                    // (uniform)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (whole || 16 * I + lk + 4 * r < n) {
        pb[(size_t)((16 * I + 4 * r) * n)] = v[r];
        if (both) pm[16 * I + 4 * r] = v[r];
      }
    }
  };
  const int nblocks = min(npanels, (na + 15) / 16);         // (a last panel that holds the residual column only brings no rows)
  bool bad = false;
  if (wave == 0) {
    // ---- the factor wave: tile column 0 and tile (12, 12) in LDS, the pivot chains in registers
    {
      const XkKalLds m(kb);
      const int lane = lane_, li = lane & 15, lk = lane >> 4;
#pragma unroll
      for (int I = 0; I < NB; ++I)
#pragma unroll
        for (int r = 0; r < 4; ++r) m.P0[I * 256 + 64 * r + lane] = ldsym(16 * I + lk + 4 * r, li);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        m.Pc[64 * r + lane] = ldsym(lk + 4 * r, 192 + li);
        m.Pc[12 * 256 + 64 * r + lane] = ldsym(192 + lk + 4 * r, 192 + li);
      }
    }
    xk_d4 none[NB];
    for (int k = 0; k < nblocks; ++k) {
      const XkKalLds m(xk_opaque(kb));
      int nrows;
      long long *stp = a.dbg ? a.dbg + 2048 + 16 * k : nullptr;
      if (stp && lane_ == 0) stp[0] = wall_clock64();
      if (!fetch(m, k, nrows)) return false;
      if (stp && lane_ == 0) stp[1] = wall_clock64();
      xk_kal_block<false>(m, none, k, nrows, n, a.sigma2, 0, xk_launder(lane_), bad, stp);
    }
    const XkKalLds m(kb);
    const int lane = lane_;
    if (bad && lane == 0) __hip_atomic_store(a.status, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // XK_ESINGULAR
    if (a.dbg && lane == 0) a.dbg[2048 + 16 * 31 + 1] = wall_clock64();
    publish(m, lane);
    if (a.dbg && lane == 0) a.dbg[2048 + 16 * 31 + 2] = wall_clock64();
    {   // (tile column 0 is in LDS: waves 1..11 store it after their own tiles, this wave has the host to talk to)
      xk_d4 cc;
#pragma unroll
      for (int r = 0; r < 4; ++r) cc[r] = m.Pc[12 * 256 + 64 * r + lane];
      store_tile(12, 12, lane, cc, false);
    }
    if (a.dbg && lane == 0) a.dbg[2048 + 16 * 31] = wall_clock64();
    return true;
  }
  // ---- waves 1..11: tile column `wave` in registers
  xk_d4 Pt[NB];
  {
    const XkKalLds m(kb);
    const int lane = lane_, li = lane & 15, lk = lane >> 4;
#pragma unroll
    for (int I = 0; I < NB; ++I) {
#pragma unroll
      for (int r = 0; r < 4; ++r) Pt[I][r] = ldsym(16 * I + lk + 4 * r, 16 * wave + li);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) m.Pc[wave * 256 + 64 * r + lane] = ldsym(16 * wave + lk + 4 * r, 192 + li);
  }
  for (int k = 0; k < nblocks; ++k) {
    const XkKalLds m(xk_opaque(kb));
    int nrows;
    if (!fetch(m, k, nrows)) return false;
    long long *stp = (a.dbg && wave == 5) ? a.dbg + 2560 + 16 * k : nullptr;
    xk_kal_block<true>(m, Pt, k, nrows, n, a.sigma2, wave, xk_launder(lane_), bad, stp);
  }
  const XkKalLds m(kb);
  const int lane = lane_;
  publish(m, lane);
#pragma unroll
  for (int I = 0; I < NB; ++I) {
    store_tile(I, wave, lane, Pt[I], false);
    __builtin_amdgcn_sched_barrier(0);
  }
  {
    xk_d4 c, c0t, p0a, p0b;                                  // my tile of column 12; and of the factor wave's column 0: tile wave (+ tile 0 / tile 12's
    const int e = (wave == 1) ? 0 : 0;                      //  stand-in (0, 12) for waves 1 / 11)
    (void)e;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      c[r] = m.Pc[wave * 256 + 64 * r + lane];
      p0a[r] = m.P0[wave * 256 + 64 * r + lane];
      p0b[r] = m.P0[64 * r + lane];
      c0t[r] = m.Pc[64 * r + lane];
    }
    store_tile(wave, 12, lane, c, true);
    store_tile(wave, 0, lane, p0a, false);
    if (wave == 1) store_tile(0, 0, lane, p0b, false);
    if (wave == 11) store_tile(0, 12, lane, c0t, true);
  }
  return true;
}

template <class G>
__global__ __launch_bounds__(XK_PIPE_THREADS) void xk_caqr_pipe(XkCaqrPipeArgs a) {
  constexpr int RPL = G::RPL, NT = G::NT, NM = G::NM, RM = G::RM, LPC = G::LPC;
  constexpr int LDS_T = 2 * LPC * (RPL + 2), LDS_M = 2 * 16 * (RM + 2), LDS_L = 2 * 16 * (G::RL + 2);
  constexpr int LDS_MAX = LDS_T > LDS_M ? (LDS_T > LDS_L ? LDS_T : LDS_L) : (LDS_M > LDS_L ? LDS_M : LDS_L);
  __shared__ __attribute__((aligned(16))) double ubuf[LDS_MAX];
  __shared__ __attribute__((aligned(16))) double sc[2 * 4];
  __shared__ unsigned s_slot, s_ok;
  __shared__ int rp_pre[XK_PIPE_SLOTS_MAX + 1], rp_rows[LPC * RPL];   // tile workgroups: prefix sums of the accepted rows, my rows
  // landing area of the first level's load-to-LDS prefetch: per wave (4 columns) 24 strips x 4 rows x 4 columns
  __shared__ __attribute__((aligned(16))) double pfbuf[XK_PIPE_PF ? (XK_PIPE_THREADS / 64) * 24 * 16 : 2];
  constexpr bool KAL = G::COLS <= 192 && G::LPC == 4 && G::RPL <= 40;   // the Kalman role keeps [P | d] in registers: n <= 206 (not the tail geometries)
  __shared__ __attribute__((aligned(16))) double kbuf[KAL ? XK_KAL_LDS : 2];
  // tile workgroups: landing area of the factor records their rows are formed from (the Kalman role's LDS where there is one)
  constexpr int HB_OWN = KAL ? 2 : 16 * 1024;
  __shared__ __attribute__((aligned(16))) double hbuf_own[HB_OWN];
  double *hbuf = KAL ? kbuf : hbuf_own;
  constexpr int HB = KAL ? XK_KAL_LDS : HB_OWN;
  unsigned *sync = a.sync, *ab = sync + XP_ABORT * 16;
  const XkPipeArgsPtr ap = (XkPipeArgsPtr)__builtin_amdgcn_kernarg_segment_ptr();
  const unsigned xcc = xk_xcc_id();
  const long long t_entry = a.dbg ? wall_clock64() : 0;
  // Placement census: a workgroup reads the XCD it runs on (HW_REG_XCC_ID) and takes the next slot there -- the dispatcher deals
  // workgroups round-robin over the XCDs, but where it starts depends on what ran before, block b -> XCD b % 8 does NOT hold --
  // and nobody waits for the census to complete; a 33rd arrival on one XCD raises the abort word
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
  if (a.test_stall && xcc == 3 && slot == 5) return;
  if (a.dbg && slot >= NT && threadIdx.x == 0) a.dbg[45056 + ((int)xcc * 32 + slot) * 2] = wall_clock64();
#if XK_DATA_POLL
  if (slot >= NT && a.Xnext) {
    // The workgroups that have nothing to do until the tiles publish their first rows re-arm the OTHER set of cross-XCD slabs
    // for the next launch: X1 | X2 ([panel][16 strips][16 x C1P]: panel k is only ever read at columns >= 16 k, i.e. from offset
    // 256 k of a strip on) then X1P ([panel][16][256]).  Plain stores: the kernel boundary publishes them.
    // Strip by strip (2 x panels x 16 of them, dealt over these workgroups), 16 bytes per store, 32-bit index arithmetic: the flat
    // loop this replaces decided per element, with a 64-bit division and a remainder, whether it is ever read -- 20 us per
    // workgroup, which the first level of panels 0..2 started late by (LAB round 5).
    const int SSl = 16 * a.C1P, npl = (a.C1 + 15) / 16, nseg = 2 * npl * XK_PIPE_RLS;
    const int nwg = 8 * (32 - NT), wg = (int)xcc * (32 - NT) + (slot - NT), tid = (int)threadIdx.x;
    const double ny = xk_notyet();
    const xk_d2 ny2 = {ny, ny};
    for (int sg = wg; sg < nseg; sg += nwg) {
      const int k = (sg / XK_PIPE_RLS) % npl;               // the strip's panel
      double *p = a.Xnext + (size_t)sg * SSl;
      for (int e = 256 * k + 2 * tid; e < SSl; e += 2 * XK_PIPE_THREADS) *reinterpret_cast<xk_d2 *>(p + e) = ny2;
    }
    double *pp = a.Xnext + (size_t)nseg * SSl;
    const int n1p = npl * XK_PIPE_RLS * 256;
    for (int e = 2 * (wg * XK_PIPE_THREADS + tid); e < n1p; e += 2 * nwg * XK_PIPE_THREADS) *reinterpret_cast<xk_d2 *>(pp + e) = ny2;
  }
#endif
  if (a.dbg && slot >= NT && threadIdx.x == 0) a.dbg[45056 + ((int)xcc * 32 + slot) * 2 + 1] = wall_clock64();
  bool ok;
  if (slot < NT) {
    int rows_acc;
    const int TR = xk_pipe_rowplan<G>(a, (int)xcc * NT + slot, rp_pre, rp_rows, &rows_acc);
    // how many rows passed the gates: the host picks the next launch's geometry by it (status word 2, pinned host memory)
    if (xcc == 0 && slot == 0 && threadIdx.x == 0) __hip_atomic_store(a.status + 2, (a.acc_tag << 15) | min(rows_acc, 0x7fff), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    const int nvalid = min(max(rows_acc - ((int)xcc * NT + slot) * TR, 0), TR);   // rows of my tile that exist
    if (TR == 0) {                                         // more accepted rows than the tiles hold: everybody learns it from the abort word
      if (threadIdx.x == 0) { __hip_atomic_store(ab, 9u, XK_RLX_AGENT); a.status[1] = 9; }
      ok = false;
    } else if (G::RPLT < G::RPLS && TR <= LPC * G::RPLT) ok = xk_pipe_tile<G, G::RPLT>(ap, (int)xcc, slot, t_entry, ubuf, sc, &s_ok, rp_rows, hbuf, HB, nvalid);
    else if (G::RPLS < RPL && TR <= LPC * G::RPLS) ok = xk_pipe_tile<G, G::RPLS>(ap, (int)xcc, slot, t_entry, ubuf, sc, &s_ok, rp_rows, hbuf, HB, nvalid);
    else ok = xk_pipe_tile<G, RPL>(ap, (int)xcc, slot, t_entry, ubuf, sc, &s_ok, rp_rows, hbuf, HB, nvalid);
  }
  else if (slot < NT + NM) ok = xk_pipe_first<G>(ap, (int)xcc, slot - NT, ubuf, sc, &s_ok, pfbuf);
  else {
    // nothing to do until the first roots arrive: leave the other set of sync words zeroed for the next launch
    for (int i = (int)xcc * XK_PIPE_THREADS + threadIdx.x; i < XP_WORDS * 16; i += 8 * XK_PIPE_THREADS) a.sync_next[i] = 0u;
    if ((int)xcc < XK_PIPE_NLW) ok = xk_pipe_last<G>(ap, (int)xcc, ubuf, sc, &s_ok);
    else {
      ok = true;
      if constexpr (KAL) {
        if (a.kal) {
          ok = xk_pipe_kalman<G>(ap, (xk_ldsd *)kbuf, &s_ok);
          // A role that gave up has not told the host anything yet (a finished one wrote the correction, then the marker, from
          // inside -- relaxed system-scope stores, acknowledged (vmcnt) before the marker is stored; a system-scope RELEASE on the
          // marker is a write-back of this XCD's whole L2 on the launch's tail and is what -DXK_SYNC_STRICT=1 builds: XK_MARKER_ORDER,
          // xk_xcd_sync.hip.h -- nothing but those words is for the host, the posterior is ordered by the stream)
          if (!ok) {
            if (threadIdx.x == 0) __hip_atomic_store(a.status + 1, (int)__hip_atomic_load(ab, XK_RLX_AGENT), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (threadIdx.x == 0 && a.done_flag) __hip_atomic_store(a.done_flag, a.done_seq, XK_MARKER_ORDER, __HIP_MEMORY_SCOPE_SYSTEM);
          }
          if (a.dbg && threadIdx.x == 0) a.dbg[2048 + 16 * 31 + 3] = wall_clock64();
          return;
        }
      }
    }
  }
  if (!ok && threadIdx.x == 0) a.status[1] = (int)__hip_atomic_load(ab, XK_RLX_AGENT);
}
