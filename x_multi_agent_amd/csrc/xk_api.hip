// xk_api.hip -- C ABI (include/xk.h) of the MI355X-native xVIO EKF-update engine.
// Host-side orchestration only: every number in a result is produced by the
// HIP kernels in xk_feature.hip.h / xk_linalg.hip.h / xk_ci.hip.h.
#include "../../include/xk.h"
#ifdef XK_LAB
#include "../../include/xk_lab.h"
#endif

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

#include "xk_chi2_table.h"
#include "xk_feature.hip.h"
#include "xk_slaminit.hip.h"
#include "xk_linalg.hip.h"
#include "xk_caqr_pipe.hip.h"
#include "xk_ci.hip.h"

#define XK_VERSION_NUM 200
#define XK_STAGE_SLOTS 8
#define XK_PDBG_WORDS 65536     // debug stamps of the single launch (lab build)

struct xk_handle {
  int device;
  hipStream_t stream;
  hipStream_t copy_stream;   // gate flags travel to the host beside the QR kernels, not between them
  hipEvent_t ev_flags, ev_flags_done;
  hipEvent_t ev[16];
  // capacities
  int N, Mmax, Kmax, n, na, C1, C1P, DB, ntiles_max;
  // staged problem
  int n_poses, K, M;
  size_t obs_cap;
  double *d_q, *d_p, *d_obs, *d_feat, *d_zlast;
  int *d_trk_off, *d_anchor, *d_tsz;
  double *d_P, *d_Pout;
  double *d_Psnap;        // xk_snapshot_P (slot of the caller)
  double *d_Psnap2;       // ... slot of the filter loop (x::Ekf saves the prior of an update the IMU thread may lap)
  double *d_fq;           // f_d, q_d of xk_cov_propagate
  double *d_chi95, *d_chi90;
  double *d_A;
  double *d_Hc;            // factor records of the MSCKF tracks (xk_feature.hip.h: XkFeatArgs::Hc), hc_stride doubles each, 64-row slots only
  int hc_stride;
  int opt_hlite;           // 1 (default): the per-feature kernel leaves factor records when the single launch is expected to run
  bool rows_compact;       // the last build left records, not tiles, for slots [0, K)
  int *d_tile_rows;
  double *d_panel[2];   // CAQR: 16 x 16 panel blocks of the even (tiles, level 2, ..) / odd merge levels
  int *d_inl, *d_inl_s, *d_gn;
  double *d_gam, *d_gam_s, *d_gpf;
  double *d_R;
  int nleaf, nlevels;   // of the last compression
  // single-launch CAQR (xk_caqr_pipe.hip.h): cross-XCD exchange slabs, XCD-local strips and panel blocks,
  // two sets of sync words (a launch uses one and zeroes the other for its successor)
  double *d_x1;            // both sets of the cross-XCD slabs (X1 | X2 | X1P each)
  size_t xslab_doubles;    // ... doubles per set
  double *d_rs, *d_rpb;
  unsigned *d_xsync;
  int xsync_phase;
  int pipe_rows_nominal;   // rows the last single launch was queued for, every track counted as accepted
  // Geometry with two first-level groups per XCD (XkPipeNarrow2: 152 tiles): taken when the rows expected to pass the gates fit it.
  // The expectation is the acceptance ratio the last single launch reported (status word 2) applied to this update's nominal rows.
  int opt_split;           // 0 never, 1 adaptive (default); lab: 2 whenever the NOMINAL rows fit, 3 always
  double acc_ratio;        // accepted / nominal rows of the last single launch (0: none yet)
  int pipe_tag;            // tag of the last single launch's accepted-rows word (status word 2)
  bool last_split;         // the last single launch used that geometry
  int split_backoff;       // updates for which it stays off after it found more rows than it holds
  int overflow_rows;       // a single launch of that many nominal rows found more accepted rows than its tiles hold: not tried again at that size
  // SPLIT compression (round 6; systems with SLAM features whose update cannot ride inside the launch, n > 206: BASELINE config 2).
  // The rows of MSCKF tracks are zero in the features' columns (msckf_update.cpp:412-416) and the features' own rows are 2 M in number:
  // only the tracks' rows need compressing, and only in the 6 N pose columns (+ the residual) -- a system of <= 199 columns instead of
  // 6 N + 3 M + 1.  T = [R1 | 0 | z1 ; H_slam | res_slam] (6 N + 2 M rows, T^T T = H^T H and T^T z = H^T res exactly as for the R of
  // the whole stack) goes to d_R2; xk_qr_compress, which hands out the reference's upper-triangular T_H, keeps compressing everything.
  double *d_R2;         // [C1P][C1P] row-major: rows [0, 6 N) = R1 of the tracks' rows, rows [6 N, 6 N + 2 M) = the SLAM rows as built
  int split_active;     // what the last launch_compress left in d_R2 (compressed_spec follows it): 0 nothing, 1 the split compression,
                        // 2 the SLAM rows alone, uncompressed (no track in the stack: rows <= columns, vio_updater.cpp:487 does not compress either)
  bool want_full_T;     // xk_qr_compress is running: compress everything into d_R
  int opt_slam_split;
  int opt_pipe_min_rows;   // nominal rows from which the single launch is queued (1; lab: XK_PIPE_MIN_ROWS)
  bool last_resident;   // the last launch_compress took the single-launch resident schedule
  bool last_pipe;       // ... the pipelined one (its sync words: a launch that gave up leaves them dirty)
  // Tall systems (128-row slots: windows of 34..64 poses, BASELINE config 3): the multi-launch schedule factors the first panels,
  // ONE launch of xk_caqr_pipe<XkPipeTail> the last <= 96 columns with every row in registers (round 6)
  bool tail_capable;    // decided at xk_create: 128-row slots, 256 CUs, the kernel fits a CU
  bool tail_ok;         // armed (cleared when a tail launch gave up; re-armed like the fast path)
  bool last_tail;       // the last launch_compress ended in such a launch
  int tail_backoff_len; // updates the tail stays off after it found more rows than it holds (reason 9): 64, doubling while that keeps happening
  int tail_clean, tail_backoff, opt_tail;   // opt_tail: 0 off, 1 (default) the plan that fits (192 columns in one or two launches, else 96 in one), 2 the 96-column launch only
  bool tail_four;       // the plan of this compression uses the 4-lanes-per-column geometry (<= 192 columns)
  long long *d_pdbg;
  long long *feat_dbg;  // probe builds only: per-workgroup phase stamps of xk_msckf_feature
  bool attr_slaminit, attr_feat_batch;   // hipFuncSetAttribute done for this handle's device
  int n_cu;
  bool persist_ok;      // cleared when a launch gave up (workgroups not co-resident): the multi-launch schedule takes over
  bool fast_capable;    // decided at xk_create: 256 CUs, one 768-thread workgroup of the single-launch kernels fits a CU
  int fast_giveups, fast_reason;   // launches that gave up so far / why the last one did (xk_caqr_status)
  int clean_classic, rearm_after;  // multi-launch updates since the last give-up / how many of them re-arm the fast path
  // experiment switches and test hooks of the compression, read from the environment ONCE at xk_create (the per-update path
  // calls no getenv); xk_set_option changes them on a live handle (tests do)
  int opt_resident, opt_poison, opt_test_stall, opt_tall26;
  int opt_kalman;          // the Kalman update inside the single launch (xk_pipe_kalman) where the geometry allows it
  bool last_fused;         // the last launch_compress also queued the Kalman update (posterior in d_Pout, correction written)
  bool compress_deferred;  // xk_build_compress_async queued the rows only: the compression waits for xk_apply_update, where the Kalman
                           // role can ride along on the covariance the applyCI entries in between have left (MULTI_UAV order)
  int fused_cov_update;    // what the queued pass was asked for (xk_build_compress_update[_pass]_async): xk_apply_update must ask the same
  bool fused_ct_zero;
  std::vector<double> *fused_ct;
  bool fused_pending;      // xk_build_compress_update_async ran: xk_apply_update only has to wait
  unsigned long long fused_seq;   // ... for this completion marker (0: for the stream)
  bool xsync_dirty;     // a pipelined launch gave up: its counters are mid-count, clear both sets before the next one
  bool have_rows, have_R;
  double sigma_img;
  // update workspace
  int CM, LDA;
  double *d_Maug, *d_X, *d_corr, *d_ct, *d_tmpH, *d_tmpS, *d_tmpP, *d_rdiag, *d_tmpz;
  double *h_win;        // host copy of the staged window lists (7 doubles per pose), see flush_window
  bool win_pending;     // ... which have not reached d_q / d_p yet
  bool win_valid;       // h_win holds the lists of the window in use
  unsigned *d_done_cnt;  // workgroup counter of the completion marker
  unsigned long long done_seq, done_seen, flags_after_seq;   // completion markers (XK_SPIN_DONE): launched / seen / launched when the gate flags were queued
  int *d_status;        // status words; they live in PINNED HOST memory (h_out + n): kernels write them only on failure
  double *h_out;        // pinned host, device-visible: [n] correction of xk_apply_update + the status words
  // CI / payload
  double *d_payload;
  double *d_ci;  // scratch for the CI kernels
  double *d_ciws;          // workspace of the device-resident CI round (lazily allocated)
  hipStream_t ci_stream[8];   // ... and its side streams: shared track j >= 1 runs its stages before the gate on ci_stream[j],
  hipEvent_t ci_fork, ci_join[8];   // next to track 0 on the engine's stream (forked and joined with events)
  XkFeatBatch *d_batch;    // per-agent descriptors of the batched feature launch, [8 tracks][8 agents]
  XkFeatBatch *h_batch;    // pinned staging of the same
  int *h_ci_cols;          // pinned: per shared track, the block columns of xk_scale_blocks [8][128]
  double *h_ci_w;          // pinned: per shared track, 1/w0 [8]
  int *h_trk_off;          // host copy of the staged track offsets
  // MSCKF-SLAM tracks (features being initialised this frame)
  int anchor_max;          // largest staged SLAM anchor index (rechecked against the staged window at build time)
  int K2;
  bool ms_built;           // their column-space rows on the device belong to the staged tracks
  int *d_trk2_off, *h_trk2_off, *d_inl2, *d_gn2;
  double *d_obs2, *d_gpf2, *d_W2, *d_gam2, *d_H1, *d_H2, *d_r1, *d_feat2;
  int *d_csr_i;            // sparse congruence operand: row pointers then column indices
  double *d_csr_v;         //   and values
  size_t csr_cap;          //   capacity in non-zeros
  // pinned staging ring for inputs copied to the device WITHOUT a host synchronisation (window, tracks, sparse operands):
  // a slot is reused XK_STAGE_SLOTS calls later, by which time an update's final synchronisation has long passed
  char *h_stage[XK_STAGE_SLOTS];
  int stage_since_sync;    // slots handed out since the stream was last known to be idle (stage_slot)
  size_t stage_bytes;
  int stage_next;
  bool flags_direct;       // no SLAM rows in the last build: nothing was copied, the kernel wrote the cache
  bool flags_cached;       // h_flag_* hold the gate results of the last build (fetched with the update's status)
  int *h_flag_i;
  double *h_flag_d;
  char *trk_slot;          // xk_stage_tracks_begin .. _end: the staging slot being filled
  int trk_slot_K, trk_slot_nobs;
  bool async_pending;      // xk_build_compress_async ran: xk_apply_update owns the retry if the single-launch CAQR gave up
  // host pinned staging
  double *h_pin;
  size_t h_pin_doubles;
  int *h_pin_i;
  char err[256];
};

static int fail(xk_handle *h, int code, const char *what, hipError_t e = hipSuccess) {
  if (h) {
    if (e != hipSuccess) snprintf(h->err, sizeof(h->err), "%s: %s", what, hipGetErrorString(e));
    else snprintf(h->err, sizeof(h->err), "%s", what);
  }
  return code;
}
#define HIPCHK(h, call)                                                   \
  do {                                                                    \
    hipError_t e_ = (call);                                               \
    if (e_ != hipSuccess) return fail((h), XK_EDEVICE, #call, e_);        \
  } while (0)

static inline int round_up(int x, int m) { return (x + m - 1) / m * m; }

extern "C" const char *xk_strerror(int s) {
  switch (s) {
    case XK_OK: return "ok";
    case XK_EINVAL: return "invalid argument";
    case XK_ESINGULAR: return "innovation covariance not positive definite";
    case XK_ENAN: return "non-finite value";
    case XK_EDEVICE: return "HIP runtime error";
    case XK_ENOMEM: return "out of memory";
    case XK_ECAPACITY: return "problem exceeds handle capacity";
  }
  return "unknown status";
}
extern "C" const char *xk_last_error(const xk_handle *h) { return h ? h->err : "null handle"; }
extern "C" int xk_version(void) { return XK_VERSION_NUM; }
extern "C" void *xk_stream(xk_handle *h) { return h ? (void *)h->stream : nullptr; }

template <typename T>
static hipError_t dalloc(T **p, size_t count) {
  return hipMalloc((void **)p, sizeof(T) * (count ? count : 1));
}

static int create_impl(int device, int n_poses_max, int n_feat_max, int k_max, xk_handle **out);
extern "C" int xk_destroy(xk_handle *h);
// (a failure part-way through releases everything allocated so far)
// Experiment switches.  The RELEASE library (libxk.so) never looks at the environment: every switch has its default.  The LAB
// build (-DXK_LAB: x_multi_agent_amd/lab/libxk.so, include/xk_lab.h) reads them -- once each -- and carries the test hooks,
// the debug exports and the probe kernels the tests and tools/exp use.
#ifdef XK_LAB
static int env_int(const char *name, int dflt) {
  const char *v = getenv(name);
  return v ? atoi(v) : dflt;
}
#else
static inline int env_int(const char *, int dflt) { return dflt; }
#endif

extern "C" int xk_create(int device, int n_poses_max, int n_feat_max, int k_max, xk_handle **out) {
  if (!out) return XK_EINVAL;
  *out = nullptr;
  xk_handle *h = nullptr;
  const int rc = create_impl(device, n_poses_max, n_feat_max, k_max, &h);
  if (rc != XK_OK) { if (h) xk_destroy(h); return rc; }
  *out = h;
  return XK_OK;
}

static int create_impl(int device, int n_poses_max, int n_feat_max, int k_max, xk_handle **out) {
  if ( n_poses_max < 2 || n_poses_max > 64 || n_feat_max < 0 || k_max < 0) return XK_EINVAL;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return XK_EDEVICE;
  if (device < 0 || device >= ndev) return XK_EINVAL;
  xk_handle *h = (xk_handle *)calloc(1, sizeof(xk_handle));
  if (!h) return XK_ENOMEM;
  *out = h;   // the caller destroys it if anything below fails
  h->device = device;
  h->N = n_poses_max;
  h->Mmax = n_feat_max;
  h->Kmax = k_max;
  h->n = XK_CORE + 6 * n_poses_max + 3 * n_feat_max;
  h->na = h->n - XK_CORE;
  h->C1 = h->na + 1;
  h->C1P = round_up(h->C1, 64);
  if (h->C1P > 512) return XK_ECAPACITY;
  const int dmax = 2 * n_poses_max - 3;
  h->DB = dmax <= 64 ? 64 : 128;   // rows per tile slot (one track per tile; SLAM rows are packed DB per tile)
  const int slam_tiles = (2 * n_feat_max + h->DB - 1) / h->DB;
  h->ntiles_max = k_max + n_feat_max + slam_tiles;   // MSCKF tracks, MSCKF-SLAM tracks, packed SLAM rows
  h->CM = round_up(h->n + 1, 16);
  h->LDA = h->CM + round_up(h->n + 1, 16);
  HIPCHK(h, hipSetDevice(device));
  HIPCHK(h, hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
  HIPCHK(h, hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking));
  HIPCHK(h, hipEventCreateWithFlags(&h->ev_flags, hipEventDisableTiming));
  HIPCHK(h, hipEventCreateWithFlags(&h->ev_flags_done, hipEventDisableTiming));
  for (auto &e : h->ev) HIPCHK(h, hipEventCreate(&e));
  const size_t nn = (size_t)h->n * h->n;
  h->obs_cap = (size_t)k_max * n_poses_max;
  // window lists in one allocation, observations + track offsets in another: a staging call is ONE host-to-device copy
  // (small copies run as copy kernels of ~5 us each on the update's critical path)
  HIPCHK(h, dalloc(&h->d_q, 7 * (size_t)n_poses_max));
  HIPCHK(h, hipMalloc((void **)&h->d_done_cnt, sizeof(unsigned)));
  HIPCHK(h, hipMemset(h->d_done_cnt, 0, sizeof(unsigned)));
  h->h_win = (double *)calloc(7 * (size_t)n_poses_max, sizeof(double));
  if (!h->h_win) return fail(h, XK_EDEVICE, "host allocation");
  h->d_p = h->d_q + 4 * (size_t)n_poses_max;
  HIPCHK(h, dalloc(&h->d_obs, 2 * h->obs_cap + ((size_t)k_max + 2) / 2 + 1));
  h->d_trk_off = (int *)(h->d_obs + 2 * h->obs_cap);
  HIPCHK(h, dalloc(&h->d_feat, 3 * (size_t)n_feat_max));
  HIPCHK(h, dalloc(&h->d_zlast, 2 * (size_t)n_feat_max));
  HIPCHK(h, dalloc(&h->d_anchor, (size_t)n_feat_max));
  HIPCHK(h, dalloc(&h->d_tsz, (size_t)n_feat_max));
  HIPCHK(h, dalloc(&h->d_P, nn));
  HIPCHK(h, dalloc(&h->d_Pout, nn));
  HIPCHK(h, dalloc(&h->d_fq, (size_t)450));
  HIPCHK(h, dalloc(&h->d_chi95, (size_t)XK_CHI2_LEN));
  HIPCHK(h, dalloc(&h->d_chi90, (size_t)XK_CHI2_LEN));
  HIPCHK(h, hipMemcpy(h->d_chi95, XK_CHI2_095, sizeof(double) * XK_CHI2_LEN, hipMemcpyHostToDevice));
  HIPCHK(h, hipMemcpy(h->d_chi90, XK_CHI2_090, sizeof(double) * XK_CHI2_LEN, hipMemcpyHostToDevice));
  // (+ 256 rows behind the slots: where the first of two tail launches leaves its R for the second one, XkCaqrPipeArgs::extra_row0 --
  //  zero outside the trapezoid the launch writes, like d_R)
  HIPCHK(h, dalloc(&h->d_A, ((size_t)h->ntiles_max * h->DB + 256) * h->C1P));
  HIPCHK(h, hipMemset(h->d_A + (size_t)h->ntiles_max * h->DB * h->C1P, 0, sizeof(double) * 256 * h->C1P));
  h->hc_stride = xk_hc_stride(h->DB, h->C1P);
  h->opt_hlite = env_int("XK_HLITE", 1);
  h->rows_compact = false;
  HIPCHK(h, dalloc(&h->d_Hc, (size_t)std::max(k_max, 1) * h->hc_stride));
  HIPCHK(h, dalloc(&h->d_tile_rows, (size_t)h->ntiles_max));
  for (auto &pp : h->d_panel) HIPCHK(h, dalloc(&pp, (size_t)(h->ntiles_max + 10) * 256));
  HIPCHK(h, dalloc(&h->d_inl, (size_t)k_max));
  HIPCHK(h, dalloc(&h->d_inl_s, (size_t)n_feat_max));
  HIPCHK(h, dalloc(&h->d_gn, (size_t)k_max));
  HIPCHK(h, dalloc(&h->d_gam, (size_t)k_max));
  HIPCHK(h, dalloc(&h->d_gam_s, (size_t)n_feat_max));
  HIPCHK(h, dalloc(&h->d_gpf, 3 * (size_t)k_max));
  HIPCHK(h, dalloc(&h->d_R, (size_t)h->C1P * h->C1P));
  HIPCHK(h, hipMemset(h->d_R, 0, sizeof(double) * (size_t)h->C1P * h->C1P));
  // (d_R2: the measurement systems that are NOT the upper-triangular R of the whole stack -- split compression, uncompressed stacks;
  //  sized for C1P rows, and CM <= C1P + 16 rows of an uncompressed stack fit because its rows are at most n)
  HIPCHK(h, dalloc(&h->d_R2, (size_t)(h->C1P + 32) * h->C1P));
  HIPCHK(h, hipMemset(h->d_R2, 0, sizeof(double) * (size_t)(h->C1P + 32) * h->C1P));
  {
    hipDeviceProp_t prop;
    HIPCHK(h, hipGetDeviceProperties(&prop, device));
    h->n_cu = prop.multiProcessorCount;
    // The single-launch schedules need all 256 workgroups co-resident, one per CU.  Decide what can be decided up front:
    // the device must expose 256 CUs to this process (SPX mode, no CU mask visible in the properties) and the runtime must
    // agree that a 768-thread workgroup of each kernel fits a CU; what cannot be known here (another process on the GPU,
    // a CU mask set behind the runtime's back) is caught by the placement census and the bounded spins inside the launch.
    h->persist_ok = h->DB == 64 && h->C1 <= XkPipeWide::COLS && h->n_cu == 256;
    if (h->persist_ok) {
      int nb1 = 0;
      const void *kfn = h->C1 <= XkPipeNarrow::COLS ? (const void *)xk_caqr_pipe<XkPipeNarrow> : (const void *)xk_caqr_pipe<XkPipeWide>;
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb1, kfn, XK_PIPE_THREADS, 0) != hipSuccess) nb1 = 0;
      h->persist_ok = nb1 >= 1;
      if (h->persist_ok && h->C1 <= XkPipeNarrow::COLS) {
        int nb2 = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb2, (const void *)xk_caqr_pipe<XkPipeNarrow2>, XK_PIPE_THREADS, 0) != hipSuccess) nb2 = 0;
        if (nb2 < 1) h->opt_split = -1;      // (that geometry's kernel does not fit a CU: never taken)
      }
      (void)hipGetLastError();
    }
    h->fast_capable = h->persist_ok;
    h->tail_capable = h->DB == 128 && h->n_cu == 256 && h->C1 > 32;
    if (h->tail_capable) {
      int nbt = 0, nbt4 = 0;
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nbt, (const void *)xk_caqr_pipe<XkPipeTail>, XK_PIPE_THREADS, 0) != hipSuccess) nbt = 0;
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nbt4, (const void *)xk_caqr_pipe<XkPipeTail4>, XK_PIPE_THREADS, 0) != hipSuccess) nbt4 = 0;
      h->tail_capable = nbt >= 1 && nbt4 >= 1;
      (void)hipGetLastError();
    }
    h->tail_ok = h->tail_capable;
    h->opt_tail = env_int("XK_CAQR_TAIL", 1);
    h->opt_slam_split = env_int("XK_SLAM_SPLIT", 1);
    h->opt_pipe_min_rows = env_int("XK_PIPE_MIN_ROWS", 1);   // (512 until round 6: smaller stacks went to the multi-launch schedule -- 23 launches, 0.45 ms against 0.32)
    h->rearm_after = env_int("XK_CAQR_REARM", 64);
    h->opt_resident = env_int("XK_CAQR_RESIDENT", 1);
    h->opt_poison = env_int("XK_CAQR_RESIDENT_POISON", 0);
    h->opt_test_stall = env_int("XK_CAQR_TEST_STALL", 0);
    h->opt_tall26 = env_int("XK_CAQR_TALL26", 1);
    h->opt_kalman = env_int("XK_PIPE_KALMAN", 1);
    if (h->opt_split >= 0) h->opt_split = env_int("XK_PIPE_SPLIT", 1);   // (-1: the 152-tile kernel does not fit a CU on this device -- stays off)
    if (!h->fast_capable && h->DB == 64 && h->C1 <= XkPipeWide::COLS) {
      // Say so once, where an operator sees it: every update of this handle takes the multi-launch schedule (~1.6x slower).
      snprintf(h->err, sizeof(h->err), "single-launch CAQR unavailable on device %d: %s; the multi-launch schedule serves every update",
               device, h->n_cu != 256 ? "the process does not see 256 compute units (partition mode or CU mask)"
                                      : "a 768-thread workgroup of the kernel does not fit a compute unit");
      static const int quiet = env_int("XK_QUIET", 0);
      if (!quiet) fprintf(stderr, "xk: %s (n_cu = %d)\n", h->err, h->n_cu);
    }
    if (h->persist_ok || h->tail_capable) {
      // cross-XCD slabs of the single launch, TWO sets (a launch works on one and re-arms the other for its successor with the
      // NOT-YET pattern of the data-polled hand-offs, xk_xcd_sync.hip.h): per set X1 | X2 ([panels][16 strips][16 x C1P]) | X1P
      // (the tail launch of a tall system runs the last XkPipeTail::COLS / 16 panels only)
      const size_t np = h->persist_ok ? (size_t)(h->C1 + 15) / 16 : (size_t)XkPipeTail4::COLS / 16, strips = np * XK_PIPE_RLS;
      h->xslab_doubles = strips * 16 * h->C1P * 2 + strips * 256;
      HIPCHK(h, dalloc(&h->d_x1, 2 * h->xslab_doubles));
      HIPCHK(h, hipMemsetD32Async((hipDeviceptr_t)h->d_x1, (int)(XK_NOTYET_BITS & 0xffffffffu), 2 * 2 * h->xslab_doubles, h->stream));
      HIPCHK(h, dalloc(&h->d_rs, (size_t)8 * XK_PIPE_NT_MAX * 16 * h->C1P));
      HIPCHK(h, dalloc(&h->d_rpb, (size_t)8 * XK_PIPE_NT_MAX * 256));
      HIPCHK(h, dalloc(&h->d_xsync, (size_t)2 * XP_WORDS * 16));
      HIPCHK(h, hipMemset(h->d_xsync, 0, sizeof(unsigned) * 2 * XP_WORDS * 16));
      h->xsync_phase = 0;
#ifdef XK_LAB
      HIPCHK(h, dalloc(&h->d_pdbg, (size_t)XK_PDBG_WORDS));
      HIPCHK(h, hipMemset(h->d_pdbg, 0, sizeof(long long) * XK_PDBG_WORDS));
#endif
    }
  }
  HIPCHK(h, dalloc(&h->d_Maug, (size_t)h->CM * h->LDA));
  HIPCHK(h, dalloc(&h->d_X, (size_t)h->CM * h->LDA));
  HIPCHK(h, dalloc(&h->d_corr, (size_t)h->n + 4));   // + the status words right behind it: one copy brings both back
  HIPCHK(h, dalloc(&h->d_ct, (size_t)h->n));
  HIPCHK(h, dalloc(&h->d_tmpH, (size_t)h->CM * h->n));
  HIPCHK(h, dalloc(&h->d_tmpS, (size_t)h->CM * h->CM));
  HIPCHK(h, dalloc(&h->d_tmpP, nn));
  HIPCHK(h, dalloc(&h->d_rdiag, (size_t)h->CM));
  HIPCHK(h, dalloc(&h->d_tmpz, (size_t)h->CM));
  // The status words and the correction of the resident path (xk_apply_update) are written by the kernels straight into
  // pinned host memory through its device-visible address: after the stream synchronisation that ends an update the host
  // reads them in place -- no device-to-host copy (a ~4 us blit kernel plus its launch gap at the end of every frame).
  HIPCHK(h, hipHostMalloc((void **)&h->h_out, sizeof(double) * ((size_t)h->n + 4), hipHostMallocDefault));
  memset(h->h_out, 0, sizeof(double) * ((size_t)h->n + 4));
  h->d_status = (int *)(h->h_out + h->n);
  HIPCHK(h, dalloc(&h->d_payload, (size_t)xk_payload_doubles(n_poses_max, n_feat_max)));
  HIPCHK(h, dalloc(&h->d_ci, (size_t)4 * nn + 64 * (size_t)h->n + 1024));
  h->h_pin_doubles = nn + 8 * (size_t)h->n + 4 * (size_t)k_max + 4 * (size_t)n_feat_max + 1024;
  HIPCHK(h, hipHostMalloc((void **)&h->h_pin, sizeof(double) * h->h_pin_doubles));
  h->csr_cap = 24 * (size_t)h->n + 3 * (size_t)n_feat_max * h->n;
  {
    const size_t m = (size_t)std::max(n_feat_max, 1);
    HIPCHK(h, dalloc(&h->d_trk2_off, m + 1));
    HIPCHK(h, dalloc(&h->d_inl2, m));
    HIPCHK(h, dalloc(&h->d_gn2, m));
    HIPCHK(h, dalloc(&h->d_obs2, 2 * m * n_poses_max));
    HIPCHK(h, dalloc(&h->d_gpf2, 3 * m));
    HIPCHK(h, dalloc(&h->d_W2, m * h->DB * h->na));
    HIPCHK(h, dalloc(&h->d_gam2, m));
    HIPCHK(h, dalloc(&h->d_H1, 3 * m * h->n));
    HIPCHK(h, dalloc(&h->d_H2, 9 * m));
    HIPCHK(h, dalloc(&h->d_r1, 3 * m));
    HIPCHK(h, dalloc(&h->d_feat2, 3 * m));
    h->h_trk2_off = (int *)calloc(m + 1, sizeof(int));
    if (!h->h_trk2_off) return fail(h, XK_ENOMEM, "host track offsets");
  }
  HIPCHK(h, dalloc(&h->d_csr_v, h->csr_cap + XK_CORE * XK_CORE + 9 * (size_t)n_feat_max * n_feat_max + 7 * (size_t)n_poses_max + ((size_t)h->n + 2 + h->csr_cap) / 2 + 1));
  h->d_csr_i = nullptr;   // (the integer part follows the values of each operand)
  h->h_trk_off = (int *)calloc((size_t)k_max + 1, sizeof(int));
  if (!h->h_trk_off) return fail(h, XK_ENOMEM, "host track offsets");
  HIPCHK(h, hipHostMalloc((void **)&h->h_pin_i, sizeof(int) * ((size_t)k_max + n_feat_max + 512)));
  h->stage_bytes = std::max({sizeof(double) * 2 * h->obs_cap + sizeof(int) * ((size_t)k_max + 1), sizeof(double) * 7 * (size_t)n_poses_max,
                             (sizeof(int) + sizeof(double)) * h->csr_cap + sizeof(int) * ((size_t)h->n + 1) + sizeof(double) * (XK_CORE * XK_CORE + 9 * (size_t)n_feat_max * n_feat_max + 7 * (size_t)n_poses_max),
                             sizeof(double) * 8 * (size_t)std::max(n_feat_max, 1)}) + 256;
  for (auto &sp : h->h_stage) HIPCHK(h, hipHostMalloc((void **)&sp, h->stage_bytes));

  HIPCHK(h, hipHostMalloc((void **)&h->h_flag_i, sizeof(int) * ((size_t)k_max + n_feat_max + 8)));
  HIPCHK(h, hipHostMalloc((void **)&h->h_flag_d, sizeof(double) * ((size_t)k_max + n_feat_max + 8)));
  memset(h->d_status, 0, sizeof(int) * 4);
  HIPCHK(h, hipMemset(h->d_tile_rows, 0, sizeof(int) * (size_t)h->ntiles_max));
  h->sigma_img = 0.0;
  *out = h;
  return XK_OK;
}

extern "C" int xk_destroy(xk_handle *h) {
  if (!h) return XK_OK;
  hipSetDevice(h->device);
  hipStreamSynchronize(h->stream);
  void *ptrs[] = {h->d_q, h->d_obs, h->d_feat, h->d_zlast, h->d_anchor, h->d_tsz,
                  h->d_P, h->d_Pout, h->d_chi95, h->d_chi90, h->d_A, h->d_tile_rows, h->d_panel[0], h->d_panel[1], h->d_inl, h->d_inl_s,
                  h->d_gn, h->d_gam, h->d_gam_s, h->d_gpf, h->d_R, h->d_Maug, h->d_X, h->d_corr,
                  h->d_ct, h->d_tmpH, h->d_tmpS, h->d_tmpP, h->d_rdiag, h->d_tmpz, h->d_payload,
                  h->d_ci};
  for (void *p : ptrs)
    if (p) hipFree(p);
  for (void *p2 : {(void *)h->d_trk2_off, (void *)h->d_inl2, (void *)h->d_gn2, (void *)h->d_obs2, (void *)h->d_gpf2, (void *)h->d_W2,
                   (void *)h->d_gam2, (void *)h->d_H1, (void *)h->d_H2, (void *)h->d_r1, (void *)h->d_feat2})
    if (p2) hipFree(p2);
  free(h->h_trk2_off);
  if (h->d_csr_v) hipFree(h->d_csr_v);
  if (h->d_Hc) hipFree(h->d_Hc);
  if (h->d_R2) hipFree(h->d_R2);
  delete h->fused_ct;
  if (h->d_Psnap) hipFree(h->d_Psnap);
  if (h->d_Psnap2) hipFree(h->d_Psnap2);
  if (h->d_fq) hipFree(h->d_fq);
  for (void *p4 : {(void *)h->d_rs, (void *)h->d_rpb, (void *)h->d_xsync})
    if (p4) hipFree(p4);
  for (void *p3 : {(void *)h->d_x1, (void *)h->d_pdbg})
    if (p3) hipFree(p3);
  if (h->d_ciws) hipFree(h->d_ciws);
  if (h->d_batch) hipFree(h->d_batch);
  if (h->h_batch) hipHostFree(h->h_batch);
  if (h->h_ci_cols) hipHostFree(h->h_ci_cols);
  if (h->h_ci_w) hipHostFree(h->h_ci_w);
  free(h->h_trk_off);
  if (h->h_out) hipHostFree(h->h_out);
  free(h->h_win);
  if (h->d_done_cnt) hipFree(h->d_done_cnt);
  if (h->h_pin) hipHostFree(h->h_pin);
  if (h->h_pin_i) hipHostFree(h->h_pin_i);
  for (auto &sp : h->h_stage)
    if (sp) hipHostFree(sp);

  if (h->h_flag_i) hipHostFree(h->h_flag_i);
  if (h->h_flag_d) hipHostFree(h->h_flag_d);
  for (auto &e : h->ev)
    if (e) hipEventDestroy(e);
  if (h->copy_stream) { hipStreamSynchronize(h->copy_stream); hipStreamDestroy(h->copy_stream); }
  for (int j = 1; j < 8; ++j) {
    if (h->ci_stream[j]) { hipStreamSynchronize(h->ci_stream[j]); hipStreamDestroy(h->ci_stream[j]); }
    if (h->ci_join[j]) hipEventDestroy(h->ci_join[j]);
  }
  if (h->ci_fork) hipEventDestroy(h->ci_fork);
  if (h->ev_flags) hipEventDestroy(h->ev_flags);
  if (h->ev_flags_done) hipEventDestroy(h->ev_flags_done);
  if (h->stream) hipStreamDestroy(h->stream);
  free(h);
  return XK_OK;
}

// ---------------------------------------------------------------------------
// staging
// ---------------------------------------------------------------------------
// next slot of the pinned ring: host inputs are copied there and go to the device with an asynchronous copy, so that
// staging never waits for the device (the caller's buffers are free on return, as before)
// A slot is reused XK_STAGE_SLOTS staging calls later.  Nothing in between need have synchronised the stream (a loop of
// xk_cov_congruence, repeated re-staging), and the copy that reads a slot may still be queued.  The handle counts the slots
// handed out since it last KNEW its stream to be idle (every update ends with a synchronisation: xk_apply_update,
// read_status); when the ring is about to wrap without one, it synchronises itself.  In a filter loop that never happens --
// a frame uses five or six slots -- so staging costs no event and no wait there.
static void stage_stream_idle(xk_handle *h) { h->stage_since_sync = 0; }
static char *stage_slot(xk_handle *h, size_t bytes) {
  if (bytes > h->stage_bytes) return nullptr;
  if (++h->stage_since_sync >= XK_STAGE_SLOTS) {
    // about to wrap onto a slot whose copy may still be queued: wait; if the wait itself fails the slot is not handed out
    if (hipStreamSynchronize(h->stream) != hipSuccess) { --h->stage_since_sync; return nullptr; }
    h->stage_since_sync = 1;
  }
  const int s = h->stage_next;
  h->stage_next = (s + 1) % XK_STAGE_SLOTS;
  return h->h_stage[s];
}

// Staged window lists that no kernel has carried to the device yet: one host-to-device copy through the pinned ring.
static int flush_window(xk_handle *h) {
  if (!h->win_pending) return XK_OK;
  const size_t bytes = sizeof(double) * 7 * (size_t)h->n_poses;
  double *st = (double *)stage_slot(h, bytes);
  if (!st) return fail(h, XK_ECAPACITY, "staging slot too small");
  memcpy(st, h->h_win, bytes);
  HIPCHK(h, hipMemcpyAsync(h->d_q, st, bytes, hipMemcpyHostToDevice, h->stream));
  h->win_pending = false;
  return XK_OK;
}

extern "C" int xk_stage_window(xk_handle *h, const double *C_q_G, const double *G_p_C, int n_poses) {
  if (!h || !C_q_G || !G_p_C) return XK_EINVAL;
  if (n_poses < 2 || n_poses > h->N) return fail(h, XK_ECAPACITY, "n_poses outside [2, n_poses_max]");
  HIPCHK(h, hipSetDevice(h->device));
  // Kept on the host until the next consumer.  In a filter frame that is the congruence launch of manage(), whose operand copy
  // carries the lists along (congruence()): a copy of their own runs as a ~5 us copy kernel on the frame's critical path.
  // Everything else gets them by flush_window.  Staging the same lists again (manage() and constructUpdate both do) is free.
  const bool same = h->win_valid && n_poses == h->n_poses && memcmp(h->h_win, C_q_G, sizeof(double) * 4 * n_poses) == 0 &&
                    memcmp(h->h_win + 4 * n_poses, G_p_C, sizeof(double) * 3 * n_poses) == 0;
  if (!same) {
    memcpy(h->h_win, C_q_G, sizeof(double) * 4 * n_poses);
    memcpy(h->h_win + 4 * n_poses, G_p_C, sizeof(double) * 3 * n_poses);
    h->d_p = h->d_q + 4 * (size_t)n_poses;                             // positions right behind the attitudes in use
    h->win_pending = true;
    h->win_valid = true;
    h->n_poses = n_poses;
  }
  h->have_rows = h->have_R = false;
  h->ms_built = false;
  return XK_OK;
}

// Track staging in two halves, so that a host that builds its CSR lists anyway can build them IN the pinned staging
// memory (one pass over the tracker's lists instead of list -> vector -> staging copy; 0.2 MB at the headline size):
//   xk_stage_tracks_begin(K, n_obs, &off, &obs)  ->  fill off[0..K], obs[0..2 n_obs)  ->  xk_stage_tracks_end()
extern "C" int xk_stage_tracks_begin(xk_handle *h, int K, int n_obs, int **trk_off, double **obs_xy) {
  if (!h || K < 0 || n_obs < 0 || !trk_off || !obs_xy) return XK_EINVAL;
  if (K > h->Kmax) return fail(h, XK_ECAPACITY, "K > k_max");
  if ((size_t)n_obs > h->obs_cap) return fail(h, XK_ECAPACITY, "too many observations");
  const size_t ob = sizeof(double) * 2 * (size_t)n_obs;
  char *st = stage_slot(h, ob + sizeof(int) * (K + 1));
  if (!st) return fail(h, XK_ECAPACITY, "staging slot too small");
  h->trk_slot = st; h->trk_slot_K = K; h->trk_slot_nobs = n_obs;
  *obs_xy = (double *)st;
  *trk_off = (int *)(st + ob);
  return XK_OK;
}

extern "C" int xk_stage_tracks_end(xk_handle *h) {
  if (!h || !h->trk_slot) return XK_EINVAL;
  const int K = h->trk_slot_K;
  const size_t ob = sizeof(double) * 2 * (size_t)h->trk_slot_nobs;
  const int *trk_off = (const int *)(h->trk_slot + ob);
  char *st = h->trk_slot;
  h->trk_slot = nullptr;
  int lmax = 0;
  if (K > 0) {
    if (trk_off[0] != 0) return fail(h, XK_EINVAL, "trk_off[0] != 0");
    for (int k = 0; k < K; ++k) {
      const int L = trk_off[k + 1] - trk_off[k];
      if (L < 2 || L > h->N) return fail(h, XK_EINVAL, "track length outside [2, n_poses_max]");
      lmax = std::max(lmax, L);
    }
    if (trk_off[K] != h->trk_slot_nobs) return fail(h, XK_EINVAL, "trk_off[K] != n_obs");
    HIPCHK(h, hipSetDevice(h->device));
    h->d_trk_off = (int *)(h->d_obs + 2 * (size_t)trk_off[K]);         // offsets right behind the observations in use
    HIPCHK(h, hipMemcpyAsync(h->d_obs, st, ob + sizeof(int) * (K + 1), hipMemcpyHostToDevice, h->stream));
    memcpy(h->h_trk_off, trk_off, sizeof(int) * (K + 1));
  }
  h->K = K;
  h->have_rows = h->have_R = false;
  h->h_pin_i[0] = lmax;                                                 // the longest track, validated against n_poses at build time
  return XK_OK;
}

extern "C" int xk_stage_tracks(xk_handle *h, const int *trk_off, const double *obs_xy, int K) {
  if (!h || K < 0 || (K > 0 && (!trk_off || !obs_xy))) return XK_EINVAL;
  if (K > 0 && trk_off[0] != 0) return fail(h, XK_EINVAL, "trk_off[0] != 0");
  if (K > 0 && trk_off[K] < 0) return fail(h, XK_EINVAL, "negative observation count");
  int *so = nullptr;
  double *sx = nullptr;
  const int rc = xk_stage_tracks_begin(h, K, K > 0 ? trk_off[K] : 0, &so, &sx);
  if (rc != XK_OK) return rc;
  if (K > 0) {
    memcpy(sx, obs_xy, sizeof(double) * 2 * (size_t)trk_off[K]);
    memcpy(so, trk_off, sizeof(int) * (K + 1));
  } else {
    so[0] = 0;
  }
  return xk_stage_tracks_end(h);
}

extern "C" int xk_stage_slam(xk_handle *h, const double *feat, const int *anchor_idxs, const int *track_sizes,
                             const double *z_last, int M) {
  if (!h || M < 0 || (M > 0 && (!feat || !anchor_idxs || !track_sizes || !z_last))) return XK_EINVAL;
  if (M > h->Mmax) return fail(h, XK_ECAPACITY, "M > n_feat_max");
  h->anchor_max = -1;
  for (int j = 0; j < M; ++j) {   // a stale anchor (-1 in StateManager's unused slots) or size would index the window / chi-square table out of bounds
    if (anchor_idxs[j] < 0 || anchor_idxs[j] >= h->N) return fail(h, XK_EINVAL, "SLAM anchor index outside [0, n_poses_max)");
    if (track_sizes[j] < 1) return fail(h, XK_EINVAL, "SLAM track size < 1");
    h->anchor_max = std::max(h->anchor_max, anchor_idxs[j]);
  }
  if (M > 0) {
    HIPCHK(h, hipSetDevice(h->device));
    char *st = stage_slot(h, sizeof(double) * 6 * M);
    if (!st) return fail(h, XK_ECAPACITY, "staging slot too small");
    double *sd = (double *)st;
    int *si = (int *)(sd + 5 * M);
    memcpy(sd, feat, sizeof(double) * 3 * M);
    memcpy(sd + 3 * M, z_last, sizeof(double) * 2 * M);
    memcpy(si, anchor_idxs, sizeof(int) * M);
    memcpy(si + M, track_sizes, sizeof(int) * M);
    HIPCHK(h, hipMemcpyAsync(h->d_feat, sd, sizeof(double) * 3 * M, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->d_zlast, sd + 3 * M, sizeof(double) * 2 * M, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->d_anchor, si, sizeof(int) * M, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->d_tsz, si + M, sizeof(int) * M, hipMemcpyHostToDevice, h->stream));
  }
  h->M = M;
  h->have_rows = h->have_R = false;
  return XK_OK;
}

// MSCKF-SLAM tracks: the tracks whose feature becomes a persistent (SLAM) feature this frame
// (VioUpdater::constructUpdate, vio_updater.cpp:311-321).
extern "C" int xk_stage_msckf_slam(xk_handle *h, const int *trk_off, const double *obs_xy, int K2) {
  if (!h || K2 < 0 || (K2 > 0 && (!trk_off || !obs_xy))) return XK_EINVAL;
  if (K2 > h->Mmax) return fail(h, XK_ECAPACITY, "more MSCKF-SLAM tracks than feature slots");
  int lmax = 0;
  if (K2 > 0) {
    if (trk_off[0] != 0) return fail(h, XK_EINVAL, "trk_off[0] != 0");
    for (int k = 0; k < K2; ++k) {
      const int L = trk_off[k + 1] - trk_off[k];
      if (L < 2 || L > h->N) return fail(h, XK_EINVAL, "track length outside [2, n_poses_max]");
      lmax = std::max(lmax, L);
    }
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipMemcpyAsync(h->d_trk2_off, trk_off, sizeof(int) * (K2 + 1), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->d_obs2, obs_xy, sizeof(double) * 2 * trk_off[K2], hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    memcpy(h->h_trk2_off, trk_off, sizeof(int) * (K2 + 1));
  }
  h->K2 = K2;
  h->ms_built = false;
  h->h_pin_i[1] = lmax;
  h->have_rows = h->have_R = false;
  return XK_OK;
}

extern "C" int xk_msckf_slam_results(xk_handle *h, int *inlier, double *gamma, double *H1, int ldh1, double *H2, int ldh2,
                                     double *r1, double *features) {
  if (!h) return XK_EINVAL;
  const int k = h->K2, n = h->n;
  if (k == 0) return XK_OK;
  if (!h->ms_built) return fail(h, XK_EINVAL, "xk_msckf_build has not run on the staged MSCKF-SLAM tracks");
  if ((H1 && ldh1 < 3 * k) || (H2 && ldh2 < 3 * k)) return XK_EINVAL;
  HIPCHK(h, hipSetDevice(h->device));
  std::vector<double> h1((size_t)3 * k * n), h2((size_t)9 * k);
  if (inlier) HIPCHK(h, hipMemcpyAsync(inlier, h->d_inl2, sizeof(int) * k, hipMemcpyDeviceToHost, h->stream));
  if (gamma) HIPCHK(h, hipMemcpyAsync(gamma, h->d_gam2, sizeof(double) * k, hipMemcpyDeviceToHost, h->stream));
  if (r1) HIPCHK(h, hipMemcpyAsync(r1, h->d_r1, sizeof(double) * 3 * k, hipMemcpyDeviceToHost, h->stream));
  if (features) HIPCHK(h, hipMemcpyAsync(features, h->d_feat2, sizeof(double) * 3 * k, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipMemcpyAsync(h1.data(), h->d_H1, sizeof(double) * h1.size(), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipMemcpyAsync(h2.data(), h->d_H2, sizeof(double) * h2.size(), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  if (H1)     // device: [track][3][n] row-major -> (3k x n) column-major
    for (int j = 0; j < k; ++j)
      for (int r = 0; r < 3; ++r)
        for (int c = 0; c < n; ++c) H1[(size_t)(3 * j + r) + (size_t)c * ldh1] = h1[((size_t)j * 3 + r) * n + c];
  if (H2) {   // block diagonal (msckf_slam_update.cpp:231)
    for (int c = 0; c < 3 * k; ++c)
      for (int r = 0; r < 3 * k; ++r) H2[r + (size_t)c * ldh2] = 0.0;
    for (int j = 0; j < k; ++j)
      for (int c = 0; c < 3; ++c)
        for (int r = 0; r < 3; ++r) H2[(size_t)(3 * j + r) + (size_t)(3 * j + c) * ldh2] = h2[9 * (size_t)j + r + 3 * c];
  }
  return XK_OK;
}

static bool inv3(const double *a /*col-major*/, double *o) {
  const double c00 = a[4] * a[8] - a[7] * a[5], c01 = a[7] * a[2] - a[1] * a[8], c02 = a[1] * a[5] - a[4] * a[2];
  const double det = a[0] * c00 + a[3] * c01 + a[6] * c02;
  if (!(fabs(det) > 0.0)) return false;
  const double id = 1.0 / det;
  o[0] = c00 * id; o[1] = c01 * id; o[2] = c02 * id;
  o[3] = (a[6] * a[5] - a[3] * a[8]) * id; o[4] = (a[0] * a[8] - a[6] * a[2]) * id; o[5] = (a[3] * a[2] - a[0] * a[5]) * id;
  o[6] = (a[3] * a[7] - a[6] * a[4]) * id; o[7] = (a[6] * a[1] - a[0] * a[7]) * id; o[8] = (a[0] * a[4] - a[3] * a[1]) * id;
  return true;
}

static int congruence(xk_handle *h, const int *row_ptr, const int *col_idx, const double *val, int nnz, const double *q,
                      int qdim, int qoff);

// StateManager::initMsckfSlamFeatures + addFeatureStates (state_manager.cpp:151-174,199-226) on the resident
// covariance: with G = H2^-1 H1 the new feature states are  f - G corr + H2^-1 r1,  their covariance blocks
// -G P (cross) and G P G^T + sigma^2 H2^-1 H2^-T (diagonal) -- one congruence with J = [I; -G] on the rows of
// the new features plus the noise block.
extern "C" int xk_init_msckf_slam_features(xk_handle *h, int n_features, const double *correction, double sigma_img,
                                           double *new_features) {
  if (!h || !correction || !new_features || n_features < 0 || !(sigma_img > 0.0)) return XK_EINVAL;
  const int k = h->K2, n = h->n;
  if (k == 0) return XK_OK;
  if (!h->ms_built) return fail(h, XK_EINVAL, "xk_msckf_build has not run on the staged MSCKF-SLAM tracks");
  if (n_features + k > h->Mmax) return fail(h, XK_ECAPACITY, "not enough free feature slots");
  HIPCHK(h, hipSetDevice(h->device));
  std::vector<double> h1((size_t)3 * k * n), h2((size_t)9 * k), r1((size_t)3 * k), f((size_t)3 * k);
  HIPCHK(h, hipMemcpyAsync(h1.data(), h->d_H1, sizeof(double) * h1.size(), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipMemcpyAsync(h2.data(), h->d_H2, sizeof(double) * h2.size(), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipMemcpyAsync(r1.data(), h->d_r1, sizeof(double) * r1.size(), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipMemcpyAsync(f.data(), h->d_feat2, sizeof(double) * f.size(), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  const int ns = XK_CORE + 6 * h->N + 3 * n_features;     // first row of the new features
  const double var_img = sigma_img * sigma_img;
  std::vector<double> G((size_t)3 * k * n, 0.0), Q((size_t)9 * k * k, 0.0);
  for (int j = 0; j < k; ++j) {
    double hi[9];
    if (!inv3(&h2[9 * (size_t)j], hi)) return fail(h, XK_ESINGULAR, "H2 is singular (camera hovering)");   // state_manager.cpp:158-160
    for (int r = 0; r < 3; ++r) {
      double *g = &G[((size_t)3 * j + r) * n];
      for (int c = 0; c < n; ++c)
        g[c] = hi[r] * h1[((size_t)j * 3) * n + c] + hi[r + 3] * h1[((size_t)j * 3 + 1) * n + c] + hi[r + 6] * h1[((size_t)j * 3 + 2) * n + c];
      double fr = f[3 * j + r] + hi[r] * r1[3 * j] + hi[r + 3] * r1[3 * j + 1] + hi[r + 6] * r1[3 * j + 2];
      for (int c = 0; c < n; ++c) fr -= g[c] * correction[c];
      new_features[3 * j + r] = fr;
      for (int c2 = 0; c2 < 3; ++c2)      // sigma^2 H2^-1 H2^-T, block (j, j)
        Q[(size_t)(3 * j + r) + (size_t)(3 * j + c2) * 3 * k] = var_img * (hi[r] * hi[c2] + hi[r + 3] * hi[c2 + 3] + hi[r + 6] * hi[c2 + 6]);
    }
  }
  std::vector<int> rp(n + 1), ci;
  std::vector<double> v;
  for (int r = 0; r < n; ++r) {
    rp[r] = (int)ci.size();
    if (r >= ns && r < ns + 3 * k) {
      const double *g = &G[(size_t)(r - ns) * n];
      for (int c = 0; c < n; ++c)
        if (g[c] != 0.0) { ci.push_back(c); v.push_back(-g[c]); }
    } else {
      ci.push_back(r);
      v.push_back(1.0);
    }
  }
  rp[n] = (int)ci.size();
  if (ci.size() > h->csr_cap) return fail(h, XK_ECAPACITY, "sparse operand too large");
  return congruence(h, rp.data(), ci.data(), v.data(), (int)ci.size(), Q.data(), 3 * k, ns);
}

// StateManager::initStandardSlamFeatures + addFeatureStates (state_manager.cpp:176-226): k new features with no
// correlation to the rest, variances sigma_img^2, sigma_img^2, sigma_rho_0^2.
extern "C" int xk_init_standard_slam_features(xk_handle *h, int n_features, int k, double sigma_img, double sigma_rho_0) {
  if (!h || n_features < 0 || k < 0) return XK_EINVAL;
  if (k == 0) return XK_OK;
  if (n_features + k > h->Mmax) return fail(h, XK_ECAPACITY, "not enough free feature slots");
  const int n = h->n, ns = XK_CORE + 6 * h->N + 3 * n_features;
  std::vector<int> rp(n + 1), ci;
  std::vector<double> v, Q((size_t)9 * k * k, 0.0);
  for (int r = 0; r < n; ++r) {
    rp[r] = (int)ci.size();
    if (r < ns || r >= ns + 3 * k) { ci.push_back(r); v.push_back(1.0); }
  }
  rp[n] = (int)ci.size();
  for (int j = 0; j < 3 * k; ++j) Q[(size_t)j + (size_t)j * 3 * k] = (j % 3 == 2) ? sigma_rho_0 * sigma_rho_0 : sigma_img * sigma_img;
  return congruence(h, rp.data(), ci.data(), v.data(), (int)ci.size(), Q.data(), 3 * k, ns);
}

extern "C" int xk_upload_P(xk_handle *h, const double *P, int ldp, int n) {
  if (!h || !P || n != h->n || ldp < n) return XK_EINVAL;
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipMemcpy2DAsync(h->d_P, sizeof(double) * n, P, sizeof(double) * ldp, sizeof(double) * n, n,
                             hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return XK_OK;
}

extern "C" int xk_download_P(xk_handle *h, double *P, int ldp, int n) {
  if (!h || !P || n != h->n || ldp < n) return XK_EINVAL;
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipMemcpy2DAsync(P, sizeof(double) * ldp, h->d_P, sizeof(double) * n, sizeof(double) * n, n,
                             hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return XK_OK;
}

// ---------------------------------------------------------------------------
// launch helpers (all asynchronous on h->stream)
// ---------------------------------------------------------------------------

static int split_plan(const xk_handle *h);
static long split_rows_nominal(const xk_handle *h);
static int launch_build(xk_handle *h, double sigma_img) {
  if (h->n_poses < 2) return fail(h, XK_EINVAL, "window not staged");
  if (h->K > 0 && h->h_pin_i[0] > h->n_poses) return fail(h, XK_EINVAL, "track longer than the staged window");
  const int slam_tiles = (2 * h->M + h->DB - 1) / h->DB;
  XkFeatArgs fa;
  memset(&fa, 0, sizeof(fa));
  size_t feat_lds = 0;
  { int rcw = flush_window(h); if (rcw != XK_OK) return rcw; }   // (normally carried by the frame's congruence launch already)
  if (h->K > 0) {
    XkFeatArgs &a = fa;
    a.q = h->d_q; a.p = h->d_p; a.n_poses = h->n_poses; a.n_poses_max = h->N;
    a.trk_off = h->d_trk_off; a.obs = h->d_obs; a.K = h->K;
    a.P = h->d_P; a.n = h->n; a.var_img = sigma_img * sigma_img; a.chi95 = h->d_chi95;
    a.A = h->d_A; a.DB = h->DB; a.C1P = h->C1P; a.na = h->na;
    // factor records instead of tiles: H0 never visits HBM.  Whoever compresses forms its rows from them -- the tile workgroups
    // of the single launch (narrow systems; next to the wide geometry's 80-row tiles and 2 lanes per column forming the entries
    // costs more than the tiles' trip through HBM -- config 2: 1546 -> 1521 updates/s -- so those keep their tiles), or the
    // first pass of the multi-launch schedule (128-row slots: always; 64-row slots: when the single launch was armed and then
    // not taken or gave up).
    h->rows_compact = h->opt_hlite && h->d_Hc && !h->feat_dbg && split_plan(h) != 3 &&      // (an uncompressed small stack is copied from tiles)
                      (h->DB == 128 || (h->opt_resident && h->persist_ok && (h->C1 <= XkPipeNarrow::COLS || h->opt_hlite >= 2)));   // (lab: 2 = the wide geometry too)
    a.Hc = h->rows_compact ? h->d_Hc : nullptr; a.hs = h->hc_stride; a.hcvr = xk_hc_vr(h->DB);
    a.tile_rows = h->d_tile_rows; a.inlier = h->d_inl; a.gamma = h->d_gam; a.gpf = h->d_gpf; a.gn_iters = h->d_gn;
    a.gpf_in = nullptr; a.up_out = nullptr; a.batch = nullptr; a.dbg = h->feat_dbg;
    a.inlier_h = h->h_flag_i; a.gamma_h = h->h_flag_d;     // gate results also straight into the pinned flag cache
    const bool packed = xk_feature_packed(h->n_poses);      // windows of more than 33 poses: gate matrix as a packed triangle
    feat_lds = xk_feature_lds_bytes(h->n_poses, packed);
    // (with SLAM features the tracks and the features share one launch, below)
    if (packed) hipLaunchKernelGGL(xk_msckf_feature_packed, dim3(h->K), dim3(XK_FEAT_THREADS), feat_lds, h->stream, a);
    else if (h->M == 0 || h->feat_dbg) hipLaunchKernelGGL(xk_msckf_feature, dim3(h->K), dim3(XK_FEAT_THREADS), feat_lds, h->stream, a);
  }
  if (h->K2 > 0) {   // tracks that become persistent features this frame: tiles K .. K + K2 - 1
    if (h->h_pin_i[1] > h->n_poses) return fail(h, XK_EINVAL, "MSCKF-SLAM track longer than the staged window");
    XkTriMultiArgs ta{h->d_q, h->d_p, h->d_obs2, 0, h->d_gpf2, h->d_gn2, h->d_trk2_off, h->n_poses};
    hipLaunchKernelGGL(xk_triangulate_multi, dim3(h->K2), dim3(64), 0, h->stream, ta);
    XkSlamInitArgs a;
    a.q = h->d_q; a.p = h->d_p; a.n_poses = h->n_poses; a.n_poses_max = h->N;
    a.trk_off = h->d_trk2_off; a.obs = h->d_obs2; a.gpf = h->d_gpf2;
    a.P = h->d_P; a.n = h->n; a.na = h->na; a.var_img = sigma_img * sigma_img; a.chi95 = h->d_chi95;
    a.A = h->d_A + (size_t)h->K * h->DB * h->C1P; a.DB = h->DB; a.C1P = h->C1P; a.W = h->d_W2;
    a.tile_rows = h->d_tile_rows + h->K; a.inlier = h->d_inl2; a.gamma = h->d_gam2;
    a.H1 = h->d_H1; a.H2 = h->d_H2; a.r1 = h->d_r1; a.features = h->d_feat2;
    const size_t lds = xk_slaminit_lds_bytes(h->n_poses);
    if (!h->attr_slaminit) {   // per handle = per device: a process may hold handles on several GPUs
      hipFuncSetAttribute((const void *)xk_msckf_slam_init, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512);
      h->attr_slaminit = true;
    }
    hipLaunchKernelGGL(xk_msckf_slam_init, dim3(h->K2), dim3(XK_FEAT_THREADS), lds, h->stream, a);
    h->ms_built = true;
  }
  if (h->M > 0) {
    if (h->anchor_max >= h->n_poses) return fail(h, XK_EINVAL, "SLAM anchor outside the staged window");
    XkSlamArgs s;
    s.q = h->d_q; s.p = h->d_p; s.n_poses = h->n_poses; s.n_poses_max = h->N;
    s.feat = h->d_feat; s.anchor_idxs = h->d_anchor; s.track_sizes = h->d_tsz; s.z_last = h->d_zlast; s.M = h->M;
    s.P = h->d_P; s.n = h->n; s.var_img = sigma_img * sigma_img; s.chi90 = h->d_chi90; s.chi_len = XK_CHI2_LEN;
    s.A = h->d_A + (size_t)(h->K + h->K2) * h->DB * h->C1P; s.DB = h->DB; s.C1P = h->C1P; s.na = h->na;
    s.inlier = h->d_inl_s; s.gamma = h->d_gam_s;
    if (h->K > 0 && !h->feat_dbg && !xk_feature_packed(h->n_poses)) hipLaunchKernelGGL(xk_build_rows, dim3(h->K + h->M), dim3(XK_FEAT_THREADS), feat_lds, h->stream, fa, s);
    else hipLaunchKernelGGL(xk_slam_rows, dim3(h->M), dim3(64), 0, h->stream, s);
    // rows per SLAM tile (gated-out features leave zero rows, as in the reference)
    std::vector<int> tr(slam_tiles);
    for (int t = 0; t < slam_tiles; ++t) tr[t] = std::min(h->DB, 2 * h->M - t * h->DB);
    for (int t = 0; t < slam_tiles; ++t) h->h_pin_i[8 + t] = tr[t];
    if (hipMemcpyAsync(h->d_tile_rows + h->K + h->K2, h->h_pin_i + 8, sizeof(int) * slam_tiles, hipMemcpyHostToDevice,
                       h->stream) != hipSuccess)
      return fail(h, XK_EDEVICE, "tile_rows upload");
  }
  h->sigma_img = sigma_img;
  h->have_rows = true;
  h->have_R = false;
  h->compress_deferred = false;
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(h, XK_EDEVICE, "build launch", e);
  return XK_OK;
}

// The stack of an update that is NOT compressed (rows <= columns: vio_updater.cpp:487 compresses only `if (h.rows() > h.cols())`): slot t's rows
// -- the tile the per-feature kernel left -- go to rows [off_t, off_t + 2 L_t - 3) of T, off_t = 2 trk_off[t] - 3 t (every track counted in:
// the host queues the update before it knows the gates' verdicts); a rejected track's rows (tile_rows = 0) are zero rows, which the update
// ignores (a zero row of H with noise sigma^2 moves nothing).  One workgroup per slot.
__global__ __launch_bounds__(256) void xk_stack_rows(const double *A, const int *tile_rows, const int *trk_off, int DB, int C1P, int row0, double *T) {
  const int t = blockIdx.x;
  const int nr = 2 * (trk_off[t + 1] - trk_off[t]) - 3, off = row0 + 2 * trk_off[t] - 3 * t, valid = min(tile_rows[t], nr);
  const double *src = A + (size_t)t * DB * C1P;
  double *dst = T + (size_t)off * C1P;
  for (int e = threadIdx.x; e < nr * C1P; e += blockDim.x) dst[e] = (e / C1P < valid) ? src[e] : 0.0;
}

__global__ void xk_mark_done(unsigned long long *p, unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }

struct UpdateSpec {
  const double *T;   // c x kdim measurement matrix over state columns [col0, col0+kdim)
  long str, stc;
  int c, kdim, col0;
  const double *z;   // residual (device), stride sz
  long sz;
  const double *rdiag;  // device vector (c) or null -> rscalar
  double rscalar;
  const double *S;   // externally supplied innovation covariance (device, row stride ss, col stride 1)... or null
  long ssr, ssc;
  const double *Pin;  // n x n col-major
  double *Pout;       // n x n col-major (may equal neither Pin)
  const double *ct;   // device corr_total or null
  int cov_update;
  double *corr;       // where the correction goes: null -> h->d_corr (device); xk_apply_update passes pinned host memory
  unsigned long long *done_flag;   // optional completion marker (pinned host memory) written by the last launch ...
  unsigned long long done_seq;     // ... with this value
  int tri;           // T is upper trapezoidal (T[r][k] == 0 for k < r: the compressed R): the products skip the zero blocks
};

template <int RPL>

static void launch_merge(xk_handle *h, XkCaqrArgs &a, int groups, int csplit) {
  hipLaunchKernelGGL(HIP_KERNEL_NAME(xk_caqr_merge<RPL>), dim3(groups, csplit), dim3(16 * (16 + a.chalf)), 0, h->stream, a);
}

// QR compression of the staged tile stack (vio_updater.cpp:487-512): CAQR, panels of 16 columns.
// fuse: (optional) the Kalman update that follows this compression.  If the single launch takes it along (narrow geometry,
// correction_total = 0, covariance update, no external S), h->last_fused says so and the caller must NOT queue launch_update.
// The last columns [ccut, C1) of a tall system in ONE launch or TWO (xk_caqr_pipe<XkPipeTail> / <XkPipeTail4>): the rows of slots
// [slot0, slot0 + nslots) -- as the multi-launch schedule left them after the panels before ccut: R's rows zeroed where they were
// taken out, the leaders' first 32 rows holding merged rows -- plus `nextra` rows from behind the slots (the R of the launch before)
// are gathered into registers once, the panels run as in the single launch of the narrow systems, and rows ccut.. of R go to Rout
// (row stride C1P, column ccut at Rout[0]).
static int launch_pipe_tail(xk_handle *h, bool four, int ccut, int slot0, int nslots, int arity1, int nextra, double *Rout) {
  XkCaqrPipeArgs pa;
  memset(&pa, 0, sizeof(pa));
  pa.A = h->d_A + (size_t)slot0 * h->DB * h->C1P + ccut; pa.tile_rows = h->d_tile_rows + slot0; pa.nslots = nslots; pa.slot_rows = h->DB;
  pa.lead_stride = arity1;                        // (slot0 is a multiple of it)
  pa.nextra = nextra; pa.extra_row0 = (long)(h->ntiles_max - slot0) * h->DB;
  pa.Hc = nullptr; pa.hs = 0; pa.nhc = 0;
  pa.C1P = h->C1P; pa.C1 = h->C1 - ccut; pa.Rout = Rout; pa.S = h->d_rs; pa.PB = h->d_rpb;
  pa.status = h->d_status;
  if (h->xsync_dirty) {
    if (hipMemsetAsync(h->d_xsync, 0, sizeof(unsigned) * 2 * XP_WORDS * 16, h->stream) != hipSuccess) return fail(h, XK_EDEVICE, "sync words");
    if (hipMemsetD32Async((hipDeviceptr_t)h->d_x1, (int)(XK_NOTYET_BITS & 0xffffffffu), 2 * 2 * h->xslab_doubles, h->stream) != hipSuccess) return fail(h, XK_EDEVICE, "slabs");
    h->xsync_dirty = false; h->xsync_phase = 0;
  }
  {
    const size_t np_ = (size_t)(pa.C1 + 15) / 16, strips_ = np_ * XK_PIPE_RLS, x1n = strips_ * 16 * h->C1P;
    double *set = h->d_x1 + (size_t)h->xsync_phase * h->xslab_doubles;
    pa.X1 = set; pa.X2 = set + x1n; pa.X1P = set + 2 * x1n;
    pa.Xnext = h->d_x1 + (size_t)(h->xsync_phase ^ 1) * h->xslab_doubles;
    pa.xnext_doubles = (long)h->xslab_doubles;
  }
  pa.sync = h->d_xsync + (size_t)h->xsync_phase * XP_WORDS * 16;
  pa.sync_next = h->d_xsync + (size_t)(h->xsync_phase ^ 1) * XP_WORDS * 16;
  h->xsync_phase ^= 1;
  if (h->opt_poison) {   // test hook, as in the single launch of the narrow systems: every workgroup gives up at its first spin
    const unsigned seven = 7u;
    if (hipMemcpyAsync(pa.sync + XP_ABORT * 16, &seven, sizeof(unsigned), hipMemcpyHostToDevice, h->stream) != hipSuccess) return fail(h, XK_EDEVICE, "poison");
    if (hipStreamSynchronize(h->stream) != hipSuccess) return fail(h, XK_EDEVICE, "poison");
  }
  static const int pdbg2 = env_int("XK_CAQR_PERSIST_DBG", 0);
  pa.dbg = pdbg2 ? h->d_pdbg : nullptr;
  pa.test_stall = h->opt_test_stall;
  h->pipe_tag = (h->pipe_tag % 0x7fff) + 1;
  pa.acc_tag = h->pipe_tag;
  h->pipe_rows_nominal = 0;                       // (the acceptance ratio belongs to the narrow geometries)
  if (four) hipLaunchKernelGGL(xk_caqr_pipe<XkPipeTail4>, dim3(h->n_cu), dim3(XK_PIPE_THREADS), 0, h->stream, pa);
  else hipLaunchKernelGGL(xk_caqr_pipe<XkPipeTail>, dim3(h->n_cu), dim3(XK_PIPE_THREADS), 0, h->stream, pa);
  h->last_tail = true;
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(h, XK_EDEVICE, "caqr tail launch", e);
  return XK_OK;
}

static int launch_compress(xk_handle *h, hipEvent_t mid = nullptr, const UpdateSpec *fuse = nullptr) {
  h->last_fused = false;
  h->last_tail = false;
  // A compression that xk_build_compress_async left for xk_apply_update is no longer pending once ANY compression runs (xk_apply_update
  // takes the flag down before it comes here; an xk_qr_compress in between does the work now, and xk_apply_update then applies d_R as
  // it stands instead of compressing rows the multi-launch schedule has already reduced in place).
  h->compress_deferred = false;
  if (!h->have_rows) return fail(h, XK_EINVAL, "xk_msckf_build has not run on the staged inputs");
  const int slam_tiles = (2 * h->M + h->DB - 1) / h->DB;
  const int ntiles = h->K + h->K2 + slam_tiles;
  if (ntiles == 0) {   // no measurement rows at all: [T_H | z] = 0 (the reference skips the update, updater.cpp:106)
    if (hipMemsetAsync(h->d_R, 0, sizeof(double) * (size_t)h->C1P * h->C1P, h->stream) != hipSuccess) return fail(h, XK_EDEVICE, "R memset");
    h->nleaf = 0; h->nlevels = 0; h->have_R = true;
    return XK_OK;
  }
  // (d_R was zeroed at creation; the merges rewrite the whole upper trapezoid every update and nothing else)
  XkCaqrArgs a;
  memset(&a, 0, sizeof(a));
  a.A = h->d_A; a.tile_rows = h->d_tile_rows; a.ntiles = ntiles; a.TS = h->DB;
  a.C1P = h->C1P; a.C1 = h->C1; a.Rout = h->d_R; a.dbg = nullptr;
  static const int wt_env = env_int("XK_CAQR_WT", 0);
  a.wt = wt_env;
  {   // tallest tile: 2 L_max - 3 rows for the track tiles, full slots for packed SLAM rows
    const int lmax = std::max(h->K > 0 ? h->h_pin_i[0] : 0, h->K2 > 0 ? h->h_pin_i[1] : 0);
    a.rows_max = (h->M > 0) ? h->DB : std::min(h->DB, std::max(16, 2 * lmax - 3));
  }
  // first-level arity: 40 strips per workgroup once 20 x 20 no longer covers the stack in two levels
  static const int arity1_env = env_int("XK_CAQR_ARITY1", 0);
  const int arity1 = arity1_env ? arity1_env : (ntiles > 400 ? 40 : 20);
  static const int chalf = env_int("XK_CAQR_CHALF", 8);
  static const int overlap_env = env_int("XK_CAQR_OVERLAP", 1);
  // per-tile kernel: 4 lanes per column, at most 192 (64-row tiles) / 128 (128-row tiles) columns per workgroup
  const int tile_cols = (h->DB == 64) ? 192 : 128;
  const int groups1 = (ntiles + arity1 - 1) / arity1;
  // overlapped schedule (xk_caqr_fused): exactly two merge levels, the second one a single 20-way group
  const bool overlap = overlap_env && (arity1 == 20 || arity1 == 40) && groups1 >= 2 && groups1 <= 20;
  a.hole_stride = 0; a.lead_off = 0; a.lead_all = 0; a.pend = 0;
  static const int skip_env = env_int("XK_CAQR_SKIP_REJECTED", 1);
  a.lead_stride = skip_env ? arity1 : 0;
  int launches = 0;
  // register-resident single launch (xk_caqr_pipe.hip.h): MSCKF tracks only, valid rows <= 184 fat tiles of 128
  const int resident_env = h->opt_resident;
  const bool fast_shape = h->K + h->K2 > 0 || h->M > 0;
  const int sp_mode = split_plan(h);              // (before the re-arming below: what compressed_spec saw)
  const bool sp = sp_mode == 1;
  h->split_active = 0;
  if (sp_mode == 3) {
    // a small stack, not compressed (see split_plan): tracks' rows by slot, MSCKF-SLAM tracks' behind them, then the SLAM rows
    const int Rt = h->K > 0 ? 2 * h->h_trk_off[h->K] - 3 * h->K : 0;   // (h_trk_off holds nothing when no track is staged)
    if (h->K > 0) hipLaunchKernelGGL(xk_stack_rows, dim3(h->K), dim3(256), 0, h->stream, h->d_A, h->d_tile_rows, h->d_trk_off, h->DB, h->C1P, 0, h->d_R2);
    if (h->K2 > 0)
      hipLaunchKernelGGL(xk_stack_rows, dim3(h->K2), dim3(256), 0, h->stream, h->d_A + (size_t)h->K * h->DB * h->C1P, h->d_tile_rows + h->K, h->d_trk2_off, h->DB,
                         h->C1P, Rt, h->d_R2);
    const long Rall = split_rows_nominal(h);
    if (h->M > 0 && hipMemcpyAsync(h->d_R2 + (size_t)Rall * h->C1P, h->d_A + (size_t)(h->K + h->K2) * h->DB * h->C1P, sizeof(double) * 2 * (size_t)h->M * h->C1P,
                                   hipMemcpyDeviceToDevice, h->stream) != hipSuccess)
      return fail(h, XK_EDEVICE, "SLAM rows");
    if (mid) hipEventRecord(mid, h->stream);
    h->split_active = 3;
    h->nleaf = 0; h->nlevels = 0; h->have_R = true; h->last_resident = false; h->last_pipe = false;
    hipError_t e3 = hipGetLastError();
    if (e3 != hipSuccess) return fail(h, XK_EDEVICE, "stack rows", e3);
    return XK_OK;
  }
  if (sp_mode == 2) {
    // SLAM rows only: no compression (see split_plan) -- the rows as built are the system the update applies
    if (hipMemcpyAsync(h->d_R2 + (size_t)6 * h->N * h->C1P, h->d_A, sizeof(double) * 2 * (size_t)h->M * h->C1P, hipMemcpyDeviceToDevice, h->stream) != hipSuccess)
      return fail(h, XK_EDEVICE, "SLAM rows");
    if (mid) hipEventRecord(mid, h->stream);
    h->split_active = 2;
    h->nleaf = 0; h->nlevels = 0; h->have_R = true; h->last_resident = false; h->last_pipe = false;
    return XK_OK;
  }
  if (resident_env && !h->persist_ok && h->fast_capable && h->rearm_after > 0 && fast_shape && ++h->clean_classic > h->rearm_after) {
    h->persist_ok = true;                         // (the sync words of a launch that gave up are cleared below)
    h->clean_classic = 0;
  }
  if (resident_env && h->persist_ok && fast_shape) {
    // The launch compacts the stack itself (xk_pipe_rowplan: rows of rejected tracks cost nothing), so what the tiles must hold
    // is the rows that PASS the gates -- which the host does not know when it queues the launch.  It queues on the nominal count
    // (every track accepted) up to a quarter over the capacity; a launch that finds more accepted rows than its tiles hold gives
    // up at once (reason 9) and the multi-launch schedule serves the update -- and the following ones of that size.
    // (split compression, see xk_handle::d_R2: the tracks' rows only, in the pose columns + the residual)
    const int C1s = sp ? 6 * h->N + 1 : h->C1, nslots_p = sp ? h->K + h->K2 : ntiles;
    const bool narrow = C1s <= XkPipeNarrow::COLS;
    const int rows_cap = narrow ? XkPipeNarrow::ROWS : XkPipeWide::ROWS;
    int R_nom = sp ? 0 : 2 * h->M;
    for (int k = 0; k < h->K; ++k) R_nom += 2 * (h->h_trk_off[k + 1] - h->h_trk_off[k]) - 3;
    for (int k = 0; k < h->K2; ++k) R_nom += 2 * (h->h_trk2_off[k + 1] - h->h_trk2_off[k]) - 3;
    // two first-level groups per XCD when the rows expected to pass fit 152 tiles (2 % and half a tile's worth of margin); a launch
    // that finds more gives up at once (reason 9) and that geometry stays off for a while -- the 184-tile launch redoes the update
    bool split = false;
    if (narrow && h->opt_split > 0) {
      if (h->split_backoff > 0) --h->split_backoff;
      else if (h->opt_split >= 3) split = true;                              // (lab: always -- a stack that does not fit gives up, reason 9)
      else if (h->opt_split == 2) split = R_nom <= XkPipeNarrow2::ROWS;
      else if (h->acc_ratio > 0.0) split = (long)(h->acc_ratio * 1.02 * R_nom) + 64 <= XkPipeNarrow2::ROWS;
    }
    const int NTP = 8 * (narrow ? (split ? XkPipeNarrow2::NT : XkPipeNarrow::NT) : XkPipeWide::NT);
    h->pipe_rows_nominal = R_nom;
    if (R_nom >= h->opt_pipe_min_rows && nslots_p <= XK_PIPE_SLOTS_MAX && (long)R_nom * 4 <= (long)rows_cap * 5 && (h->overflow_rows == 0 || R_nom < h->overflow_rows)) {
      XkCaqrPipeArgs pa;
      memset(&pa, 0, sizeof(pa));
      pa.A = h->d_A; pa.tile_rows = h->d_tile_rows; pa.nslots = nslots_p; pa.slot_rows = 64;   // (no leaders, no extra rows: lead_stride = nextra = 0)
      pa.Hc = h->d_Hc; pa.hs = h->hc_stride; pa.nhc = h->rows_compact ? h->K : 0;
      pa.C1P = h->C1P; pa.C1 = C1s; pa.Rout = sp ? h->d_R2 : h->d_R; pa.S = h->d_rs; pa.PB = h->d_rpb;
      pa.res_col = sp ? h->na : 0;
      h->split_active = sp ? 1 : 0;
      pa.status = h->d_status;
      if (h->xsync_dirty) {
        if (hipMemsetAsync(h->d_xsync, 0, sizeof(unsigned) * 2 * XP_WORDS * 16, h->stream) != hipSuccess) return fail(h, XK_EDEVICE, "sync words");
        // (a launch that gave up leaves its slabs half written and the other set half re-armed: arm both)
        if (hipMemsetD32Async((hipDeviceptr_t)h->d_x1, (int)(XK_NOTYET_BITS & 0xffffffffu), 2 * 2 * h->xslab_doubles, h->stream) != hipSuccess) return fail(h, XK_EDEVICE, "slabs");
        h->xsync_dirty = false; h->xsync_phase = 0;
      }
      {
        // (panels of THIS launch's system: the launch re-arms the other set by the same count, xk_caqr_pipe entry)
        const size_t np_ = (size_t)(C1s + 15) / 16, strips_ = np_ * XK_PIPE_RLS, x1n = strips_ * 16 * h->C1P;
        double *set = h->d_x1 + (size_t)h->xsync_phase * h->xslab_doubles;
        pa.X1 = set; pa.X2 = set + x1n; pa.X1P = set + 2 * x1n;
        pa.Xnext = h->d_x1 + (size_t)(h->xsync_phase ^ 1) * h->xslab_doubles;
        pa.xnext_doubles = (long)h->xslab_doubles;
      }
      pa.sync = h->d_xsync + (size_t)h->xsync_phase * XP_WORDS * 16;
      pa.sync_next = h->d_xsync + (size_t)(h->xsync_phase ^ 1) * XP_WORDS * 16;
      h->xsync_phase ^= 1;
      // test hook: raise the abort word before the launch -- every workgroup gives up at its first spin, exactly what an
      // uneven placement or a missing workgroup leads to, and the host has to redo the update with the multi-launch schedule
      if (h->opt_poison) {
        const unsigned seven = 7u;
        if (hipMemcpyAsync(pa.sync + XP_ABORT * 16, &seven, sizeof(unsigned), hipMemcpyHostToDevice, h->stream) != hipSuccess)
          return fail(h, XK_EDEVICE, "poison");
        if (hipStreamSynchronize(h->stream) != hipSuccess) return fail(h, XK_EDEVICE, "poison");
      }
      static const int pdbg2 = env_int("XK_CAQR_PERSIST_DBG", 0);
      pa.dbg = pdbg2 ? h->d_pdbg : nullptr;
      pa.test_stall = h->opt_test_stall;
      h->pipe_tag = (h->pipe_tag % 0x7fff) + 1;               // 1 .. 32767: tags the accepted-rows word of THIS launch (eval_status)
      pa.acc_tag = h->pipe_tag;
      pa.kal = 0; pa.kn = h->n; pa.Pin = nullptr; pa.Pout = nullptr; pa.sigma2 = 0.0; pa.corr = nullptr; pa.ct = nullptr; pa.done_flag = nullptr; pa.done_seq = 0;
      if (fuse && narrow && h->opt_kalman && !fuse->S && !fuse->rdiag && fuse->T == h->d_R &&
          h->n <= 206 && h->n_cu == 256) {
        // (a pass that leaves the covariance alone, cov_update = 0: the role needs the block-by-block posterior to get the
        //  correction right, so it runs as ever and its posterior goes to a scratch matrix; Pout becomes a copy of the prior below)
        pa.kal = 1; pa.Pin = fuse->Pin; pa.Pout = fuse->cov_update ? fuse->Pout : h->d_tmpP; pa.sigma2 = fuse->rscalar; pa.ct = fuse->ct;
        pa.corr = fuse->corr ? fuse->corr : h->d_corr;
        pa.done_flag = fuse->done_flag; pa.done_seq = fuse->done_seq;
        h->last_fused = true;
      }
      h->last_split = split;
      if (split) hipLaunchKernelGGL(xk_caqr_pipe<XkPipeNarrow2>, dim3(h->n_cu), dim3(XK_PIPE_THREADS), 0, h->stream, pa);
      else if (narrow) hipLaunchKernelGGL(xk_caqr_pipe<XkPipeNarrow>, dim3(h->n_cu), dim3(XK_PIPE_THREADS), 0, h->stream, pa);
      else hipLaunchKernelGGL(xk_caqr_pipe<XkPipeWide>, dim3(h->n_cu), dim3(XK_PIPE_THREADS), 0, h->stream, pa);
      if (sp) {
        // the SLAM features' rows go into the compressed system as they were built: 2 M rows behind R1's 6 N (BEHIND the launch: row 6 N
        // of its output -- the residual column's own row of R, which nobody reads -- is the first of them)
        const int slam0 = h->K + h->K2;
        if (hipMemcpyAsync(h->d_R2 + (size_t)6 * h->N * h->C1P, h->d_A + (size_t)slam0 * h->DB * h->C1P, sizeof(double) * 2 * (size_t)h->M * h->C1P,
                           hipMemcpyDeviceToDevice, h->stream) != hipSuccess)
          return fail(h, XK_EDEVICE, "SLAM rows");
      }
      if (pa.kal && !fuse->cov_update && fuse->Pout != fuse->Pin &&
          hipMemcpyAsync(fuse->Pout, fuse->Pin, sizeof(double) * (size_t)h->n * h->n, hipMemcpyDeviceToDevice, h->stream) != hipSuccess)
        return fail(h, XK_EDEVICE, "prior copy");
      if (mid) hipEventRecord(mid, h->stream);
      h->nleaf = NTP; h->nlevels = 1; h->have_R = true; h->last_resident = true; h->last_pipe = true;
      hipError_t e = hipGetLastError();
      if (e != hipSuccess) return fail(h, XK_EDEVICE, "caqr launch", e);
      return XK_OK;
    }
  }
  // (the tiles of the tracks may be factor records: the first tile pass below forms its rows from them and leaves tiles behind)
  a.Hc = h->d_Hc; a.hs = h->hc_stride; a.hcvr = xk_hc_vr(h->DB); a.nhc = (h->rows_compact && h->K > 0) ? h->K : 0;
  h->rows_compact = false;
  // 128-row slots whose tallest tile has <= 104 rows: 26 rows per lane (two workgroups per CU), see xk_caqr_tile
  const int tall26_env = h->opt_tall26;
  auto tile_geom = [&](int c0, int &tsplit, int &tchalf, int &tthreads) {
    const int trail = std::max(0, h->C1 - c0 - 16);
    tsplit = std::max(1, (trail + (tile_cols - 16) - 1) / (tile_cols - 16));
    tchalf = (trail + tsplit - 1) / tsplit;
    tthreads = round_up(4 * (16 + tchalf), 64);
  };
  auto launch_tile = [&](XkCaqrArgs &t) {
    int tsplit, tthreads;
    tile_geom(t.c0, tsplit, t.chalf, tthreads);
    const dim3 tgrid(ntiles, tsplit), tblock(tthreads);
    if (h->DB == 64) {
      if (tsplit == 1) hipLaunchKernelGGL(HIP_KERNEL_NAME(xk_caqr_tile<16, false>), tgrid, tblock, 0, h->stream, t);
      else hipLaunchKernelGGL(HIP_KERNEL_NAME(xk_caqr_tile<16, true>), tgrid, tblock, 0, h->stream, t);
    } else if (t.rows_max <= 104 && tall26_env) {
      if (tsplit == 1) hipLaunchKernelGGL(HIP_KERNEL_NAME(xk_caqr_tile<26, false>), tgrid, tblock, 0, h->stream, t);
      else hipLaunchKernelGGL(HIP_KERNEL_NAME(xk_caqr_tile<26, true>), tgrid, tblock, 0, h->stream, t);
    } else {
      if (tsplit == 1) hipLaunchKernelGGL(HIP_KERNEL_NAME(xk_caqr_tile<32, false>), tgrid, tblock, 0, h->stream, t);
      else hipLaunchKernelGGL(HIP_KERNEL_NAME(xk_caqr_tile<32, true>), tgrid, tblock, 0, h->stream, t);
    }
  };
  // Tall systems: the LAST columns in one launch with every row in registers (xk_caqr_pipe<XkPipeTail>, xk_caqr_pipe.hip.h).  ccut =
  // first column the tail takes (a panel boundary; 0: no tail).  Like the single launch of the narrow systems it is queued on the
  // nominal row count up to 5/4 of its capacity -- the launch counts the rows that passed the gates itself and gives up at once
  // (reason 9) when they do not fit -- and only reads the stack: a tail that gives up is redone from the rows as they stand.
  int ccut = 0, tail_half = 0;                    // tail_half > 0: two launches, slots [0, tail_half) then [tail_half, ntiles) + the first one's R
  if (h->tail_capable && !h->tail_ok && h->opt_resident && h->opt_tail && h->rearm_after > 0 && h->tail_backoff == 0 && ++h->tail_clean > h->rearm_after) {
    h->tail_ok = true;
    h->tail_clean = 0;
  }
  if (h->tail_backoff > 0) --h->tail_backoff;
  else if (overlap && h->tail_ok && h->opt_resident && h->opt_tail && ntiles <= XK_PIPE_SLOTS_MAX) {
    // nominal rows (every track accepted) of the slots before each group boundary -- MSCKF tracks, then tracks that become features,
    // then the SLAM rows packed DB to a slot (vio_updater.cpp:406-422) --; a leader's first 32 rows count whatever its slot holds
    std::vector<long> pre((size_t)groups1 + 1, 0);
    auto slot_rows_nominal = [&](int t) {
      if (t < h->K) return 2 * (h->h_trk_off[t + 1] - h->h_trk_off[t]) - 3;
      if (t < h->K + h->K2) return 2 * (h->h_trk2_off[t - h->K + 1] - h->h_trk2_off[t - h->K]) - 3;
      return std::min(h->DB, 2 * h->M - (t - h->K - h->K2) * h->DB);
    };
    for (int g = 0; g < groups1; ++g) {
      long r = 0;
      for (int t = g * arity1; t < std::min(ntiles, (g + 1) * arity1); ++t) {
        const int v = slot_rows_nominal(t);
        r += (t % arity1 == 0) ? std::max(v, 32) : v;
      }
      pre[g + 1] = pre[g] + r;
    }
    const long R_nom = pre[groups1];
    if (R_nom >= 64 * 8) {
    const int cc4 = 16 * std::max(1, (h->C1 - XkPipeTail4::COLS + 15) / 16), cc8 = 16 * ((h->C1 - XkPipeTail::COLS + 15) / 16);
    const int nx = h->C1 - cc4;                   // rows of the first launch's R
    int gh = 0;                                   // groups in the first half: the boundary that balances first half against second half + R
    for (int g = 1; g < groups1; ++g)
      if (std::labs(2 * pre[g] - R_nom - nx) < std::labs(2 * pre[gh] - R_nom - nx) || gh == 0) gh = g;
    const long capq = (long)XkPipeTail4::ROWS * 5 / 4;
    h->tail_four = false;
    if (h->opt_tail != 2 && R_nom * 4 <= (long)XkPipeTail4::ROWS * 5) { ccut = cc4; tail_half = 0; h->tail_four = true; }   // (the whole stack fits one 192-column launch)
    else if (h->opt_tail != 2 && gh > 0 && pre[gh] <= capq && R_nom - pre[gh] + nx <= capq) { ccut = cc4; tail_half = gh * arity1; h->tail_four = true; }
    else if (cc8 >= 16 && R_nom * 4 <= (long)XkPipeTail::ROWS * 5) { ccut = cc8; tail_half = 0; }
    if (ccut >= h->C1) ccut = 0;
    }
  }
  h->last_tail = false;
  if (overlap) {
    a.rows_max = std::max(a.rows_max, 32);        // a leader's pivot strip alternates between rows 0..15 and 16..31
    for (int c0 = 0, k = 0; c0 < h->C1; c0 += 16, ++k) {
      const int trail = std::max(0, h->C1 - c0 - 16);
      const int csplit = std::max(1, (trail + chalf - 1) / chalf);
      // last level inside the fused launch: 32 lanes per column next to 64-row tiles (768-thread workgroups),
      // 16 next to 128-row tiles (512-thread workgroups); whole waves either way
      // (2 trailing columns per workgroup next to 64-row tiles: 9 waves; measured 22.5 us per fused launch against
      //  23.4 at 4-8 columns -- the fewer waves share a step, the shorter it is, and there are CUs to spare)
      static const int lchalf_env = env_int("XK_CAQR_LCHALF", 0);
      const int llanes = (h->DB == 64) ? 32 : 16;
      // (about 84 last-level workgroups at most: wider systems take more columns per workgroup)
      const int lauto = std::min(8, 2 * std::max(1, (trail + 2 * 84 - 1) / (2 * 84)));
      const int lchalf = (h->DB == 64) ? std::min(8, 2 * std::max(1, (lchalf_env ? lchalf_env : lauto) / 2))
                                       : std::min(16, 4 * std::max(1, (lchalf_env ? lchalf_env : 16) / 4));   // (16: config 3 418 -> 425 updates/s against 8, round 5 sweep)
      const int lsplit = std::max(1, (trail + lchalf - 1) / lchalf);
      const int lead_off = (k & 1) ? 16 : 0;
      if (k == 0) {
        XkCaqrArgs t = a;
        t.c0 = 0; t.stride = 1; t.final_level = 0; t.pin = nullptr; t.pout = h->d_panel[0];
        launch_tile(t);
        if (mid) hipEventRecord(mid, h->stream);
      }
      XkCaqrArgs m = a;                            // first level: leaders' pivot strips at lead_off, + the pending strips
      // columns per workgroup: at least `chalf`, and enough that groups x splits fits one workgroup per CU -- two
      // merge workgroups on a CU run ~1.6x longer than one (28.8 us at 420 workgroups, 17-19 us below 256)
      static const int adapt = env_int("XK_CAQR_ADAPT", 1);
      static const int cus = env_int("XK_CAQR_CUS", 256);
      const int per_cu = std::max(1, cus / groups1);
      const int mchalf = adapt ? std::min(arity1 == 40 ? 16 : 32, std::max(chalf, 2 * ((trail + 2 * per_cu - 1) / (2 * per_cu)))) : chalf;
      const int msplit = std::max(1, (trail + mchalf - 1) / mchalf);
      m.c0 = c0; m.stride = 1; m.final_level = 0; m.pin = h->d_panel[0]; m.pout = h->d_panel[1]; m.chalf = mchalf;
      m.lead_off = lead_off; m.lead_all = 0; m.pend = (k > 0) ? 1 : 0;
      static const int m32_env = env_int("XK_CAQR_M32", 1);
      if (arity1 == 40 && m32_env) hipLaunchKernelGGL(xk_caqr_merge32, dim3(groups1, msplit), dim3(32 * (16 + mchalf)), 0, h->stream, m);
      else if (arity1 == 40) launch_merge<42>(h, m, groups1, msplit);
      else launch_merge<22>(h, m, groups1, msplit);
      ++launches;
      XkCaqrArgs l = a;                            // last level: the leaders' pivot strips -> 16 rows of R
      l.c0 = c0; l.stride = arity1; l.final_level = 1; l.pin = h->d_panel[1]; l.pout = h->d_panel[0]; l.chalf = lchalf;
      l.lead_off = lead_off; l.lead_all = 1; l.pend = 0;
      if (ccut > 0 && c0 + 16 == ccut) {
        // the last panel of the multi-launch part: its last level alone, then the tail launch takes the stack as it stands
        launch_merge<20>(h, l, 1, lsplit);
        ++launches;
        int rct;
        if (tail_half > 0) {
          double *Ra = h->d_A + (size_t)h->ntiles_max * h->DB * h->C1P + ccut;           // the first launch's R: behind the slots
          rct = launch_pipe_tail(h, true, ccut, 0, tail_half, arity1, 0, Ra);
          if (rct == XK_OK) rct = launch_pipe_tail(h, true, ccut, tail_half, ntiles - tail_half, arity1, h->C1 - ccut, h->d_R + (size_t)ccut * h->C1P + ccut);
          ++launches;
        } else rct = launch_pipe_tail(h, h->tail_four, ccut, 0, ntiles, arity1, 0, h->d_R + (size_t)ccut * h->C1P + ccut);
        if (rct != XK_OK) return rct;
        ++launches;
        break;
      }
      if (c0 + 16 < h->C1) {
        XkCaqrArgs t = a;                          // ... next to the tile step of the next panel
        t.c0 = c0 + 16; t.stride = 1; t.final_level = 0; t.pin = nullptr; t.pout = h->d_panel[0];
        t.hole_stride = arity1; t.lead_off = 16 - lead_off;
        int tsplit, tthreads;
        tile_geom(t.c0, tsplit, t.chalf, tthreads);
                const dim3 grid(lsplit + ntiles * tsplit), block(std::max(tthreads, round_up(llanes * (16 + lchalf), 64)));
        if (h->DB == 64) {
          if (tsplit == 1) hipLaunchKernelGGL(HIP_KERNEL_NAME(xk_caqr_fused<16, false>), grid, block, 0, h->stream, t, l, lsplit, tsplit);
          else hipLaunchKernelGGL(HIP_KERNEL_NAME(xk_caqr_fused<16, true>), grid, block, 0, h->stream, t, l, lsplit, tsplit);
        } else if (t.rows_max <= 104 && tall26_env) {
          if (tsplit == 1) hipLaunchKernelGGL(HIP_KERNEL_NAME(xk_caqr_fused<26, false>), grid, block, 0, h->stream, t, l, lsplit, tsplit);
          else hipLaunchKernelGGL(HIP_KERNEL_NAME(xk_caqr_fused<26, true>), grid, block, 0, h->stream, t, l, lsplit, tsplit);
        } else {
          if (tsplit == 1) hipLaunchKernelGGL(HIP_KERNEL_NAME(xk_caqr_fused<32, false>), grid, block, 0, h->stream, t, l, lsplit, tsplit);
          else hipLaunchKernelGGL(HIP_KERNEL_NAME(xk_caqr_fused<32, true>), grid, block, 0, h->stream, t, l, lsplit, tsplit);
        }
      } else {
        launch_merge<20>(h, l, 1, lsplit);
      }
      ++launches;
    }
  } else {
    for (int c0 = 0; c0 < h->C1; c0 += 16) {
      const int trail = std::max(0, h->C1 - c0 - 16);
      a.c0 = c0; a.stride = 1; a.final_level = 0; a.pin = nullptr; a.pout = h->d_panel[0];
      launch_tile(a);
      if (c0 == 0 && mid) hipEventRecord(mid, h->stream);
      if (c0 > 0) ++launches;                                        // (the first tile launch is timed as its own stage)
      a.chalf = chalf;
      const int csplit = std::max(1, (trail + a.chalf - 1) / a.chalf);
      int stride = 1, level = 0;
      do {
        const int left = (ntiles + stride - 1) / stride;             // strips still alive at this level
        const int arity = (stride == 1) ? arity1 : (left > 20 ? 40 : 20);
        a.stride = stride;
        a.final_level = (left <= arity) ? 1 : 0;
        const int groups = (left + arity - 1) / arity;
        a.pin = h->d_panel[level & 1];
        a.pout = h->d_panel[(level + 1) & 1];
        ++level;
        if (arity == 40) launch_merge<40>(h, a, groups, csplit);
        else launch_merge<20>(h, a, groups, csplit);
        ++launches;
        stride *= arity;
      } while (stride < ntiles);
    }
  }
  h->nleaf = ntiles;
  h->nlevels = launches;
  h->have_R = true;
  h->last_resident = false; h->last_pipe = h->last_tail;
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(h, XK_EDEVICE, "caqr launch", e);
  return XK_OK;
}

static void gemm(xk_handle *h, const XkGemmArgs &g) {
  const int tiles = xk_gemm_grid(g);
  if (tiles <= 0) return;
  hipLaunchKernelGGL(xk_gemm_f64, dim3(tiles), dim3(64 * XK_GEMM_WAVES), 0, h->stream, g);
}


// Kalman algebra on the device (updater.cpp:117-141 / :144-161).  ev (optional)
// = {before, after-gemm-part...} is not used here; stage split is timed by the caller.
static int launch_update(xk_handle *h, const UpdateSpec &u, float *gemm_ms_accum = nullptr) {
  (void)gemm_ms_accum;
  const int c = u.c, n = h->n, LDA = h->LDA;
  if (c <= 0 || c > h->CM) return fail(h, XK_ECAPACITY, "measurement rows exceed workspace");
  XkGemmArgs g;
  memset(&g, 0, sizeof(g));
  // W = T * Pin[col0:col0+kdim, :]                      (H P)
  g.A = u.T; g.sar = u.str; g.sac = u.stc;
  g.B = u.Pin + u.col0; g.sbr = 1; g.sbc = n;
  g.C = h->d_Maug + c; g.scr = LDA; g.scc = 1;
  g.D = g.C; g.sdr = LDA; g.sdc = 1;
  g.M = c; g.N = n + 1; g.K = u.kdim; g.alpha = 1.0; g.beta = 0.0; g.mode = 0;
  static const int struct_env = env_int("XK_GEMM_STRUCT", 1);   // 0: every product as a general one (A/B switch)
  g.tri_a = struct_env && u.tri;
  // extra column: z' = res + H corr_tot  (updater.cpp:126) lands next to W in the augmented matrix
  g.xcol = 1; g.bx = u.ct ? u.ct + u.col0 : nullptr; g.sbx = 1; g.dx = u.z; g.sdx = u.sz;
  g.cx = h->d_Maug + c + n; g.scx = LDA;
  gemm(h, g);
  if (u.S) {
    XkCopyArgs cp{u.S, h->d_Maug, c, c, u.ssr, u.ssc, (long)LDA, 1};
    hipLaunchKernelGGL(xk_copy2d, dim3((c * c + 255) / 256), dim3(256), 0, h->stream, cp);
  } else {
    // S = W[:, col0:col0+kdim] * T^T + R               (H P H^T + R)
    memset(&g, 0, sizeof(g));
    g.A = h->d_Maug + c + u.col0; g.sar = LDA; g.sac = 1;
    g.B = u.T; g.sbr = u.stc; g.sbc = u.str;
    g.C = h->d_Maug; g.scr = LDA; g.scc = 1;
    g.D = g.C; g.sdr = LDA; g.sdc = 1;
    g.M = c; g.N = c; g.K = u.kdim; g.alpha = 1.0; g.beta = 0.0; g.mode = 1;
    g.tri_b = struct_env && u.tri;
    g.sym_cols = struct_env ? c : 0;   // only the upper triangle of S is read (xk_chol_whole / xk_chol_step)
    g.diag = u.rdiag; g.diag_scalar = u.rscalar;
    gemm(h, g);
  }
  // blocked Cholesky with the right-hand sides carried along: one launch per 32-column block step
  const int ncols = c + n + 1;
  static const int whole_env = env_int("XK_CHOL_WHOLE", 1);
  static const int split_env = env_int("XK_CHOL_SPLIT", 1);
  if (whole_env && (c <= 16 * XK_CHOLW_MAXB || split_env)) {
    // Every block step inside one launch (xk_chol_whole), one workgroup per 16 right-hand-side columns.  Systems with
    // more than 192 rows (BASELINE configs 2 and 3: c = 331 / 301) are cut into 192-row slabs:
    //   X[0:b, b:]  = L_11^-1 [S_12 | W_1 | z_1]        xk_chol_whole on the slab, the rest of ITS ROWS as right-hand sides
    //   M[b:, b:]  -= X[0:b, b:c]^T X[0:b, b:]           one fp64-MFMA GEMM (Schur complement of S and of the right-hand sides)
    // and the remainder is the same problem again: 2 slabs = 3 launches instead of 11 block-step launches.
    const int B = 16 * XK_CHOLW_MAXB;
    for (int off = 0; off < c;) {
      const int cb = std::min(B, c - off);
      XkCholWholeArgs d;
      d.Maug = h->d_Maug + (size_t)off * LDA + off; d.ld = LDA; d.c = cb; d.ncols = ncols - off;
      d.X = h->d_X + (size_t)off * LDA + off; d.status = h->d_status;
      xk_cholw_table((cb + 15) / 16, d.tab);
      hipLaunchKernelGGL(xk_chol_whole, dim3((ncols - off - cb + 15) / 16), dim3(64 * XK_CHOLW_WAVES), 0, h->stream, d);
      off += cb;
      if (off < c) {
        XkGemmArgs s;
        memset(&s, 0, sizeof(s));
        const double *Xs = h->d_X + (size_t)(off - cb) * LDA + off;       // X[slab rows, off:]
        s.A = Xs; s.sar = 1; s.sac = LDA;                                   // A[i][k] = X[k][off + i]
        s.B = Xs; s.sbr = LDA; s.sbc = 1;                                   // B[k][j] = X[k][off + j]
        s.C = h->d_Maug + (size_t)off * LDA + off; s.scr = LDA; s.scc = 1;
        s.D = s.C; s.sdr = LDA; s.sdc = 1;
        s.M = c - off; s.N = ncols - off; s.K = cb; s.alpha = -1.0; s.beta = 1.0; s.mode = 0;
        s.sym_cols = struct_env ? c - off : 0;   // (the Schur complement of S: upper triangle only)
        gemm(h, s);
      }
    }
  } else
  for (int kb = 0; kb < c; kb += XK_CHOL_NB) {
    const int nb = std::min(XK_CHOL_NB, c - kb);
    const int rest = ncols - (kb + nb), mrem = c - kb - nb;
    XkCholStepArgs d{h->d_Maug, LDA, kb, nb, c, ncols, h->d_X, (rest + 15) / 16, h->d_status};
    hipLaunchKernelGGL(xk_chol_step, dim3(d.ncb * (1 + (mrem + 15) / 16)), dim3(64), 0, h->stream, d);
  }
  // P+ = sym(P - X^T X),  X = L^-1 W                   (I-KH)P, (P+P^T)/2
  if (u.cov_update) {
    memset(&g, 0, sizeof(g));
    g.A = h->d_X + c; g.sar = 1; g.sac = LDA;
    g.B = h->d_X + c; g.sbr = LDA; g.sbc = 1;
    g.D = u.Pin; g.sdr = 1; g.sdc = n;
    g.C = u.Pout; g.scr = 1; g.scc = n;
    g.M = n; g.N = n + 1; g.K = c; g.alpha = -1.0; g.beta = 1.0; g.mode = 2;
    g.sym_cols = struct_env ? n : 0;       // tiles below the diagonal: mirror images of the ones above
    // extra column: corr = X^T (L^-1 z') - corr_tot   (K z' - corr_tot, updater.cpp:126)
    g.xcol = 1; g.bx = h->d_X + c + n; g.sbx = LDA; g.ex = u.ct; g.cx = u.corr ? u.corr : h->d_corr; g.scx = 1;
    if (u.done_flag) { g.done_cnt = h->d_done_cnt; g.done_flag = u.done_flag; g.done_seq = u.done_seq; }
    gemm(h, g);
  } else {
    if (u.Pout != u.Pin) hipMemcpyAsync(u.Pout, u.Pin, sizeof(double) * (size_t)n * n, hipMemcpyDeviceToDevice, h->stream);
    XkCorrArgs cr{h->d_X, LDA, c, n, c, c + n, u.ct, u.corr ? u.corr : h->d_corr};
    hipLaunchKernelGGL(xk_corr, dim3((n + 63) / 64), dim3(64), 0, h->stream, cr);
    if (u.done_flag) hipLaunchKernelGGL(xk_mark_done, dim3(1), dim3(1), 0, h->stream, u.done_flag, u.done_seq);
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(h, XK_EDEVICE, "update launch", e);
  return XK_OK;
}

// Will (did) the compression of the staged update take the split form?  One predicate for compressed_spec -- which callers evaluate
// BEFORE launch_compress -- and for launch_compress itself, on handle state neither of them changes in between; it repeats the
// conditions under which the single launch is taken at all (a split system goes nowhere else).
static int split_geometry_rows(const xk_handle *h) { return 6 * h->N + 1 <= XkPipeNarrow::COLS ? XkPipeNarrow::ROWS : XkPipeWide::ROWS; }
static long split_rows_nominal(const xk_handle *h) {
  long r = 0;
  for (int k = 0; k < h->K; ++k) r += 2 * (h->h_trk_off[k + 1] - h->h_trk_off[k]) - 3;
  for (int k = 0; k < h->K2; ++k) r += 2 * (h->h_trk2_off[k + 1] - h->h_trk2_off[k]) - 3;
  return r;
}
static int split_plan(const xk_handle *h) {
  if (!h->d_R2 || !h->opt_slam_split || h->want_full_T) return 0;
  // 3: a SMALL stack -- a handful of tracks ended this frame (+ the SLAM rows): nominal rows <= n.  Not compressed either (the same branch of
  // vio_updater.cpp:487; the reference counts accepted rows, this counts nominal ones: it cannot wait for the verdicts): rows as built -> update.
  if (h->K + h->K2 > 0) {
    const long R = split_rows_nominal(h) + 2L * h->M;
    if (R <= std::min(h->n, h->CM)) return 3;
  }
  if (h->M <= 0) return 0;
  // 2: the stack is the SLAM features' rows and nothing else (no track ended this frame -- the common frame of a filter with persistent
  // features): 2 M rows against n > 3 M columns.  The reference compresses only when rows > columns (vio_updater.cpp:487); neither does this:
  // the rows go to the update as built, no QR launch at all (any n, any window).
  if (h->K + h->K2 == 0) return 2 * h->M <= h->CM ? 2 : 0;
  if (h->DB != 64 || !h->opt_resident || !h->persist_ok) return 0;
  if (h->n <= 206 || 6 * h->N + 1 > XkPipeWide::COLS) return 0;
  const long R = split_rows_nominal(h);
  return (R >= h->opt_pipe_min_rows && h->K + h->K2 <= XK_PIPE_SLOTS_MAX && R * 4 <= (long)split_geometry_rows(h) * 5 &&
          (h->overflow_rows == 0 || R < h->overflow_rows)) ? 1 : 0;
}

static UpdateSpec compressed_spec(xk_handle *h, const double *d_ct, int cov_update) {
  UpdateSpec u;
  memset(&u, 0, sizeof(u));
  if (const int mode = h->have_R ? h->split_active : split_plan(h)) {
    // the split form (d_R2): 6 N rows of R1 over the pose columns, then the 2 M rows of the SLAM features as built; mode 2: those rows alone
    const int r0 = mode == 2 ? 6 * h->N : 0;
    u.T = h->d_R2 + (size_t)r0 * h->C1P; u.str = h->C1P; u.stc = 1;
    u.c = mode == 3 ? (int)split_rows_nominal(h) + 2 * h->M : 6 * h->N + 2 * h->M - r0;   // (3: every track's rows, then the SLAM rows)
    u.kdim = h->na; u.col0 = XK_CORE;
    u.z = u.T + h->na; u.sz = h->C1P;
    u.rdiag = nullptr; u.rscalar = h->sigma_img * h->sigma_img;    // the SLAM rows carry sigma_img^2 too (slam_update.cpp: r = var_img I)
    u.Pin = h->d_P; u.Pout = h->d_Pout; u.ct = d_ct; u.cov_update = cov_update;
    u.tri = 0;                                     // (the SLAM rows are not below anybody's diagonal)
    return u;
  }
  u.T = h->d_R; u.str = h->C1P; u.stc = 1;      // R[0], rows 0..na-1, active columns
  u.c = h->na; u.kdim = h->na; u.col0 = XK_CORE;
  u.z = h->d_R + h->na; u.sz = h->C1P;           // residual column
  u.rdiag = nullptr; u.rscalar = h->sigma_img * h->sigma_img;  // vio_updater.cpp:508-509
  u.Pin = h->d_P; u.Pout = h->d_Pout; u.ct = d_ct; u.cov_update = cov_update;
  u.tri = 1;                                      // d_R: zero below the diagonal (zeroed at creation, only the trapezoid is ever written)
  return u;
}

#define XK_RETRY_CLASSIC 1000   // internal: the single-launch CAQR gave up, the multi-launch schedule must redo the update
static int eval_status(xk_handle *h, int st, int pst, bool allow_retry) {
  // What the last single launch found (status word 2): accepted rows in the low 15 bits, the launch's tag above them.  The word is a
  // relaxed system-scope store of a tile workgroup, not ordered with the completion marker (another workgroup's store): a count that
  // carries another launch's tag is a late arrival and is left alone -- it must not be divided by THIS launch's nominal rows.
  if (h->last_pipe && h->pipe_rows_nominal > 0) {
    const int w2 = h->d_status[2];
    if (w2 > 0 && (w2 >> 15) == h->pipe_tag) h->acc_ratio = (double)(w2 & 0x7fff) / h->pipe_rows_nominal;
  }
  if (pst == 0 && h->last_tail) h->tail_backoff_len = 64;      // (a tail that ran through: the next overflow starts at 64 updates off again)
  if (st != 0 || pst != 0) {
    hipStreamSynchronize(h->stream);
    h->d_status[0] = h->d_status[1] = 0;
  }
  if (pst != 0) {
    // reasons: 1 grid not resident, 2 XCD barrier, 3 uneven XCD placement, 4/5 waiting for the last / first level
    // The fast path steps aside, but not for the life of the handle: after `rearm_after` clean multi-launch updates it is
    // tried again (the other tenant of the GPU may be gone); every further give-up doubles that distance, so a permanently
    // shared GPU costs one bounded retry (<= 2 ms, xk_spin_ge) every few thousand updates at most.
    if (h->last_tail) {
      // the tail launch of a tall system gave up: it only READ the stack, but the retry below rebuilds the rows anyway (one code path);
      // reason 9 = more rows passed the gates than its tiles hold -- off for the next 64 updates; anything else = co-residency
      if (pst == 9) { h->tail_backoff = std::max(64, h->tail_backoff_len); h->tail_backoff_len = std::min(4096, 2 * std::max(64, h->tail_backoff_len)); }
      else { h->tail_ok = false; h->tail_clean = -1; if (h->fast_giveups >= 1) h->rearm_after = std::min(4096, std::max(1, h->rearm_after) * 2); }
      h->fast_giveups++; h->fast_reason = pst;
      h->xsync_dirty = true;
      h->have_rows = h->have_R = false;
      snprintf(h->err, sizeof(h->err), "single-launch CAQR tail gave up (reason %d); the multi-launch schedule finishes the factorisation", pst);
      return allow_retry ? XK_RETRY_CLASSIC : XK_EDEVICE;
    }
    if (pst == 9 && h->last_split) {
      // the 152-tile geometry was chosen on the LAST update's acceptance ratio and this update passed more: not a co-residency
      // problem and not a capacity cliff of the fast path -- the 184-tile launch redoes the update, the split geometry stays off
      // for the next 64 updates
      h->split_backoff = 64;
      h->fast_giveups++; h->fast_reason = pst;
      h->xsync_dirty = true;
      h->have_rows = h->have_R = false;
      snprintf(h->err, sizeof(h->err), "single-launch CAQR (152 tiles): more rows passed the gates than expected; redone with 184 tiles");
      return allow_retry ? XK_RETRY_CLASSIC : XK_EDEVICE;
    }
    if (pst == 9) {
      // not a co-residency problem: more rows passed the gates than the tiles of the single launch hold.  The fast path stays
      // armed for smaller stacks; this size goes to the multi-launch schedule from now on.
      h->overflow_rows = h->overflow_rows ? std::min(h->overflow_rows, h->pipe_rows_nominal) : h->pipe_rows_nominal;
      h->fast_giveups++; h->fast_reason = pst;
      h->xsync_dirty = true;
      h->have_rows = h->have_R = false;
      snprintf(h->err, sizeof(h->err), "single-launch CAQR: %d nominal rows held more accepted rows than its tiles; multi-launch schedule from that size on", h->pipe_rows_nominal);
      return allow_retry ? XK_RETRY_CLASSIC : XK_EDEVICE;
    }
    h->persist_ok = false;
    h->fast_giveups++; h->fast_reason = pst; h->clean_classic = -1;   // (-1: the retry of THIS update is not a clean update)
    if (h->fast_giveups > 1) h->rearm_after = std::min(4096, std::max(1, h->rearm_after) * 2);
    h->xsync_dirty = true;
    h->have_rows = h->have_R = false;
    snprintf(h->err, sizeof(h->err), "single-launch CAQR gave up (reason %d): workgroups not co-resident; multi-launch schedule for the next %d updates", pst, h->rearm_after);
    return allow_retry ? XK_RETRY_CLASSIC : XK_EDEVICE;
  }
  if (st != 0) return fail(h, st, "innovation covariance not positive definite");
  return XK_OK;
}
static int read_status(xk_handle *h, bool allow_retry = false) {
  HIPCHK(h, hipStreamSynchronize(h->stream));
  stage_stream_idle(h);
  return eval_status(h, h->d_status[0], h->d_status[1], allow_retry);
}

static int fetch_flags(xk_handle *h, int *inl, double *gam, int *inls, double *gams) {
  if (h->K > 0 && inl) HIPCHK(h, hipMemcpyAsync(inl, h->d_inl, sizeof(int) * h->K, hipMemcpyDeviceToHost, h->stream));
  if (h->K > 0 && gam) HIPCHK(h, hipMemcpyAsync(gam, h->d_gam, sizeof(double) * h->K, hipMemcpyDeviceToHost, h->stream));
  if (h->M > 0 && inls) HIPCHK(h, hipMemcpyAsync(inls, h->d_inl_s, sizeof(int) * h->M, hipMemcpyDeviceToHost, h->stream));
  if (h->M > 0 && gams) HIPCHK(h, hipMemcpyAsync(gams, h->d_gam_s, sizeof(double) * h->M, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return XK_OK;
}

// ---------------------------------------------------------------------------
// public staged API
// ---------------------------------------------------------------------------
extern "C" int xk_msckf_build(xk_handle *h, double sigma_img, int *inlier_msckf, double *gamma_msckf,
                              int *inlier_slam, double *gamma_slam) {
  if (!h || !(sigma_img > 0.0)) return XK_EINVAL;
  HIPCHK(h, hipSetDevice(h->device));
  int rc = launch_build(h, sigma_img);
  if (rc != XK_OK) return rc;
  return fetch_flags(h, inlier_msckf, gamma_msckf, inlier_slam, gamma_slam);
}

extern "C" int xk_qr_compress(xk_handle *h, double *T_H, int ldt, double *z) {
  if (!h) return XK_EINVAL;
  HIPCHK(h, hipSetDevice(h->device));
  h->want_full_T = true;                         // (this call hands out the reference's upper-triangular T_H: no split compression)
  int rc = launch_compress(h);
  if (rc == XK_OK) rc = read_status(h, true);
  if (rc == XK_RETRY_CLASSIC) {                  // rebuild the rows (the tiles were worked on in place) and compress the slow way
    if ((rc = launch_build(h, h->sigma_img)) == XK_OK && (rc = launch_compress(h)) == XK_OK) rc = read_status(h);
  }
  h->want_full_T = false;
  if (rc != XK_OK) return rc;
  if (T_H || z) {
    if (T_H && ldt < h->n) return XK_EINVAL;
    std::vector<double> R((size_t)h->C1 * h->C1P);
    HIPCHK(h, hipMemcpy(R.data(), h->d_R, sizeof(double) * R.size(), hipMemcpyDeviceToHost));
    if (T_H) {
      for (int j = 0; j < h->n; ++j)
        for (int i = 0; i < h->n; ++i) T_H[i + (size_t)j * ldt] = 0.0;
      // rows 0..na-1 of the triangle, placed at rows 0.. with the core columns zero
      for (int i = 0; i < h->na; ++i)
        for (int j = i; j < h->na; ++j) T_H[i + (size_t)(XK_CORE + j) * ldt] = R[(size_t)i * h->C1P + j];
    }
    if (z) {
      for (int i = 0; i < h->n; ++i) z[i] = 0.0;
      for (int i = 0; i < h->na; ++i) z[i] = R[(size_t)i * h->C1P + h->na];
    }
  }
  return XK_OK;
}

// gate results of the last build -> the handle's pinned cache (asynchronous; valid after the next synchronisation)
static int cache_flags(xk_handle *h) {
  // MSCKF gate results: the per-feature kernel writes them into the pinned cache itself (XkFeatArgs::inlier_h / gamma_h);
  // without SLAM rows there is nothing to copy and nothing to wait for but the stream (five runtime calls less per frame)
  h->flags_direct = h->M == 0;
  if (!h->flags_direct) {
    HIPCHK(h, hipEventRecord(h->ev_flags, h->stream));               // the per-feature kernels have been queued
    HIPCHK(h, hipStreamWaitEvent(h->copy_stream, h->ev_flags, 0));
    HIPCHK(h, hipMemcpyAsync(h->h_flag_i + h->Kmax, h->d_inl_s, sizeof(int) * h->M, hipMemcpyDeviceToHost, h->copy_stream));
    HIPCHK(h, hipMemcpyAsync(h->h_flag_d + h->Kmax, h->d_gam_s, sizeof(double) * h->M, hipMemcpyDeviceToHost, h->copy_stream));
    HIPCHK(h, hipEventRecord(h->ev_flags_done, h->copy_stream));
  }
  h->flags_cached = true;
  h->flags_after_seq = h->done_seq;
  return XK_OK;
}

// xk_msckf_build + xk_qr_compress without a host synchronisation and without host outputs: the launches are queued behind
// whatever is already on the handle's stream (staging copies, covariance propagation, manage()) and xk_apply_update's
// one synchronisation covers them all.  The gate results come back with that synchronisation (xk_fetch_flags).
extern "C" int xk_build_compress_async(xk_handle *h, double sigma_img) {
  if (!h || !(sigma_img > 0.0)) return XK_EINVAL;
  HIPCHK(h, hipSetDevice(h->device));
  int rc = launch_build(h, sigma_img);
  if (rc != XK_OK) return rc;
  if ((rc = cache_flags(h)) != XK_OK) return rc;
  // Who calls this instead of xk_build_compress_update[_pass]_async has something between constructUpdate and applyUpdate that
  // rewrites the covariance (the applyCI entries of the MULTI_UAV order, updater.cpp:84-97).  [T_H | z] does not depend on the
  // covariance -- the gates have read the prior in the per-feature kernel above -- so where the single launch can take the Kalman
  // update along (narrow geometry, n <= 206) the compression is not queued now but by xk_apply_update, behind those entries, with
  // the Kalman role on the covariance they left: one launch there instead of one here and five there.
  h->compress_deferred = h->opt_resident && h->persist_ok && h->opt_kalman && h->C1 <= XkPipeNarrow::COLS && h->n <= 206 && h->n_cu == 256 &&
                         h->K + h->K2 + h->M > 0 && split_plan(h) < 2;   // (stacks that are not compressed at all: nothing to defer)
  if (h->compress_deferred) h->have_R = true;     // (as far as xk_apply_update's precondition goes: it runs the compression itself)
  else if ((rc = launch_compress(h)) != XK_OK) return rc;
  h->async_pending = true;
  return XK_OK;
}

// The same with the Kalman update of Updater::applyUpdate(correction_total = 0, cov_update = true) queued as well -- inside the
// compression launch where the geometry allows it (xk_pipe_kalman), behind it otherwise.  xk_apply_update(h, NULL or zeros, 1, ..)
// then only waits for the result.  For callers that know at construction time that nothing comes between constructUpdate and
// applyUpdate (single agent, iekf_iter = 1: updater.cpp:99-110 with one pass) -- not the MULTI_UAV order, whose applyCI entries
// replace the covariance in between (updater.cpp:84-97).
static int build_compress_update_pass(xk_handle *h, double sigma_img, const double *corr_total, int cov_update) {
  if (!h || !(sigma_img > 0.0)) return XK_EINVAL;
  HIPCHK(h, hipSetDevice(h->device));
  bool ct_zero = true;
  if (corr_total)
    for (int i = 0; i < h->n && ct_zero; ++i) ct_zero = corr_total[i] == 0.0;
  const double *dct = nullptr;
  if (!ct_zero) {
    double *st = (double *)stage_slot(h, sizeof(double) * h->n);
    if (!st) return fail(h, XK_ECAPACITY, "staging slot too small");
    memcpy(st, corr_total, sizeof(double) * h->n);
    HIPCHK(h, hipMemcpyAsync(h->d_ct, st, sizeof(double) * h->n, hipMemcpyHostToDevice, h->stream));
    dct = h->d_ct;
  }
  int rc = launch_build(h, sigma_img);
  if (rc != XK_OK) return rc;
  if ((rc = cache_flags(h)) != XK_OK) return rc;
  UpdateSpec u = compressed_spec(h, dct, cov_update ? 1 : 0);
  u.corr = h->h_out;
  static const int spin_env = env_int("XK_SPIN_DONE", 1);
  if (spin_env) { u.done_flag = reinterpret_cast<unsigned long long *>(h->h_out + h->n + 2); u.done_seq = ++h->done_seq; }
  if ((rc = launch_compress(h, nullptr, &u)) != XK_OK) return rc;
  if (!h->last_fused && (rc = launch_update(h, u)) != XK_OK) return rc;
  h->async_pending = true;
  h->fused_pending = true;
  h->fused_seq = spin_env ? u.done_seq : 0;
  h->fused_cov_update = cov_update ? 1 : 0;
  h->fused_ct_zero = ct_zero;
  if (!h->fused_ct) h->fused_ct = new std::vector<double>();
  if (ct_zero) h->fused_ct->clear(); else h->fused_ct->assign(corr_total, corr_total + h->n);
  return XK_OK;
}
extern "C" int xk_build_compress_update_async(xk_handle *h, double sigma_img) { return build_compress_update_pass(h, sigma_img, nullptr, 1); }
extern "C" int xk_build_compress_update_pass_async(xk_handle *h, double sigma_img, const double *corr_total, int cov_update) {
  return build_compress_update_pass(h, sigma_img, corr_total, cov_update);
}

extern "C" int xk_fetch_flags(xk_handle *h, int *inlier_msckf, double *gamma_msckf, int *inlier_slam, double *gamma_slam) {
  if (!h) return XK_EINVAL;
  if (!h->flags_cached) return fail(h, XK_EINVAL, "xk_fetch_flags: no build since the inputs were staged");
  if (!h->flags_direct && hipEventQuery(h->ev_flags_done) != hipSuccess) {   // (normally long done: the copies ran beside the QR kernels)
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipEventSynchronize(h->ev_flags_done));
  }
  // the MSCKF results were written by the per-feature kernel: valid once the stream has passed it (after xk_apply_update
  // it has; otherwise this waits)
  // (a completion marker seen after the build was queued says the same without asking the runtime)
  if (h->K > 0 && !(h->done_seen > h->flags_after_seq) && hipStreamQuery(h->stream) != hipSuccess) {
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
  }
  if (inlier_msckf && h->K > 0) memcpy(inlier_msckf, h->h_flag_i, sizeof(int) * h->K);
  if (gamma_msckf && h->K > 0) memcpy(gamma_msckf, h->h_flag_d, sizeof(double) * h->K);
  if (inlier_slam && h->M > 0) memcpy(inlier_slam, h->h_flag_i + h->Kmax, sizeof(int) * h->M);
  if (gamma_slam && h->M > 0) memcpy(gamma_slam, h->h_flag_d + h->Kmax, sizeof(double) * h->M);
  return XK_OK;
}

extern "C" int xk_apply_update(xk_handle *h, const double *corr_total, int cov_update, double *correction) {
  if (!h || !correction) return XK_EINVAL;
  if (!h->have_R) return fail(h, XK_EINVAL, "xk_qr_compress has not run on the staged inputs");
  HIPCHK(h, hipSetDevice(h->device));
  const double *dct = nullptr;
  bool ct_zero = true;                                 // Updater::update starts every update from a zero correction_total
  if (corr_total)
    for (int i = 0; i < h->n && ct_zero; ++i) ct_zero = corr_total[i] == 0.0;
  // arguments first, before any state of the handle is touched: a mismatch with what xk_build_compress_update_async queued leaves the
  // queued update (posterior in d_Pout, marker, retry bookkeeping) exactly as it was -- the matching call can still collect it
  if (h->fused_pending) {
    bool same = (cov_update ? 1 : 0) == h->fused_cov_update && ct_zero == h->fused_ct_zero;
    if (same && !ct_zero) same = memcmp(corr_total, h->fused_ct->data(), sizeof(double) * h->n) == 0;
    if (!same) return fail(h, XK_EINVAL, "xk_apply_update: not the correction_total / cov_update the queued pass was built with");
  }
  if (corr_total && !ct_zero) {
    double *st = (double *)stage_slot(h, sizeof(double) * h->n);
    if (!st) return fail(h, XK_ECAPACITY, "staging slot too small");
    memcpy(st, corr_total, sizeof(double) * h->n);
    HIPCHK(h, hipMemcpyAsync(h->d_ct, st, sizeof(double) * h->n, hipMemcpyHostToDevice, h->stream));
    dct = h->d_ct;
  }
  const bool deferred = h->compress_deferred;
  const bool async = h->async_pending || deferred, queued = h->fused_pending;
  h->async_pending = false;
  h->fused_pending = false;
  h->compress_deferred = false;
  int rc = XK_OK;
  for (int attempt = 0; attempt < 2; ++attempt) {
    if (attempt == 1) {   // the single-launch CAQR of xk_build_compress_async gave up: rows, compression and update again
      if (deferred) {
        // ... NOT the rows: they were linearised and gated at the prior, which the applyCI entries have replaced since -- and they
        // are still there (the single launch only reads the tiles / factor records; it is the multi-launch schedule that works
        // on the tiles in place, and it has not run)
        h->have_rows = true;
      } else {
        if ((rc = launch_build(h, h->sigma_img)) != XK_OK) return rc;
        if ((rc = cache_flags(h)) != XK_OK) return rc;
      }
      if ((rc = launch_compress(h)) != XK_OK) return rc;
    }
    UpdateSpec u = compressed_spec(h, dct, cov_update);
    u.corr = h->h_out;
    const bool waiting_only = queued && attempt == 0;      // the update is already on the stream (xk_build_compress_update_async)
    // The kernels write the correction and (on failure) the status words into pinned host memory; the last workgroup of the
    // last launch then writes a sequence number next to them, which the host polls: the results are there ~5 us before the
    // runtime's completion signal says so (XK_SPIN_DONE=0: wait for that signal instead).  One wait per update, no copy.
    static const int spin_env = env_int("XK_SPIN_DONE", 1);
    unsigned long long *done = reinterpret_cast<unsigned long long *>(h->h_out + h->n + 2);
    if (waiting_only) u.done_seq = h->fused_seq;
    else {
      if (spin_env) { u.done_flag = done; u.done_seq = ++h->done_seq; }
      if (deferred && attempt == 0) {                      // the compression xk_build_compress_async left for now, Kalman role inside
        if ((rc = launch_compress(h, nullptr, &u)) != XK_OK) return rc;
        rc = h->last_fused ? XK_OK : launch_update(h, u);
      } else rc = launch_update(h, u);
      if (rc != XK_OK) return rc;
    }
    bool seen = false;
    if (spin_env && u.done_seq) {
      // Acquire load: the correction and the status words read below are ordered after the marker.  The marker is the LAST
      // host-visible store of the update.  Separate Kalman launches: a system-scope RELEASE by the last workgroup of the last
      // kernel, after every workgroup of that kernel has been counted in (done_cnt).  Kalman role inside the single launch: the
      // correction and the status word are relaxed system-scope stores whose acknowledgement the writing wave waits for
      // (s_waitcnt vmcnt(0)) before it stores the marker, relaxed as well -- gfx950 behaviour, not a memory-model guarantee; the
      // -DXK_SYNC_STRICT=1 build stores that marker with a system-scope release (XK_MARKER_ORDER, xk_xcd_sync.hip.h).  In both forms
      // the marker covers ONLY the words in h_out: the posterior (d_Pout) may still be on its way out of the other waves when
      // the marker lands and is valid to later work through STREAM ORDER -- everything that reads it is queued on h->stream
      // behind this launch (the pointer swap below is host bookkeeping).  The kernels before it on the stream (whose failure
      // paths write the status words into the same pinned allocation) had completed, their stores released to system scope at
      // their kernel boundaries, before that kernel started: whatever they wrote is visible by the time the marker is.
      // (a single launch that gave up before its Kalman role got going -- placement census -- writes no marker: the status word
      //  ends the wait)
      for (long spins = 0; spins < 40000000L && !(seen = (__atomic_load_n(done, __ATOMIC_ACQUIRE) == u.done_seq)); ++spins) {
        if ((spins & 255) == 255 && __atomic_load_n(&h->d_status[1], __ATOMIC_RELAXED) != 0) break;
        __builtin_ia32_pause();   // (~1 s, then the signal)
      }
      if (seen) h->done_seen = u.done_seq;
    }
    if (!seen) HIPCHK(h, hipStreamSynchronize(h->stream));
    stage_stream_idle(h);
    rc = eval_status(h, h->d_status[0], h->d_status[1], async && attempt == 0);
    if (rc != XK_RETRY_CLASSIC) break;
  }
  if (rc != XK_OK) return rc;
  memcpy(correction, h->h_out, sizeof(double) * h->n);
  std::swap(h->d_P, h->d_Pout);  // posterior becomes the resident covariance
  h->have_rows = h->have_R = false;
  return XK_OK;
}

extern "C" int xk_visual_update_staged(xk_handle *h, double sigma_img, double *correction, int *inlier_msckf,
                                       double *gamma_msckf, int *inlier_slam, double *gamma_slam) {
  if (!h || !correction || !(sigma_img > 0.0)) return XK_EINVAL;
  HIPCHK(h, hipSetDevice(h->device));
  if (h->K == 0 && h->K2 == 0 && h->M == 0) {  // h.size() == 0 -> no update (updater.cpp:106); MSCKF-SLAM rows count (vio_updater.cpp:413-419)
    for (int i = 0; i < h->n; ++i) correction[i] = 0.0;
    return XK_OK;
  }
  int rc = XK_OK;
  for (int attempt = 0; attempt < 2; ++attempt) {
    rc = launch_build(h, sigma_img);
    if (rc != XK_OK) return rc;
    UpdateSpec u = compressed_spec(h, nullptr, 1);
    rc = launch_compress(h, nullptr, &u);            // (the single launch takes the Kalman update along where it can)
    if (rc != XK_OK) return rc;
    if (!h->last_fused) rc = launch_update(h, u);
    if (rc != XK_OK) return rc;
    HIPCHK(h, hipMemcpyAsync(correction, h->d_corr, sizeof(double) * h->n, hipMemcpyDeviceToHost, h->stream));
    rc = fetch_flags(h, inlier_msckf, gamma_msckf, inlier_slam, gamma_slam);
    if (rc != XK_OK) return rc;
    rc = read_status(h, attempt == 0);
    if (rc != XK_RETRY_CLASSIC) break;           // (the prior is untouched in d_P: the whole update is simply redone)
  }
  if (rc != XK_OK) return rc;
  std::swap(h->d_P, h->d_Pout);
  h->have_rows = h->have_R = false;
  return XK_OK;
}

extern "C" int xk_visual_update(xk_handle *h, const double *C_q_G, const double *G_p_C, int n_poses,
                                const int *trk_off, const double *obs_xy, int K, const double *feat,
                                const int *anchor_idxs, const int *track_sizes, const double *z_last, int M,
                                double *P, int ldp, int n, double sigma_img, double *correction,
                                int *inlier_msckf, double *gamma_msckf, int *inlier_slam, double *gamma_slam) {
  int rc;
  if ((rc = xk_stage_window(h, C_q_G, G_p_C, n_poses)) != XK_OK) return rc;
  if ((rc = xk_stage_tracks(h, trk_off, obs_xy, K)) != XK_OK) return rc;
  if ((rc = xk_stage_slam(h, feat, anchor_idxs, track_sizes, z_last, M)) != XK_OK) return rc;
  if ((rc = xk_stage_msckf_slam(h, nullptr, nullptr, 0)) != XK_OK) return rc;   // this entry point has no MSCKF-SLAM tracks
  if ((rc = xk_upload_P(h, P, ldp, n)) != XK_OK) return rc;
  if ((rc = xk_visual_update_staged(h, sigma_img, correction, inlier_msckf, gamma_msckf, inlier_slam,
                                    gamma_slam)) != XK_OK)
    return rc;
  return xk_download_P(h, P, ldp, n);
}

// ---------------------------------------------------------------------------
// dense (unfused) Kalman algebra
// ---------------------------------------------------------------------------
extern "C" int xk_apply_update_dense(xk_handle *h, double *P, int ldp, int n, const double *H, int ldh, int m,
                                     const double *res, const double *r_diag, double *correction_total,
                                     int cov_update, double *correction) {
  if (!h || !P || !H || !res || !r_diag || !correction || n != h->n || ldp < n || ldh < m || m <= 0) return XK_EINVAL;
  if (m > h->CM) return fail(h, XK_ECAPACITY, "m exceeds the dense workspace (n+1 rows); compress first");
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipMemcpy2DAsync(h->d_tmpP, sizeof(double) * n, P, sizeof(double) * ldp, sizeof(double) * n, n,
                             hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, hipMemcpy2DAsync(h->d_tmpH, sizeof(double) * m, H, sizeof(double) * ldh, sizeof(double) * m, n,
                             hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, hipMemcpyAsync(h->d_tmpz, res, sizeof(double) * m, hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, hipMemcpyAsync(h->d_rdiag, r_diag, sizeof(double) * m, hipMemcpyHostToDevice, h->stream));
  const double *dct = nullptr;
  if (correction_total) {
    HIPCHK(h, hipMemcpyAsync(h->d_ct, correction_total, sizeof(double) * n, hipMemcpyHostToDevice, h->stream));
    dct = h->d_ct;
  }
  UpdateSpec u;
  memset(&u, 0, sizeof(u));
  u.T = h->d_tmpH; u.str = 1; u.stc = m;  // column-major m x n
  u.c = m; u.kdim = n; u.col0 = 0;
  u.z = h->d_tmpz; u.sz = 1; u.rdiag = h->d_rdiag;
  u.Pin = h->d_tmpP; u.Pout = h->d_Pout; u.ct = dct; u.cov_update = cov_update;
  int rc = launch_update(h, u);
  if (rc != XK_OK) return rc;
  HIPCHK(h, hipMemcpyAsync(correction, h->d_corr, sizeof(double) * n, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipMemcpy2DAsync(P, sizeof(double) * ldp, h->d_Pout, sizeof(double) * n, sizeof(double) * n, n,
                             hipMemcpyDeviceToHost, h->stream));
  rc = read_status(h);
  if (rc != XK_OK) return rc;
  if (correction_total)
    for (int i = 0; i < n; ++i) correction_total[i] += correction[i];  // updater.cpp:140
  return XK_OK;
}

extern "C" int xk_apply_ci(xk_handle *h, double *P_out, int ldp, const double *ci_P, int ldc, int n,
                           const double *H, int ldh, int m, const double *res, const double *S, int lds,
                           double *correction) {
  if (!h || !P_out || !ci_P || !H || !res || !S || !correction || n != h->n || ldp < n || ldc < n || ldh < m ||
      lds < m || m <= 0)
    return XK_EINVAL;
  if (m > h->CM) return fail(h, XK_ECAPACITY, "m exceeds the dense workspace");
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipMemcpy2DAsync(h->d_tmpP, sizeof(double) * n, ci_P, sizeof(double) * ldc, sizeof(double) * n, n,
                             hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, hipMemcpy2DAsync(h->d_tmpH, sizeof(double) * m, H, sizeof(double) * ldh, sizeof(double) * m, n,
                             hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, hipMemcpy2DAsync(h->d_tmpS, sizeof(double) * m, S, sizeof(double) * lds, sizeof(double) * m, m,
                             hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, hipMemcpyAsync(h->d_tmpz, res, sizeof(double) * m, hipMemcpyHostToDevice, h->stream));
  UpdateSpec u;
  memset(&u, 0, sizeof(u));
  u.T = h->d_tmpH; u.str = 1; u.stc = m;
  u.c = m; u.kdim = n; u.col0 = 0;
  u.z = h->d_tmpz; u.sz = 1;
  u.S = h->d_tmpS; u.ssr = 1; u.ssc = m;
  u.Pin = h->d_tmpP; u.Pout = h->d_Pout; u.ct = nullptr; u.cov_update = 1;
  int rc = launch_update(h, u);
  if (rc != XK_OK) return rc;
  HIPCHK(h, hipMemcpyAsync(correction, h->d_corr, sizeof(double) * n, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipMemcpy2DAsync(P_out, sizeof(double) * ldp, h->d_Pout, sizeof(double) * n, sizeof(double) * n, n,
                             hipMemcpyDeviceToHost, h->stream));
  return read_status(h);
}

// applyCI on the RESIDENT covariance: P <- sym((I - K H) ci_P) replaces the handle's covariance and stays on the
// device (the host mirror's resident mode; the compressed [T_H | z] of a pending xk_apply_update is not touched).
// A build + compression queued by xk_build_compress_async whose single-launch CAQR gave up must be redone (multi-launch
// schedule) while the covariance it was linearised at is still the resident one: BEFORE a CI entry replaces it.  After this
// the compressed [T_H | z] is known to be good and xk_apply_update has nothing to retry.
static int settle_async(xk_handle *h) {
  if (!h->async_pending) return XK_OK;
  HIPCHK(h, hipStreamSynchronize(h->stream));
  stage_stream_idle(h);
  int rc = eval_status(h, h->d_status[0], h->d_status[1], true);
  if (rc == XK_RETRY_CLASSIC) {
    if ((rc = launch_build(h, h->sigma_img)) != XK_OK) return rc;
    if ((rc = cache_flags(h)) != XK_OK) return rc;
    if ((rc = launch_compress(h)) != XK_OK) return rc;
  }
  if (rc == XK_OK) h->async_pending = false;
  return rc;
}

extern "C" int xk_apply_ci_resident(xk_handle *h, const double *ci_P, int ldc, int n, const double *H, int ldh, int m,
                                    const double *res, const double *S, int lds, double *correction) {
  if (!h || !ci_P || !H || !res || !S || !correction || n != h->n || ldc < n || ldh < m || lds < m || m <= 0) return XK_EINVAL;
  if (m > h->CM) return fail(h, XK_ECAPACITY, "m exceeds the dense workspace");
  HIPCHK(h, hipSetDevice(h->device));
  {
    const int rs = settle_async(h);
    if (rs != XK_OK) return rs;
  }
  HIPCHK(h, hipMemcpy2DAsync(h->d_tmpP, sizeof(double) * n, ci_P, sizeof(double) * ldc, sizeof(double) * n, n,
                             hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, hipMemcpy2DAsync(h->d_tmpH, sizeof(double) * m, H, sizeof(double) * ldh, sizeof(double) * m, n,
                             hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, hipMemcpy2DAsync(h->d_tmpS, sizeof(double) * m, S, sizeof(double) * lds, sizeof(double) * m, m,
                             hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, hipMemcpyAsync(h->d_tmpz, res, sizeof(double) * m, hipMemcpyHostToDevice, h->stream));
  UpdateSpec u;
  memset(&u, 0, sizeof(u));
  u.T = h->d_tmpH; u.str = 1; u.stc = m;
  u.c = m; u.kdim = n; u.col0 = 0;
  u.z = h->d_tmpz; u.sz = 1;
  u.S = h->d_tmpS; u.ssr = 1; u.ssc = m;
  u.Pin = h->d_tmpP; u.Pout = h->d_Pout; u.ct = nullptr; u.cov_update = 1;
  int rc = launch_update(h, u);
  if (rc != XK_OK) return rc;
  HIPCHK(h, hipMemcpyAsync(correction, h->d_corr, sizeof(double) * n, hipMemcpyDeviceToHost, h->stream));
  rc = read_status(h);
  if (rc != XK_OK) return rc;
  std::swap(h->d_P, h->d_Pout);
  return XK_OK;
}

// Keeps / brings back a copy of the resident covariance on the device (restore = 0: save, 1: restore).  Benchmarks
// use it to replay a frame from the same prior without a PCIe upload.
extern "C" int xk_snapshot_P(xk_handle *h, int restore) {
  if (!h || restore < 0 || restore > 3) return XK_EINVAL;
  HIPCHK(h, hipSetDevice(h->device));
  const size_t bytes = sizeof(double) * (size_t)h->n * h->n;
  double *&slot = (restore >= 2) ? h->d_Psnap2 : h->d_Psnap;     // 0 / 1: the caller's slot; 2 / 3: the filter loop's own (x::Ekf)
  const bool back = restore & 1;
  if (!slot) {
    if (back) return fail(h, XK_EINVAL, "xk_snapshot_P: nothing saved");
    HIPCHK(h, dalloc(&slot, (size_t)h->n * h->n));
  }
  HIPCHK(h, hipMemcpyAsync(back ? h->d_P : slot, back ? slot : h->d_P, bytes, hipMemcpyDeviceToDevice, h->stream));
  return XK_OK;
}

// ---------------------------------------------------------------------------
// measurement
// ---------------------------------------------------------------------------
extern "C" int xk_bench_staged(xk_handle *h, double sigma_img, int warmup, int steps, xk_timing *out) {
  if (!h || !out || steps <= 0 || warmup < 0) return XK_EINVAL;
  HIPCHK(h, hipSetDevice(h->device));
  memset(out, 0, sizeof(*out));
  const char *names[XK_NSTAGE] = {"xk_msckf_feature", "xk_slam_rows", "xk_caqr_panel0",
                                  "xk_caqr_rest",
                                  "xk_kalman_update", "(unused)"};
  for (int s = 0; s < XK_NSTAGE; ++s) snprintf(out->stage_name[s], sizeof(out->stage_name[s]), "%s", names[s]);
  double acc[XK_NSTAGE] = {0, 0, 0, 0, 0, 0}, tot = 0;
  for (int attempt = 0; attempt < 2; ++attempt) {
  for (int s = 0; s < XK_NSTAGE; ++s) acc[s] = 0;
  tot = 0;
  for (int it = 0; it < warmup + steps; ++it) {
    HIPCHK(h, hipEventRecord(h->ev[0], h->stream));
    int rc = launch_build(h, sigma_img);
    if (rc != XK_OK) return rc;
    HIPCHK(h, hipEventRecord(h->ev[1], h->stream));
    UpdateSpec u = compressed_spec(h, nullptr, 1);
    rc = launch_compress(h, h->ev[2], &u);
    if (rc != XK_OK) return rc;
    HIPCHK(h, hipEventRecord(h->ev[3], h->stream));
    if (!h->last_fused) rc = launch_update(h, u);     // (fused: the Kalman update is inside stage 2's launch, stage 4 reads 0)
    if (rc != XK_OK) return rc;
    HIPCHK(h, hipEventRecord(h->ev[4], h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (it >= warmup) {
      float ms;
      hipEventElapsedTime(&ms, h->ev[0], h->ev[1]); acc[0] += ms;
      hipEventElapsedTime(&ms, h->ev[1], h->ev[2]); acc[2] += ms;
      hipEventElapsedTime(&ms, h->ev[2], h->ev[3]); acc[3] += ms;
      hipEventElapsedTime(&ms, h->ev[3], h->ev[4]); acc[4] += ms;
      hipEventElapsedTime(&ms, h->ev[0], h->ev[4]); tot += ms;
    }
  }
  const int rc = read_status(h, attempt == 0);
  if (rc == XK_RETRY_CLASSIC) continue;            // the single-launch CAQR gave up somewhere: measure the multi-launch schedule
  if (rc != XK_OK) return rc;
  break;
  }
  for (int s = 0; s < XK_NSTAGE; ++s) out->stage_ms[s] = (float)(acc[s] / steps);
  out->total_ms = (float)(tot / steps);
  out->stage_launches[0] = (h->K > 0) + (h->M > 0);
  out->stage_launches[2] = 1;
  out->stage_launches[3] = h->last_resident ? 0 : h->nlevels;   // (the resident schedule IS stage 2's one launch)
  const int nblk = (h->na + XK_CHOL_NB - 1) / XK_CHOL_NB;
  out->stage_launches[4] = h->last_fused ? 0 : 4 + 3 * nblk;
  out->n = h->n; out->c1 = h->C1; out->k_tracks = h->K; out->n_leaf = h->nleaf; out->n_levels = h->nlevels;
  // stacked rows actually folded (inlier rows)
  {
    const int slam_tiles = (2 * h->M + h->DB - 1) / h->DB;
    std::vector<int> tr(h->K + h->K2 + slam_tiles);
    HIPCHK(h, hipMemcpy(tr.data(), h->d_tile_rows, sizeof(int) * tr.size(), hipMemcpyDeviceToHost));
    int rows = 0;
    for (int v : tr) rows += v;
    out->rows_stacked = rows;
  }
  // leave the posterior of the last step in d_Pout; the staged prior stays in d_P
  return XK_OK;
}

// ---------------------------------------------------------------------------
// covariance intersection (fixed weights)
// ---------------------------------------------------------------------------
static int check_w(double w) {  // ci.cpp:59-62,98-101 throw; -1<=w<0 is the NLopt branch (out of scope)
  if (w > 1.0 || w == 0 || w < -1) return XK_EINVAL;
  if (w < 0.0) return XK_EINVAL;
  return XK_OK;
}

// S (device, m x m row-major ld m in d_tmpS) (+)= alpha * Hd (m x nn col-major) * Pd (nn x nn) * Hd^T
static void hpht_accum(xk_handle *h, const double *Hd, const double *Pd, int m, int nn, double alpha, bool first) {
  XkGemmArgs g;
  memset(&g, 0, sizeof(g));
  g.A = Hd; g.sar = 1; g.sac = m;
  g.B = Pd; g.sbr = 1; g.sbc = nn;
  g.C = h->d_Maug; g.scr = nn; g.scc = 1; g.D = g.C; g.sdr = nn; g.sdc = 1;
  g.M = m; g.N = nn; g.K = nn; g.alpha = 1.0; g.beta = 0.0;
  gemm(h, g);  // W = H P  (m x nn, row-major ld nn)
  memset(&g, 0, sizeof(g));
  g.A = h->d_Maug; g.sar = nn; g.sac = 1;
  g.B = Hd; g.sbr = m; g.sbc = 1;  // B[k][j] = H[j][k]
  g.C = h->d_tmpS; g.scr = 1; g.scc = m; g.D = g.C; g.sdr = 1; g.sdc = m;
  g.M = m; g.N = m; g.K = nn; g.alpha = alpha; g.beta = first ? 0.0 : 1.0;
  gemm(h, g);
}

extern "C" int xk_fuse_ci_msckf(xk_handle *h, const double *P, int ldp, int n, const double *H, int ldh, int m,
                                int k, const double *const *Ps, const int *ns, const double *const *Hs,
                                double w_other, double *S, int lds, double *w_result) {
  if (!h || !P || !H || !S || !w_result || k < 0 || m <= 0 || ldp < n || ldh < m || lds < m) return XK_EINVAL;
  if (check_w(w_other) != XK_OK) return fail(h, XK_EINVAL, "The CI weights must be lower than 1.0 and larger 0.0");
  if (n > h->n || m > h->CM) return fail(h, XK_ECAPACITY, "fuse_ci dims exceed workspace");
  HIPCHK(h, hipSetDevice(h->device));
  const double w0 = 1.0 - (double)k * w_other;
  HIPCHK(h, hipMemcpy2DAsync(h->d_tmpP, sizeof(double) * n, P, sizeof(double) * ldp, sizeof(double) * n, n,
                             hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, hipMemcpy2DAsync(h->d_tmpH, sizeof(double) * m, H, sizeof(double) * ldh, sizeof(double) * m, n,
                             hipMemcpyHostToDevice, h->stream));
  hpht_accum(h, h->d_tmpH, h->d_tmpP, m, n, 1.0 / w0, true);
  for (int i = 0; i < k; ++i) {
    if (ns[i] > h->n) return fail(h, XK_ECAPACITY, "other agent's state larger than workspace");
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipMemcpyAsync(h->d_tmpP, Ps[i], sizeof(double) * (size_t)ns[i] * ns[i], hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->d_tmpH, Hs[i], sizeof(double) * (size_t)m * ns[i], hipMemcpyHostToDevice, h->stream));
    hpht_accum(h, h->d_tmpH, h->d_tmpP, m, ns[i], 1.0 / w_other, false);
  }
  HIPCHK(h, hipMemcpy2DAsync(S, sizeof(double) * lds, h->d_tmpS, sizeof(double) * m, sizeof(double) * m, m,
                             hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  *w_result = 1.0 / w0;
  return XK_OK;
}

extern "C" int xk_fuse_ci_slam(xk_handle *h, const double *Pa, int lda, int na, const double *Ha, int ldha,
                               const double *Pb, int ldb, int nb, const double *Hb, int ldhb, int m,
                               double w_other, double *S, int lds, double *w_result) {
  if (!h || !Pa || !Ha || !Pb || !Hb || !S || !w_result || m <= 0 || lda < na || ldb < nb || ldha < m || ldhb < m ||
      lds < m)
    return XK_EINVAL;
  if (check_w(w_other) != XK_OK)
    return fail(h, XK_EINVAL, "The CI weights must be lower than 1.0 and larger than 0.0");
  if (na > h->n || nb > h->n || m > h->CM) return fail(h, XK_ECAPACITY, "fuse_ci dims exceed workspace");
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipMemcpy2DAsync(h->d_tmpP, sizeof(double) * na, Pa, sizeof(double) * lda, sizeof(double) * na, na,
                             hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, hipMemcpy2DAsync(h->d_tmpH, sizeof(double) * m, Ha, sizeof(double) * ldha, sizeof(double) * m, na,
                             hipMemcpyHostToDevice, h->stream));
  hpht_accum(h, h->d_tmpH, h->d_tmpP, m, na, 1.0 / (1.0 - w_other), true);
  HIPCHK(h, hipStreamSynchronize(h->stream));
  HIPCHK(h, hipMemcpy2DAsync(h->d_tmpP, sizeof(double) * nb, Pb, sizeof(double) * ldb, sizeof(double) * nb, nb,
                             hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, hipMemcpy2DAsync(h->d_tmpH, sizeof(double) * m, Hb, sizeof(double) * ldhb, sizeof(double) * m, nb,
                             hipMemcpyHostToDevice, h->stream));
  hpht_accum(h, h->d_tmpH, h->d_tmpP, m, nb, 1.0 / w_other, false);
  HIPCHK(h, hipMemcpy2DAsync(S, sizeof(double) * lds, h->d_tmpS, sizeof(double) * m, sizeof(double) * m, m,
                             hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  *w_result = 1.0 / (1.0 - w_other);
  return XK_OK;
}

extern "C" int xk_multi_slam_match(xk_handle *h, const double *C_q_G, const double *G_p_C, int n_poses,
                                   const double *feat, int anchor_idx, int feature_id, const double *P, int ldp,
                                   int n, int n_poses_max, const double *o_C_q_G, const double *o_G_p_C,
                                   int o_n_poses, const double *o_feat, int o_anchor_idx, int o_feature_id,
                                   const double *o_P, int ldop, int no, int o_n_poses_max, double sigma_landmark,
                                   double ci_slam_w, int *inlier, double *gamma, double *H, int ldh, double *res,
                                   double *S, double *P_j, int ldpj) {
  if (!h || !C_q_G || !G_p_C || !feat || !P || !o_C_q_G || !o_G_p_C || !o_feat || !o_P || !inlier || !gamma || !H ||
      !res || !S || !P_j)
    return XK_EINVAL;
  if (anchor_idx < 0) return fail(h, XK_EINVAL, "anchor_idx < 0");       // throws, multi_slam_update.cpp:83-85
  if (n != h->n || ldp < n || ldop < no || ldh < 3 || ldpj < n) return XK_EINVAL;
  const int Mf = (n - XK_CORE - 6 * n_poses_max) / 3, oMf = (no - XK_CORE - 6 * o_n_poses_max) / 3;
  if (feature_id < 0 || feature_id >= Mf || o_feature_id < 0 || o_feature_id >= oMf || anchor_idx >= n_poses ||
      o_anchor_idx < 0 || o_anchor_idx >= o_n_poses)
    return XK_EINVAL;
  if (feat[3 * feature_id + 2] == 0) return fail(h, XK_EINVAL, "rho = 0");  // throws, :86-88
  if (check_w(ci_slam_w) != XK_OK)
    return fail(h, XK_EINVAL, "The CI weights must be lower than 1.0 and larger than 0.0");
  if (no > h->n || n_poses > h->N || o_n_poses > h->N) return fail(h, XK_ECAPACITY, "match dims exceed workspace");
  HIPCHK(h, hipSetDevice(h->device));
  // scratch layout in d_ci
  double *d = h->d_ci;
  double *dq = d, *dp = dq + 4 * h->N, *df = dp + 3 * h->N, *doq = df + 3 * (Mf > 0 ? Mf : 1);
  double *dop = doq + 4 * h->N, *dof = dop + 3 * h->N, *dH = dof + 3 * (oMf > 0 ? oMf : 1);
  double *dout = dH + 3 * (size_t)n, *dcols_f = dout + 16;
  int *dcols = (int *)dcols_f;
  HIPCHK(h, hipMemcpyAsync(dq, C_q_G, sizeof(double) * 4 * n_poses, hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, hipMemcpyAsync(dp, G_p_C, sizeof(double) * 3 * n_poses, hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, hipMemcpyAsync(df, feat, sizeof(double) * 3 * Mf, hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, hipMemcpyAsync(doq, o_C_q_G, sizeof(double) * 4 * o_n_poses, hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, hipMemcpyAsync(dop, o_G_p_C, sizeof(double) * 3 * o_n_poses, hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, hipMemcpyAsync(dof, o_feat, sizeof(double) * 3 * oMf, hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, hipMemcpy2DAsync(h->d_tmpP, sizeof(double) * n, P, sizeof(double) * ldp, sizeof(double) * n, n,
                             hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, hipMemcpy2DAsync(h->d_Maug, sizeof(double) * no, o_P, sizeof(double) * ldop, sizeof(double) * no, no,
                             hipMemcpyHostToDevice, h->stream));
  XkSlamMatchArgs a;
  a.q = dq; a.p = dp; a.feat = df; a.P = h->d_tmpP; a.anchor = anchor_idx; a.fid = feature_id; a.n = n; a.npm = n_poses_max;
  a.oq = doq; a.op = dop; a.ofeat = dof; a.oP = h->d_Maug; a.oanchor = o_anchor_idx; a.ofid = o_feature_id; a.no = no;
  a.onpm = o_n_poses_max;
  a.var_l = sigma_landmark * sigma_landmark; a.w = ci_slam_w; a.chi = XK_CHI2_090[3];
  a.H = dH; a.out = dout; a.cols = dcols;
  hipLaunchKernelGGL(xk_slam_match, dim3(1), dim3(64), 0, h->stream, a);
  XkScaleArgs sc{h->d_tmpP, h->d_Pout, n, 3, dcols, dout + 14};
  hipLaunchKernelGGL(xk_scale_blocks, dim3(((size_t)n * n + 255) / 256), dim3(256), 0, h->stream, sc);
  double hout[16];
  HIPCHK(h, hipMemcpyAsync(hout, dout, sizeof(double) * 16, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipMemcpy2DAsync(H, sizeof(double) * ldh, dH, sizeof(double) * 3, sizeof(double) * 3, n,
                             hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  for (int i = 0; i < 3; ++i) res[i] = hout[i];
  *gamma = hout[12];
  *inlier = hout[13] != 0.0;
  if (*inlier) {
    for (int i = 0; i < 9; ++i) S[i] = hout[3 + i];
    HIPCHK(h, hipMemcpy2DAsync(P_j, sizeof(double) * ldpj, h->d_Pout, sizeof(double) * n, sizeof(double) * n, n,
                               hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
  }
  return XK_OK;
}

// ---------------------------------------------------------------------------
// MSCKF-MSCKF CI block (msckf_update.cpp:96-279) for one track
// ---------------------------------------------------------------------------
extern "C" int xk_msckf_ci_track(xk_handle *h, const double *obs, int L, const double *C_q_G, const double *G_p_C,
                                 int n_poses, const double *P, int ldp, int n, int n_poses_max, double sigma_img,
                                 int k, const double *const *m_obs, const int *m_L, const double *const *m_q,
                                 const double *const *m_p, const int *m_nposes, const double *const *m_P,
                                 const int *m_n, double ci_msckf_w, int *self_inlier, double *self_gamma,
                                 int *has_ci, double *ci_gamma, double *H, int ldh, double *res, double *S, int lds,
                                 double *P_j, int ldpj) {
  if (!h || !obs || !C_q_G || !G_p_C || !P || !self_inlier || !self_gamma || !has_ci || k < 0) return XK_EINVAL;
  if (k > XK_CI_MAXK) return fail(h, XK_ECAPACITY, "more than 7 matched agents");
  if (n != h->n || ldp < n || L < 2 || L > n_poses || n_poses > h->N || n_poses_max != h->N) return XK_EINVAL;
  if (k > 0 && (!m_obs || !m_L || !m_q || !m_p || !m_nposes || !m_P || !m_n || !H || !res || !S || !P_j || !ci_gamma ||
                ldh < 3 * k || lds < 3 * k || ldpj < n))
    return XK_EINVAL;
  if (k > 0 && check_w(ci_msckf_w) != XK_OK)
    return fail(h, XK_EINVAL, "The CI weights must be lower than 1.0 and larger 0.0");
  int Ltot = L, nmax = n;
  for (int i = 0; i < k; ++i) {
    if (m_L[i] < 2 || m_L[i] > m_nposes[i] || m_nposes[i] > 64 || m_n[i] < XK_CORE + 6 * m_nposes[i]) return XK_EINVAL;
    Ltot += m_L[i];
    nmax = std::max(nmax, m_n[i]);
  }
  HIPCHK(h, hipSetDevice(h->device));
  *has_ci = 0;
  const int m = 3 * k, k1 = k + 1;
  // workspace (lazily sized): concatenated lists, per-agent window/obs/P, up rows, H blocks, S buffers
  const size_t upsz = 3 * (size_t)nmax + 16;
  const size_t need = 9 * (size_t)Ltot + 7 * 64 + 2 * 64 + (size_t)nmax * nmax + k1 * upsz + (size_t)std::max(m, 1) * k1 * nmax +
                      4 * 24 * 24 + 1024;
  double *ws = nullptr;
  HIPCHK(h, hipMalloc((void **)&ws, sizeof(double) * need));
  struct Guard { double *p; ~Guard() { if (p) hipFree(p); } } guard{ws};
  double *dq = ws, *dp = dq + 4 * (size_t)Ltot, *dobs = dp + 3 * (size_t)Ltot;
  double *aq = dobs + 2 * (size_t)Ltot, *ap = aq + 4 * 64, *aobs = ap + 3 * 64;   // one agent's window + track
  double *aP = aobs + 2 * 64, *up = aP + (size_t)nmax * nmax, *Hs = up + k1 * upsz;
  double *S1 = Hs + (size_t)std::max(m, 1) * k1 * nmax, *S2 = S1 + 24 * 24, *dres = S2 + 24 * 24, *dgpf = dres + 24;
  double *dscal = dgpf + 8;  // [0] ci gamma, [1] w_result
  int *dint = (int *)(dscal + 8);  // [0] gn iters, [1] inlier, [2] tile rows, [3..] block columns
  // concatenated lists: matched agents first, self last (:113-149)
  size_t at = 0;
  for (int i = 0; i < k; ++i) {
    HIPCHK(h, hipMemcpyAsync(dq + 4 * at, m_q[i] + 4 * (size_t)(m_nposes[i] - m_L[i]), sizeof(double) * 4 * m_L[i], hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(dp + 3 * at, m_p[i] + 3 * (size_t)(m_nposes[i] - m_L[i]), sizeof(double) * 3 * m_L[i], hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(dobs + 2 * at, m_obs[i], sizeof(double) * 2 * m_L[i], hipMemcpyHostToDevice, h->stream));
    at += m_L[i];
  }
  HIPCHK(h, hipMemcpyAsync(dq + 4 * at, C_q_G + 4 * (size_t)(n_poses - L), sizeof(double) * 4 * L, hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, hipMemcpyAsync(dp + 3 * at, G_p_C + 3 * (size_t)(n_poses - L), sizeof(double) * 3 * L, hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, hipMemcpyAsync(dobs + 2 * at, obs, sizeof(double) * 2 * L, hipMemcpyHostToDevice, h->stream));
  XkTriMultiArgs ta{dq, dp, dobs, Ltot, dgpf, dint, nullptr, 0};
  hipLaunchKernelGGL(xk_triangulate_multi, dim3(1), dim3(64), 0, h->stream, ta);
  // per-agent column-space rows (self first = row block 0, then the matched agents, :168-204)
  const int offs[2] = {0, 0};
  (void)offs;
  for (int i = 0; i < k1; ++i) {
    const bool self = (i == 0);
    const double *hq = self ? C_q_G : m_q[i - 1], *hp = self ? G_p_C : m_p[i - 1], *hobs = self ? obs : m_obs[i - 1];
    const double *hP = self ? P : m_P[i - 1];
    const int np_i = self ? n_poses : m_nposes[i - 1], L_i = self ? L : m_L[i - 1], n_i = self ? n : m_n[i - 1];
    const int ld_i = self ? ldp : n_i, npm_i = self ? n_poses_max : m_nposes[i - 1];
    HIPCHK(h, hipStreamSynchronize(h->stream));  // scratch reuse across agents
    HIPCHK(h, hipMemcpyAsync(aq, hq, sizeof(double) * 4 * np_i, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(ap, hp, sizeof(double) * 3 * np_i, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(aobs, hobs, sizeof(double) * 2 * L_i, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpy2DAsync(aP, sizeof(double) * n_i, hP, sizeof(double) * ld_i, sizeof(double) * n_i, n_i, hipMemcpyHostToDevice, h->stream));
    const int toff[2] = {0, L_i};
    HIPCHK(h, hipMemcpyAsync(dint + 8, toff, sizeof(int) * 2, hipMemcpyHostToDevice, h->stream));
    XkFeatArgs a;
    memset(&a, 0, sizeof(a));
    a.q = aq; a.p = ap; a.n_poses = np_i; a.n_poses_max = npm_i; a.trk_off = dint + 8; a.obs = aobs; a.K = 1;
    a.P = aP; a.n = n_i; a.var_img = sigma_img * sigma_img; a.chi95 = h->d_chi95;
    a.A = nullptr; a.DB = 0; a.C1P = 0; a.na = n_i - XK_CORE;
    a.tile_rows = dint + 2; a.inlier = dint + 1; a.gamma = dscal + 2; a.gpf = dgpf + 4; a.gn_iters = dint + 3;
    a.gpf_in = dgpf; a.up_out = up + i * upsz; a.batch = nullptr; a.dbg = nullptr; a.inlier_h = nullptr; a.gamma_h = nullptr;
    hipLaunchKernelGGL(xk_msckf_feature, dim3(1), dim3(XK_FEAT_THREADS), xk_feature_lds_bytes(np_i), h->stream, a);
    if (self) {
      int inl = 0;
      double g = 0;
      HIPCHK(h, hipMemcpyAsync(&inl, dint + 1, sizeof(int), hipMemcpyDeviceToHost, h->stream));
      HIPCHK(h, hipMemcpyAsync(&g, dscal + 2, sizeof(double), hipMemcpyDeviceToHost, h->stream));
      HIPCHK(h, hipStreamSynchronize(h->stream));
      *self_inlier = inl;
      *self_gamma = g;
      if (!inl || k == 0) return XK_OK;  // proceed_with_multi_ false (:180)
    }
  }
  // null-space projection of the landmark and split into per-agent Jacobians (:207-223)
  XkCiProjArgs pa;
  memset(&pa, 0, sizeof(pa));
  pa.k1 = k1;
  pa.res = dres;
  for (int i = 0; i < k1; ++i) {
    pa.up[i] = up + i * upsz;
    pa.n[i] = (i == 0) ? n : m_n[i - 1];
    pa.H[i] = Hs + (size_t)m * nmax * i;
  }
  hipLaunchKernelGGL(xk_ci_project, dim3(k1), dim3(256), 0, h->stream, pa);
  // S_gate = sum H_i P_i H_i^T + sigma^2 I (:217-237) and the CI-weighted S (ci.cpp:78-85 + :255)
  const double w0 = 1.0 - (double)k * ci_msckf_w, var_img = sigma_img * sigma_img;
  for (int i = 0; i < k1; ++i) {
    const double *hP = (i == 0) ? P : m_P[i - 1];
    const int n_i = (i == 0) ? n : m_n[i - 1], ld_i = (i == 0) ? ldp : n_i;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipMemcpy2DAsync(aP, sizeof(double) * n_i, hP, sizeof(double) * ld_i, sizeof(double) * n_i, n_i, hipMemcpyHostToDevice, h->stream));
    XkGemmArgs g;
    memset(&g, 0, sizeof(g));
    g.A = pa.H[i]; g.sar = 1; g.sac = m; g.B = aP; g.sbr = 1; g.sbc = n_i;
    g.C = h->d_Maug; g.scr = n_i; g.scc = 1; g.D = g.C; g.sdr = n_i; g.sdc = 1;
    g.M = m; g.N = n_i; g.K = n_i; g.alpha = 1.0; g.beta = 0.0;
    gemm(h, g);  // W = H_i P_i
    for (int pass = 0; pass < 2; ++pass) {
      memset(&g, 0, sizeof(g));
      g.A = h->d_Maug; g.sar = n_i; g.sac = 1; g.B = pa.H[i]; g.sbr = m; g.sbc = 1;
      g.C = pass ? S2 : S1; g.scr = 1; g.scc = m; g.D = g.C; g.sdr = 1; g.sdc = m;
      g.M = m; g.N = m; g.K = n_i;
      g.alpha = pass ? (i == 0 ? 1.0 / w0 : 1.0 / ci_msckf_w) : 1.0;
      g.beta = (i == 0) ? 0.0 : 1.0;
      g.mode = (i == k) ? 1 : 0;            // noise once, on the last term
      g.diag = nullptr; g.diag_scalar = var_img;
      gemm(h, g);
    }
  }
  hipLaunchKernelGGL(xk_small_gamma, dim3(1), dim3(1), 0, h->stream, S1, dres, m, dscal);
  double g_ci = 0;
  HIPCHK(h, hipMemcpyAsync(&g_ci, dscal, sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  *ci_gamma = g_ci;
  const int dof = 2 * Ltot - 3;
  if (dof >= XK_CHI2_LEN) return fail(h, XK_ECAPACITY, "chi-square table too short");
  if (!(g_ci < XK_CHI2_095[dof])) return XK_OK;   // :243-250
  // P_j: diagonal 3x3 blocks of the L observed poses scaled by w_result = 1/w0 (:256-267)
  std::vector<int> cols(2 * L);
  for (int i = 0; i < L; ++i) {
    const int pos = n_poses - L + i;
    cols[2 * i] = XK_CORE + 3 * pos;
    cols[2 * i + 1] = XK_CORE + 3 * pos + 3 * n_poses_max;
  }
  const double wres = 1.0 / w0;
  int *dcols = dint + 16;
  HIPCHK(h, hipMemcpyAsync(dcols, cols.data(), sizeof(int) * 2 * L, hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, hipMemcpyAsync(dscal + 1, &wres, sizeof(double), hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, hipMemcpy2DAsync(h->d_tmpP, sizeof(double) * n, P, sizeof(double) * ldp, sizeof(double) * n, n, hipMemcpyHostToDevice, h->stream));
  XkScaleArgs sc{h->d_tmpP, h->d_Pout, n, 2 * L, dcols, dscal + 1};
  hipLaunchKernelGGL(xk_scale_blocks, dim3(((size_t)n * n + 255) / 256), dim3(256), 0, h->stream, sc);
  HIPCHK(h, hipMemcpy2DAsync(H, sizeof(double) * ldh, pa.H[0], sizeof(double) * m, sizeof(double) * m, n, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipMemcpyAsync(res, dres, sizeof(double) * m, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipMemcpy2DAsync(S, sizeof(double) * lds, S2, sizeof(double) * m, sizeof(double) * m, m, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipMemcpy2DAsync(P_j, sizeof(double) * ldpj, h->d_Pout, sizeof(double) * n, sizeof(double) * n, n, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  *has_ci = 1;
  return XK_OK;
}

// ---------------------------------------------------------------------------
// StateManager::manage on the resident covariance
// ---------------------------------------------------------------------------
static int congruence(xk_handle *h, const int *row_ptr, const int *col_idx, const double *val, int nnz, const double *q,
                      int qdim, int qoff) {
  const int n = h->n;
  HIPCHK(h, hipSetDevice(h->device));
  // the operand goes through the pinned ring: nothing here waits for the device (a frame applies two or three of these
  // back to back -- IMU steps, manage() -- before the update's one synchronisation)
  // (+ the window lists staged since the last launch that needed them: they ride in this copy and the kernel leaves them in d_q)
  const int wn = h->win_pending ? 7 * h->n_poses : 0;
  const size_t vb = sizeof(double) * ((size_t)nnz + (q ? (size_t)qdim * qdim : 0) + wn), ib = sizeof(int) * ((size_t)n + 1 + nnz);
  char *st = stage_slot(h, vb + ib);
  if (!st) return fail(h, XK_ECAPACITY, "sparse operand exceeds the staging slot");
  double *sv = (double *)st;
  int *si = (int *)(st + vb);
  if (nnz) memcpy(sv, val, sizeof(double) * nnz);
  if (q) memcpy(sv + nnz, q, sizeof(double) * (size_t)qdim * qdim);
  const size_t woff = (size_t)nnz + (q ? (size_t)qdim * qdim : 0);
  if (wn) memcpy(sv + woff, h->h_win, sizeof(double) * wn);
  memcpy(si, row_ptr, sizeof(int) * (n + 1));
  if (nnz) memcpy(si + n + 1, col_idx, sizeof(int) * nnz);
  // [values | additive block | row pointers | column indices] in one copy
  HIPCHK(h, hipMemcpyAsync(h->d_csr_v, st, vb + ib, hipMemcpyHostToDevice, h->stream));
  double *dq = q ? h->d_csr_v + nnz : nullptr;
  const int *d_rp = (const int *)((const char *)h->d_csr_v + vb);
  XkCongArgs a{h->d_P, h->d_Pout, n, d_rp, d_rp + n + 1, h->d_csr_v, dq, qdim, qoff, wn ? h->d_csr_v + woff : nullptr, h->d_q, wn};
  h->win_pending = false;
  hipLaunchKernelGGL(xk_congruence, dim3(((size_t)n * n + 255) / 256), dim3(256), 0, h->stream, a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(h, XK_EDEVICE, "congruence launch", e);
  std::swap(h->d_P, h->d_Pout);
  h->have_rows = h->have_R = false;
  return XK_OK;
}

extern "C" int xk_cov_congruence(xk_handle *h, const int *row_ptr, const int *col_idx, const double *val, int nnz) {
  if (!h || !row_ptr || nnz < 0 || (nnz > 0 && (!col_idx || !val))) return XK_EINVAL;
  const int n = h->n;
  if (row_ptr[0] != 0 || row_ptr[n] != nnz) return fail(h, XK_EINVAL, "CSR row pointers inconsistent with nnz");
  if ((size_t)nnz > h->csr_cap) return fail(h, XK_ECAPACITY, "sparse operand has more than 24 n non-zeros");
  for (int i = 0; i < n; ++i)
    if (row_ptr[i + 1] < row_ptr[i]) return fail(h, XK_EINVAL, "CSR row pointers not monotone");
  for (int k = 0; k < nnz; ++k)
    if (col_idx[k] < 0 || col_idx[k] >= n) return fail(h, XK_EINVAL, "CSR column index outside the state");
  return congruence(h, row_ptr, col_idx, val, nnz, nullptr, 0, 0);
}

// Propagator::propagateCovarianceMatrices (propagator.cpp:166-205) on the resident covariance:
//   P_ii <- F P_ii F^T + Q,  P_iv <- F P_iv,  P_vi <- P_vi F^T (computed on its own, as the reference insists),  P_vv kept
extern "C" int xk_cov_propagate(xk_handle *h, const double *f_d, int ldf, const double *q_d, int ldq) {
  if (!h || !f_d || !q_d || ldf < XK_CORE || ldq < XK_CORE) return XK_EINVAL;
  const int n = h->n;
  HIPCHK(h, hipSetDevice(h->device));
  XkPropArgs a;
  a.P = h->d_P; a.n = n;
  for (int c = 0; c < XK_CORE; ++c)
    for (int r = 0; r < XK_CORE; ++r) {
      a.FQ[r + XK_CORE * c] = f_d[r + (size_t)c * ldf];
      a.FQ[225 + r + XK_CORE * c] = q_d[r + (size_t)c * ldq];
    }
  hipLaunchKernelGGL(xk_cov_propagate_k, dim3(1 + (2 * (n - XK_CORE) + 255) / 256), dim3(256), 0, h->stream, a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(h, XK_EDEVICE, "propagate launch", e);
  h->have_rows = h->have_R = false;
  return XK_OK;
}

// ---------------------------------------------------------------------------
// Device-resident CI round: the gathered SimpleState payloads (and the observations of the shared
// tracks) stay where RCCL put them.  Same arithmetic as xk_msckf_ci_track + xk_apply_ci per shared
// track, with every per-agent stage batched over the agents and no host staging of the n x n
// covariances.  One host synchronisation per track (the two gate decisions).
// ---------------------------------------------------------------------------
extern "C" int xk_ci_round_device(xk_handle *h, const double *d_payloads, long payload_stride, int world, int self_rank,
                                  const double *d_tracks, int n_tracks, const int *track_len, const int *n_poses_valid,
                                  const int *self_track, double sigma_img, double ci_msckf_w, int *n_fused,
                                  double *corrections) {
  if (!h || !d_payloads || !d_tracks || !track_len || !n_poses_valid || !self_track || !n_fused) return XK_EINVAL;
  const int k = world - 1, k1 = world, N = h->N, n = h->n, m = 3 * k;
  *n_fused = 0;
  if (world < 2) return XK_OK;
  if (k > XK_CI_MAXK) return fail(h, XK_ECAPACITY, "more than 7 matched agents");
  if (self_rank < 0 || self_rank >= world || n_tracks < 0 || n_tracks > 8) return XK_EINVAL;
  if (payload_stride != xk_payload_doubles(N, h->Mmax)) return fail(h, XK_EINVAL, "payload layout differs from this handle's (N, M)");
  if (h->n_poses < 2) return fail(h, XK_EINVAL, "window not staged");
  if (check_w(ci_msckf_w) != XK_OK) return fail(h, XK_EINVAL, "The CI weights must be lower than 1.0 and larger 0.0");
  if (m > h->CM) return fail(h, XK_ECAPACITY, "m exceeds the dense workspace");
  HIPCHK(h, hipSetDevice(h->device));
  const size_t upsz = 3 * (size_t)n + 16;
  const size_t Lcap = (size_t)k1 * 64;
  const size_t ci_ws_doubles = (size_t)9 * 8 * 64 + 8 * upsz + (size_t)21 * 8 * n + 8 * XK_CI_MAXCHUNK * 576 + 2 * 576 + 512;
    if (!h->d_ciws) {
    HIPCHK(h, hipMalloc((void **)&h->d_ciws, sizeof(double) * 8 * ci_ws_doubles));   // one region per shared track
    HIPCHK(h, hipMalloc((void **)&h->d_batch, sizeof(XkFeatBatch) * 64));
    HIPCHK(h, hipHostMalloc((void **)&h->h_batch, sizeof(XkFeatBatch) * 64));
    HIPCHK(h, hipHostMalloc((void **)&h->h_ci_cols, sizeof(int) * (8 * 128 + 8)));   // + the per-track own-gate flags
    HIPCHK(h, hipHostMalloc((void **)&h->h_ci_w, sizeof(double) * 48));   // [0..7] 1/w0, [8..15] joint gamma; [16 + 4 j ..] track j: own verdict, joint gamma, marker
    memset(h->h_ci_w, 0, sizeof(double) * 48);
    hipFuncSetAttribute((const void *)xk_ci_hph, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    HIPCHK(h, hipEventCreateWithFlags(&h->ci_fork, hipEventDisableTiming));
    for (int j = 1; j < 8; ++j) {
      HIPCHK(h, hipStreamCreateWithFlags(&h->ci_stream[j], hipStreamNonBlocking));
      HIPCHK(h, hipEventCreateWithFlags(&h->ci_join[j], hipEventDisableTiming));
    }
  }
  // per-track workspace (the stages before the gate of every shared track are queued back to back, then ONE
  // synchronisation fetches all the gate results)
  struct CiWs { double *dq, *dp, *dobs, *up, *Hs, *Si, *S1, *S2, *dres, *dgpf, *dscal; int *dint; };
  auto ci_ws = [&](int j) {
    CiWs w;
    double *ws = h->d_ciws + (size_t)j * ci_ws_doubles;
    w.dq = ws; w.dp = w.dq + 4 * Lcap; w.dobs = w.dp + 3 * Lcap; w.up = w.dobs + 2 * Lcap;
    w.Hs = w.up + k1 * upsz; w.Si = w.Hs + (size_t)std::max(m, 1) * k1 * n; w.S1 = w.Si + (size_t)k1 * XK_CI_MAXCHUNK * 576; w.S2 = w.S1 + 576;
    w.dres = w.S2 + 576; w.dgpf = w.dres + 24; w.dscal = w.dgpf + 8;   // dscal[0] ci gamma, [1] w_result, [2..] per-agent gamma
    w.dint = (int *)(w.dscal + 16);   // [0..7] inlier per agent, [8..15] tile rows, [16..31] gn iters, [32..159] block columns, [192..] landmarks
    return w;
  };
  // payload layout (fleet.py / xk_pack_payload): hdr[8] dyn[16] pos[3N] att[4N] feat[3M] anchors[M] cov[n*n]
  const size_t o_pos = 24, o_att = o_pos + 3 * (size_t)N, o_cov = o_att + 4 * (size_t)N + 4 * (size_t)h->Mmax;
  const size_t trk_stride = 1 + 2 * (size_t)N;
  const double w0 = 1.0 - (double)k * ci_msckf_w, var_img = sigma_img * sigma_img;
  { int rcw = flush_window(h); if (rcw != XK_OK) return rcw; }
  int fused = 0;
  int trk_L0[8], trk_dof[8];
  // Every track is validated BEFORE anything is queued or forked: an early return below this point would leave side streams
  // running into workspace and pinned words the next call reuses.
  for (int j = 0; j < n_tracks; ++j) {
    const int st = self_track[j];
    if (st < 0 || st >= h->K) return fail(h, XK_EINVAL, "shared track index outside the staged tracks");
    int Ltot = h->h_trk_off[st + 1] - h->h_trk_off[st];
    for (int r = 0; r < world; ++r) {
      if (r == self_rank) continue;
      const int np_r = n_poses_valid[r], L_r = track_len[r * n_tracks + j];
      if (L_r < 2 || L_r > np_r || np_r > N) return fail(h, XK_EINVAL, "received track / window lengths inconsistent");
      Ltot += L_r;
    }
    if (2 * Ltot - 3 >= XK_CHI2_LEN) return fail(h, XK_ECAPACITY, "chi-square table too short");
  }
  // (a HIP error after the fork: the side streams are joined before the error is returned)
  auto bail = [&](int rc) {
    for (int j = 1; j < 8; ++j)
      if (h->ci_stream[j]) hipStreamSynchronize(h->ci_stream[j]);
    hipStreamSynchronize(h->stream);
    return rc;
  };
#define CI_CHK(call)                                                                  \
  do {                                                                                \
    hipError_t e_ = (call);                                                           \
    if (e_ != hipSuccess) return bail(fail(h, XK_EDEVICE, #call, e_));                \
  } while (0)
  // The shared tracks are independent until applyCI (every P_j is built from the same prior), and a track's stages are a chain of
  // seven small launches (~160 us at 8 agents): track j >= 1 runs its chain on a side stream next to track 0's.
  const unsigned long long ci_seq = ++h->done_seq;
  static const int side_env = env_int("XK_CI_SIDE_STREAMS", 1);
  const bool side = side_env && n_tracks > 1;
  if (side) CI_CHK(hipEventRecord(h->ci_fork, h->stream));
  // Round 6: the launches of a track's chain are PREPARED here and issued stage by stage over all tracks below.  The round was bound by
  // the host issuing 2 x 8 launches one track after the other (~7 us each: the second track's chain started 60 us after the first
  // one's, tools/exp/ci_round_kernels.sh); stage-major, both chains are in flight from the first launch on.
  std::vector<std::function<hipError_t()>> ci_stage[8];
  for (int j = 0; j < n_tracks; ++j) {
    hipStream_t sj = (side && j > 0) ? h->ci_stream[j] : h->stream;
    if (side && j > 0) CI_CHK(hipStreamWaitEvent(sj, h->ci_fork, 0));
    const CiWs w = ci_ws(j);
    double *dq = w.dq, *dp = w.dp, *dobs = w.dobs, *up = w.up, *Hs = w.Hs, *Si = w.Si, *S1 = w.S1, *S2 = w.S2;
    double *dres = w.dres, *dgpf = w.dgpf, *dscal = w.dscal;
    int *dint = w.dint;
    const int st = self_track[j];
    // agent order: index 0 = self, 1.. = the others by rank
    const double *aq[XK_CI_MAXK + 1], *ap[XK_CI_MAXK + 1], *aobs[XK_CI_MAXK + 1], *aP[XK_CI_MAXK + 1];
    int anp[XK_CI_MAXK + 1], aL[XK_CI_MAXK + 1], Ltot = 0;
    aq[0] = h->d_q; ap[0] = h->d_p; aobs[0] = h->d_obs + 2 * (size_t)h->h_trk_off[st]; aP[0] = h->d_P;
    anp[0] = h->n_poses; aL[0] = h->h_trk_off[st + 1] - h->h_trk_off[st];
    for (int r = 0, i = 1; r < world; ++r) {
      if (r == self_rank) continue;
      const double *base = d_payloads + (size_t)r * payload_stride;
      aq[i] = base + o_att; ap[i] = base + o_pos; aP[i] = base + o_cov;
      aobs[i] = d_tracks + ((size_t)r * n_tracks + j) * trk_stride + 1;
      anp[i] = n_poses_valid[r]; aL[i] = track_len[r * n_tracks + j];
      ++i;
    }
    for (int i = 0; i < k1; ++i) Ltot += aL[i];
    // joint triangulation over the concatenated lists: matched agents first, self last (:113-149)
    XkCiGatherArgs ga;
    memset(&ga, 0, sizeof(ga));
    ga.k1 = k1; ga.dq = dq; ga.dp = dp; ga.dobs = dobs;
    for (int i = 0; i < k1; ++i) {
      const int src = (i < k) ? i + 1 : 0;
      ga.q[i] = aq[src]; ga.p[i] = ap[src]; ga.obs[i] = aobs[src]; ga.np[i] = anp[src]; ga.L[i] = aL[src];
    }
    ci_stage[0].push_back([=]() { hipLaunchKernelGGL(xk_ci_gather, dim3(k1), dim3(64), 0, sj, ga); return hipSuccess; });
    XkTriMultiArgs ta{dq, dp, dobs, Ltot, dgpf, dint + 16, nullptr, 0};
    ci_stage[1].push_back([=]() { hipLaunchKernelGGL(xk_triangulate_multi, dim3(1), dim3(64), 0, sj, ta); return hipSuccess; });
    // per-agent column-space rows, one workgroup per agent (:168-204)
    XkFeatBatch *hb = h->h_batch + 8 * j, *db = h->d_batch + 8 * j;
    int npmax = 0;
    for (int i = 0; i < k1; ++i) {
      hb[i].q = aq[i]; hb[i].p = ap[i]; hb[i].obs = aobs[i]; hb[i].P = aP[i];
      hb[i].n_poses = anp[i]; hb[i].n_poses_max = N; hb[i].n = n; hb[i].L = aL[i]; hb[i].up_out = up + i * upsz;
      npmax = std::max(npmax, anp[i]);
    }
    ci_stage[2].push_back([=]() { return hipMemcpyAsync(db, hb, sizeof(XkFeatBatch) * k1, hipMemcpyHostToDevice, sj); });
    XkFeatArgs a;
    memset(&a, 0, sizeof(a));
    a.K = k1; a.var_img = var_img; a.chi95 = h->d_chi95; a.n = n; a.na = n - XK_CORE; a.n_poses = npmax; a.n_poses_max = N;
    a.tile_rows = dint + 8; a.inlier = dint; a.gamma = dscal + 2; a.gpf = (double *)(dint + 192); a.gn_iters = dint + 24;
    a.gpf_in = dgpf; a.batch = db;
    ci_stage[3].push_back([=]() { hipLaunchKernelGGL(xk_msckf_feature, dim3(k1), dim3(XK_FEAT_THREADS), xk_feature_lds_bytes(npmax), sj, a); return hipSuccess; });
    // null-space projection of the landmark and split into per-agent Jacobians (:207-223)
    XkCiProjArgs pa;
    memset(&pa, 0, sizeof(pa));
    pa.k1 = k1; pa.res = dres;
    for (int i = 0; i < k1; ++i) { pa.up[i] = up + i * upsz; pa.n[i] = n; pa.H[i] = Hs + (size_t)m * n * i; }
    ci_stage[4].push_back([=]() { hipLaunchKernelGGL(xk_ci_project, dim3(k1), dim3(256), 0, sj, pa); return hipSuccess; });
    // S_i = H_i P_i H_i^T for all agents, then the gate / CI combinations and gamma
    XkCiHphArgs ha;
    memset(&ha, 0, sizeof(ha));
    ha.m = m; ha.S = Si;
    for (int i = 0; i < k1; ++i) { ha.H[i] = pa.H[i]; ha.P[i] = aP[i]; ha.n[i] = n; }
    const int nchunk = (n + XK_CI_CHUNK - 1) / XK_CI_CHUNK;
    ci_stage[5].push_back([=]() { hipLaunchKernelGGL(xk_ci_hph, dim3(k1, nchunk), dim3(256), sizeof(double) * ((size_t)m * n + 24 * 33), sj, ha); return hipSuccess; });
    // the two gate decisions (own chi-square test :180, joint test :243-250) come back per track: written by the kernel into
    // pinned host memory, a marker behind them
    XkCiCombineArgs ca{k1, m, Si, nchunk, w0, ci_msckf_w, var_img, dres, S1, S2, dscal,
                       dint, h->h_ci_w + 16 + 4 * j, reinterpret_cast<unsigned long long *>(h->h_ci_w + 16 + 4 * j + 2), ci_seq};
    ci_stage[6].push_back([=]() { hipLaunchKernelGGL(xk_ci_combine, dim3(1), dim3(512), 0, sj, ca); return hipSuccess; });
    if (side && j > 0) {
      hipEvent_t ej = h->ci_join[j];
      hipStream_t s0 = h->stream;
      ci_stage[7].push_back([=]() { hipError_t e = hipEventRecord(ej, sj); return e != hipSuccess ? e : hipStreamWaitEvent(s0, ej, 0); });
    }
    trk_L0[j] = aL[0];
    trk_dof[j] = 2 * Ltot - 3;
  }
  for (auto &stage : ci_stage)
    for (auto &issue : stage) CI_CHK(issue());
#undef CI_CHK
  if (n_tracks > 0) {
    // wait for the markers of all tracks (XK_SPIN_DONE=0, or a marker that does not come within ~1 s: the runtime's signal)
    static const int spin_env = env_int("XK_SPIN_DONE", 1);
    bool seen = spin_env != 0;
    for (int j = 0; j < n_tracks && seen; ++j) {
      const unsigned long long *mk = reinterpret_cast<const unsigned long long *>(h->h_ci_w + 16 + 4 * j + 2);
      seen = false;
      for (long spins = 0; spins < 40000000L && !(seen = (__atomic_load_n(mk, __ATOMIC_ACQUIRE) == ci_seq)); ++spins) {
        if ((spins & 255) == 255 && __atomic_load_n(&h->d_status[1], __ATOMIC_RELAXED) != 0) break;
        __builtin_ia32_pause();
      }
    }
    if (!seen) {
      HIPCHK(h, hipStreamSynchronize(h->stream));
      for (int j = 1; j < n_tracks && side; ++j) HIPCHK(h, hipStreamSynchronize(h->ci_stream[j]));
    }
  }
  int last_fused = -1;
  for (int j = 0; j < n_tracks; ++j)
    if (h->h_ci_w[16 + 4 * j] != 0.0 && h->h_ci_w[16 + 4 * j + 1] < XK_CHI2_095[trk_dof[j]]) last_fused = j;
  static const int spin_done = env_int("XK_SPIN_DONE", 1);
  unsigned long long *done = reinterpret_cast<unsigned long long *>(h->h_out + h->n + 2);
  unsigned long long wait_seq = 0;
  for (int j = 0; j < n_tracks; ++j) {
    if (h->h_ci_w[16 + 4 * j] == 0.0 || !(h->h_ci_w[16 + 4 * j + 1] < XK_CHI2_095[trk_dof[j]])) continue;
    const CiWs w = ci_ws(j);
    double *Hs = w.Hs, *S2 = w.S2, *dres = w.dres, *dscal = w.dscal;
    int *dint = w.dint;
    // P_j: diagonal 3x3 blocks of the observed poses scaled by 1/w0 (:256-267), then applyCI (updater.cpp:144-161)
    const int L = trk_L0[j];
    // (the blocks are those of the last L window poses: the kernel works their columns out itself -- no staging copies)
    XkScaleArgs sc{h->d_P, h->d_tmpP, n, 2 * L, nullptr, nullptr, 1, h->n_poses - L, L, N, 1.0 / w0};
    (void)dint; (void)dscal;
    hipLaunchKernelGGL(xk_scale_blocks, dim3(((size_t)n * n + 255) / 256), dim3(256), 0, h->stream, sc);
    UpdateSpec u;
    memset(&u, 0, sizeof(u));
    u.T = Hs; u.str = 1; u.stc = m;      // H of agent 0 (self)
    u.c = m; u.kdim = n; u.col0 = 0;
    u.z = dres; u.sz = 1;
    u.S = S2; u.ssr = 1; u.ssc = m;
    u.Pin = h->d_tmpP; u.Pout = h->d_Pout; u.ct = nullptr; u.cov_update = 1;   // every entry starts from the same prior
    if (j == last_fused && !corrections && spin_done) { u.done_flag = done; u.done_seq = wait_seq = ++h->done_seq; }   // the round's last launch marks its end
    int rc = launch_update(h, u);
    if (rc != XK_OK) return rc;
    if (corrections) HIPCHK(h, hipMemcpyAsync(corrections + (size_t)fused * n, h->d_corr, sizeof(double) * n, hipMemcpyDeviceToHost, h->stream));
    ++fused;                                       // (the next track's gate decision, or read_status below, waits)
  }
  if (fused) {
    bool seen = false;
    if (wait_seq) {   // (as xk_apply_update: the marker is the last store of the last kernel; a status word ends the wait early)
      for (long spins = 0; spins < 40000000L && !(seen = (__atomic_load_n(done, __ATOMIC_ACQUIRE) == wait_seq)); ++spins) {
        if ((spins & 255) == 255 && __atomic_load_n(&h->d_status[1], __ATOMIC_RELAXED) != 0) break;
        __builtin_ia32_pause();
      }
      if (seen) h->done_seen = wait_seq;
    }
    int rc;
    if (seen) { stage_stream_idle(h); rc = eval_status(h, h->d_status[0], h->d_status[1], false); }
    else rc = read_status(h);
    if (rc != XK_OK) return rc;
    std::swap(h->d_P, h->d_Pout);   // applyCI overwrites P: the last fused entry is the resident covariance
    h->have_rows = h->have_R = false;
  }
  *n_fused = fused;
  return XK_OK;
}

// ---------------------------------------------------------------------------
// inter-agent payload
// ---------------------------------------------------------------------------
extern "C" long xk_payload_doubles(int N, int M) {
  const long n = XK_CORE + 6L * N + 3L * M;
  return 8 + 16 + 3L * N + 4L * N + 3L * M + M + n * n;
}

__global__ void xk_pack_small(const double *hdr_dyn /*24*/, const double *p, const double *q, int n_poses,
                              const double *feat, const int *anchors, int Mcur, int N, int M, double *out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int o_pos = 24, o_att = o_pos + 3 * N, o_feat = o_att + 4 * N, o_anc = o_feat + 3 * M, o_cov = o_anc + M;
  if (i >= o_cov) return;
  double v = 0.0;
  if (i < 24) v = hdr_dyn[i];
  else if (i < o_att) { const int t = i - o_pos; v = (t < 3 * n_poses) ? p[t] : 0.0; }
  else if (i < o_feat) { const int t = i - o_att; v = (t < 4 * n_poses) ? q[t] : 0.0; }
  else if (i < o_anc) { const int t = i - o_feat; v = (t < 3 * Mcur) ? feat[t] : 0.0; }
  else { const int t = i - o_anc; v = (t < Mcur) ? (double)anchors[t] : -1.0; }
  out[i] = v;
}

extern "C" int xk_pack_payload(xk_handle *h, double agent_id, double timestamp, const double *dyn16,
                               double *d_dst, double **d_payload) {
  if (!h || !dyn16) return XK_EINVAL;
  double *dst = d_dst ? d_dst : h->d_payload;
  HIPCHK(h, hipSetDevice(h->device));
  { int rcw = flush_window(h); if (rcw != XK_OK) return rcw; }
  double *hd = h->h_pin;
  hd[0] = agent_id; hd[1] = timestamp; hd[2] = h->N; hd[3] = h->Mmax; hd[4] = h->n; hd[5] = h->n_poses;
  hd[6] = 0.0; hd[7] = 0.0;
  for (int i = 0; i < 16; ++i) hd[8 + i] = dyn16[i];
  HIPCHK(h, hipMemcpyAsync(h->d_ci, hd, sizeof(double) * 24, hipMemcpyHostToDevice, h->stream));
  const int small = 24 + 7 * h->N + 4 * h->Mmax;
  hipLaunchKernelGGL(xk_pack_small, dim3((small + 255) / 256), dim3(256), 0, h->stream, h->d_ci, h->d_p, h->d_q,
                     h->n_poses, h->d_feat, h->d_anchor, h->M, h->N, h->Mmax, dst);
  HIPCHK(h, hipMemcpyAsync(dst + small, h->d_P, sizeof(double) * (size_t)h->n * h->n,
                           hipMemcpyDeviceToDevice, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  if (d_payload) *d_payload = dst;
  return XK_OK;
}

extern "C" int xk_run_steps(xk_handle *h, double sigma_img, int steps) {
  if (!h || steps < 0 || !(sigma_img > 0.0)) return XK_EINVAL;
  HIPCHK(h, hipSetDevice(h->device));
#ifdef XK_LAB
  static const int use_graph = env_int("XK_GRAPH", 0);
  if (use_graph && steps > 1) {
    // experiment: the 30 launches of one update captured once and replayed
    hipGraph_t g = nullptr;
    hipGraphExec_t ge = nullptr;
    HIPCHK(h, hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal));
    int rc = launch_build(h, sigma_img);
    if (rc == XK_OK) rc = launch_compress(h);
    if (rc == XK_OK) { UpdateSpec u = compressed_spec(h, nullptr, 1); rc = launch_update(h, u); }
    hipError_t e = hipStreamEndCapture(h->stream, &g);
    if (rc != XK_OK) return rc;
    if (e != hipSuccess) return fail(h, XK_EDEVICE, "graph capture", e);
    HIPCHK(h, hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int it = 0; it < steps; ++it) HIPCHK(h, hipGraphLaunch(ge, h->stream));
    rc = read_status(h);
    hipGraphExecDestroy(ge);
    hipGraphDestroy(g);
    return rc;
  }
#endif
  // (every step recomputes the same update from the resident prior, so a single-launch CAQR that gave up -- e.g. two
  //  processes sharing one GPU, each with a grid that wants every CU -- costs one more pass with the multi-launch schedule)
  for (int attempt = 0; attempt < 2; ++attempt) {
    for (int it = 0; it < steps; ++it) {
      int rc = launch_build(h, sigma_img);
      if (rc != XK_OK) return rc;
      UpdateSpec u = compressed_spec(h, nullptr, 1);
      rc = launch_compress(h, nullptr, &u);
      if (rc != XK_OK) return rc;
      if (!h->last_fused) rc = launch_update(h, u);
      if (rc != XK_OK) return rc;
    }
    const int rc = read_status(h, attempt == 0);
    if (rc != XK_RETRY_CLASSIC) return rc;
  }
  return XK_EDEVICE;
}

// ---------------------------------------------------------------------------
// fp64 ceiling probe (lab build only: include/xk_lab.h)
// ---------------------------------------------------------------------------
#ifdef XK_LAB
__global__ __launch_bounds__(256) void xk_probe_mfma(double *out, int iters) {
  xk_d4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
  const double x = 1.0 + threadIdx.x * 1e-9, y = 1.0 - threadIdx.x * 1e-9;
  for (int i = 0; i < iters; ++i) {
    a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, x, a1, 0, 0, 0);
    a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, a2, 0, 0, 0);
    a3 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, y, a3, 0, 0, 0);
  }
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3];
}
__global__ __launch_bounds__(256) void xk_probe_fma(double *out, int iters) {
  double a[8];
  const double x = 1.0 + threadIdx.x * 1e-9, y = 1e-9 * threadIdx.x;
#pragma unroll
  for (int k = 0; k < 8; ++k) a[k] = k;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] = fma(a[k], x, y);
  }
  double s = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) s += a[k];
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}

extern "C" int xk_probe_fp64_peak(xk_handle *h, int use_mfma, double *tflops) {
  if (!h || !tflops) return XK_EINVAL;
  HIPCHK(h, hipSetDevice(h->device));
  const int blocks = 256 * 8, iters = 20000;
  double *buf = nullptr;
  HIPCHK(h, hipMalloc((void **)&buf, sizeof(double) * blocks * 256));
  for (int rep = 0; rep < 2; ++rep) {
    HIPCHK(h, hipEventRecord(h->ev[8], h->stream));
    if (use_mfma) hipLaunchKernelGGL(xk_probe_mfma, dim3(blocks), dim3(256), 0, h->stream, buf, iters);
    else hipLaunchKernelGGL(xk_probe_fma, dim3(blocks), dim3(256), 0, h->stream, buf, iters);
    HIPCHK(h, hipEventRecord(h->ev[9], h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
  }
  float ms = 0;
  hipEventElapsedTime(&ms, h->ev[8], h->ev[9]);
  hipFree(buf);
  const double waves = (double)blocks * 4;
  const double flops = use_mfma ? waves * iters * 4.0 * (2.0 * 16 * 16 * 4) : waves * 64.0 * iters * 8.0 * 2.0;
  *tflops = flops / (ms * 1e-3) / 1e12;
  return XK_OK;
}
#endif   // XK_LAB

// Which schedule compressed the last update, and how the fast path has fared on this handle (include/xk.h).
extern "C" int xk_set_option(xk_handle *h, const char *name, int value) {
  if (!h || !name) return XK_EINVAL;
  // operational switches of the release library: the schedule of the compression and how soon a fast path that gave up is retried
  if (!strcmp(name, "caqr_resident")) h->opt_resident = value;
  else if (!strcmp(name, "caqr_rearm")) h->rearm_after = value;
  else if (!strcmp(name, "caqr_tail")) h->opt_tail = value;
  else if (!strcmp(name, "slam_split")) h->opt_slam_split = value;
#ifdef XK_LAB
  // test hooks and A/B switches (include/xk_lab.h)
  else if (!strcmp(name, "caqr_poison")) h->opt_poison = value;
  else if (!strcmp(name, "caqr_test_stall")) h->opt_test_stall = value;
  else if (!strcmp(name, "caqr_tall26")) h->opt_tall26 = value;
  else if (!strcmp(name, "pipe_kalman")) h->opt_kalman = value;
  else if (!strcmp(name, "pipe_split")) { if (h->opt_split >= 0) h->opt_split = value; }
  else if (!strcmp(name, "caqr_hlite")) h->opt_hlite = value;
#endif
  else return fail(h, XK_EINVAL, "xk_set_option: unknown option");
  return XK_OK;
}

extern "C" int xk_caqr_status(const xk_handle *h, int *schedule, int *armed, int *giveups, int *last_reason) {
  if (!h) return XK_EINVAL;
  if (schedule) *schedule = h->split_active >= 2 ? 4 : (h->last_tail ? 3 : (!h->last_resident ? 0 : (h->last_pipe ? 2 : 1)));
  if (armed) *armed = (h->persist_ok || h->tail_ok) ? 1 : 0;
  if (giveups) *giveups = h->fast_giveups;
  if (last_reason) *last_reason = h->fast_reason;
  return XK_OK;
}

#ifdef XK_LAB
extern "C" int xk_is_lab(void) { return 1; }
// Wall-clock (100 MHz) stamps of the last single-launch CAQR (XK_CAQR_PERSIST_DBG=1): per panel k, out[8k + 0..5] = role-T
// workgroup (XCD 0, slot 0): tile step start / end, after barrier 1, first-level merge start, end, after barrier 2;
// out[8k + 6..7] = last-level workgroup 0: roots complete -> its strips published.  tools/exp/persist_trace.py prints them.
extern "C" int xk_debug_persist_stamps(xk_handle *h, long long *out, int n_out) {
  if (!h || !out || !h->d_pdbg) return XK_EINVAL;
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  HIPCHK(h, hipMemcpy(out, h->d_pdbg, sizeof(long long) * (size_t)std::min(n_out, XK_PDBG_WORDS), hipMemcpyDeviceToHost));
  return XK_OK;
}
#endif   // XK_LAB

#if defined(XK_FEAT_PROBE) && defined(XK_LAB)
extern "C" int xk_debug_feature_phases(xk_handle *h, double sigma_img, long long *out, int n_out) {
  // out[12k ..]: 8 clock64 phase stamps, wall start, wall end, (XCC_ID << 32 | HW_ID) of workgroup k
  const size_t bytes = sizeof(long long) * (size_t)(12 * h->K);
  hipMalloc((void **)&h->feat_dbg, bytes);
  hipMemset(h->feat_dbg, 0, bytes);
  int rc = launch_build(h, sigma_img);
  hipStreamSynchronize(h->stream);
  hipMemcpy(out, h->feat_dbg, std::min(bytes, sizeof(long long) * (size_t)n_out), hipMemcpyDeviceToHost);
  hipFree(h->feat_dbg);
  h->feat_dbg = nullptr;
  return rc;
}
#endif

// ---------------------------------------------------------------------------
// Place-recognition request filter + keyframe store: the component either side of the CI
// exchange on the communication axis.  Mirrors VLAD / Database / Keyframe of src/x/place_recognition.
// ---------------------------------------------------------------------------
#include "xk_place.hip.h"

#define XK_PR_MAX_KEYFRAMES 15   // database.h:70

struct xk_pr {
  xk_handle *h;
  int k, L, n_nodes, kmax, W, n_words, clusters, VW, max_desc;
  long pay_n, trk_n;
  unsigned int *d_node_desc;
  int *d_children, *d_word_of_node, *d_node_of_word;
  // keyframe store: slot s of the ring holds one keyframe; order[] lists the live slots oldest first
  unsigned int *d_vlad;       // [15][VW]
  unsigned int *d_kfdesc;     // [15][max_desc][W]
  double *d_payload;          // [15][pay_n]
  double *d_tracks;           // [15][trk_n]
  int n_desc[XK_PR_MAX_KEYFRAMES];
  long tag[XK_PR_MAX_KEYFRAMES];
  std::vector<int> *uav_ids[XK_PR_MAX_KEYFRAMES];   // Keyframe::uav_ids_ (a std::set in the reference)
  int order[XK_PR_MAX_KEYFRAMES], live;
  // scratch
  unsigned int *d_q, *d_t, *d_qvlad;
  int *d_ham, *d_knn;
  int *h_int;                 // pinned
  unsigned int *h_words;      // pinned staging for descriptors / VLADs
  size_t h_words_cap;
};

extern "C" void xk_pr_destroy(xk_pr *p) {
  if (!p) return;
  hipFree(p->d_node_desc); hipFree(p->d_children); hipFree(p->d_word_of_node); hipFree(p->d_node_of_word);
  hipFree(p->d_vlad); hipFree(p->d_kfdesc); hipFree(p->d_payload); hipFree(p->d_tracks);
  hipFree(p->d_q); hipFree(p->d_t); hipFree(p->d_qvlad); hipFree(p->d_ham); hipFree(p->d_knn);
  if (p->h_int) hipHostFree(p->h_int);
  if (p->h_words) hipHostFree(p->h_words);
  for (auto &u : p->uav_ids) delete u;
  free(p);
}

extern "C" int xk_pr_create(xk_handle *h, int k, int L, int n_nodes, int kmax, int desc_bytes,
                            const unsigned char *node_desc, const int *children, const int *word_of_node,
                            const int *node_of_word, int n_words, long payload_doubles, long tracks_doubles,
                            int max_desc, xk_pr **out) {
  if (!h || !out || !node_desc || !children || !word_of_node || !node_of_word) return XK_EINVAL;
  if (k < 1 || L < 1 || n_nodes < 2 || kmax < 1 || n_words < 1 || max_desc < 1 || payload_doubles < 0 || tracks_doubles < 0)
    return fail(h, XK_EINVAL, "xk_pr_create: bad sizes");
  if (desc_bytes < 4 || desc_bytes % 4 || desc_bytes > 4 * XK_PR_MAXW)
    return fail(h, XK_EINVAL, "xk_pr_create: descriptor size must be a multiple of 4 bytes, at most 64");
  double cl = 1.0;
  for (int i = 0; i < L; ++i) cl *= k;                       // pow(k, L), vlad.cpp:27-28
  if (cl * desc_bytes > (double)(64 << 20)) return fail(h, XK_ECAPACITY, "xk_pr_create: VLAD larger than 64 MB");
  for (int i = 0; i < n_nodes; ++i)
    if (word_of_node[i] >= n_words || word_of_node[i] >= (int)cl) return fail(h, XK_EINVAL, "xk_pr_create: word id out of range");
  xk_pr *p = (xk_pr *)calloc(1, sizeof(xk_pr));
  if (!p) return XK_ENOMEM;
  p->h = h; p->k = k; p->L = L; p->n_nodes = n_nodes; p->kmax = kmax; p->W = desc_bytes / 4; p->n_words = n_words;
  p->clusters = (int)cl; p->VW = p->clusters * p->W; p->max_desc = max_desc; p->pay_n = payload_doubles; p->trk_n = tracks_doubles;
  for (auto &u : p->uav_ids) u = new std::vector<int>();
  const size_t wcap = std::max((size_t)max_desc * p->W * 2, (size_t)p->VW * 2);
  p->h_words_cap = wcap;
  bool ok = dalloc(&p->d_node_desc, (size_t)n_nodes * p->W) == hipSuccess && dalloc(&p->d_children, (size_t)n_nodes * kmax) == hipSuccess &&
            dalloc(&p->d_word_of_node, (size_t)n_nodes) == hipSuccess && dalloc(&p->d_node_of_word, (size_t)n_words) == hipSuccess &&
            dalloc(&p->d_vlad, (size_t)XK_PR_MAX_KEYFRAMES * p->VW) == hipSuccess &&
            dalloc(&p->d_kfdesc, (size_t)XK_PR_MAX_KEYFRAMES * max_desc * p->W) == hipSuccess &&
            dalloc(&p->d_payload, (size_t)XK_PR_MAX_KEYFRAMES * payload_doubles) == hipSuccess &&
            dalloc(&p->d_tracks, (size_t)XK_PR_MAX_KEYFRAMES * tracks_doubles) == hipSuccess &&
            dalloc(&p->d_q, (size_t)max_desc * p->W) == hipSuccess && dalloc(&p->d_t, (size_t)max_desc * p->W) == hipSuccess &&
            dalloc(&p->d_qvlad, (size_t)p->VW) == hipSuccess && dalloc(&p->d_ham, (size_t)XK_PR_MAX_KEYFRAMES) == hipSuccess &&
            dalloc(&p->d_knn, (size_t)max_desc * 4) == hipSuccess &&
            hipHostMalloc((void **)&p->h_int, sizeof(int) * ((size_t)max_desc * 4 + 64)) == hipSuccess &&
            hipHostMalloc((void **)&p->h_words, sizeof(unsigned int) * wcap) == hipSuccess;
  if (ok) {
    ok = hipMemcpy(p->d_node_desc, node_desc, (size_t)n_nodes * desc_bytes, hipMemcpyHostToDevice) == hipSuccess &&
         hipMemcpy(p->d_children, children, sizeof(int) * (size_t)n_nodes * kmax, hipMemcpyHostToDevice) == hipSuccess &&
         hipMemcpy(p->d_word_of_node, word_of_node, sizeof(int) * (size_t)n_nodes, hipMemcpyHostToDevice) == hipSuccess &&
         hipMemcpy(p->d_node_of_word, node_of_word, sizeof(int) * (size_t)n_words, hipMemcpyHostToDevice) == hipSuccess;
  }
  if (!ok) { xk_pr_destroy(p); return fail(h, XK_ENOMEM, "xk_pr_create: allocation failed"); }
  *out = p;
  return XK_OK;
}

extern "C" int xk_pr_vlad_bytes(const xk_pr *p) { return p ? p->VW * 4 : 0; }
extern "C" int xk_pr_size(const xk_pr *p) { return p ? p->live : 0; }

// descriptors (host) -> VLAD in `d_dst` (device); the descriptors stay in p->d_q afterwards
static int pr_vlad_device(xk_pr *p, const unsigned char *desc, int n, unsigned int *d_dst) {
  xk_handle *h = p->h;
  if (n < 0 || n > p->max_desc) return fail(h, XK_ECAPACITY, "place recognition: more descriptors than max_desc");
  HIPCHK(h, hipMemsetAsync(d_dst, 0, sizeof(unsigned int) * p->VW, h->stream));
  if (n > 0) {
    if (!desc) return fail(h, XK_EINVAL, "place recognition: null descriptors");
    memcpy(p->h_words, desc, (size_t)n * p->W * 4);
    HIPCHK(h, hipMemcpyAsync(p->d_q, p->h_words, (size_t)n * p->W * 4, hipMemcpyHostToDevice, h->stream));
    XkVladArgs a{p->d_q, n, p->W, p->d_node_desc, p->d_children, p->kmax, p->d_word_of_node, p->d_node_of_word, d_dst};
    hipLaunchKernelGGL(xk_vlad_build, dim3((n + 255) / 256), dim3(256), 0, h->stream, a);
  }
  return XK_OK;
}

extern "C" int xk_pr_compute_vlad(xk_pr *p, const unsigned char *desc, int n, unsigned char *vlad_out) {
  if (!p || !vlad_out) return XK_EINVAL;
  xk_handle *h = p->h;
  HIPCHK(h, hipSetDevice(h->device));
  int rc = pr_vlad_device(p, desc, n, p->d_qvlad);
  if (rc != XK_OK) return rc;
  HIPCHK(h, hipMemcpyAsync(p->h_words, p->d_qvlad, sizeof(unsigned int) * p->VW, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  memcpy(vlad_out, p->h_words, sizeof(unsigned int) * p->VW);
  return XK_OK;
}

extern "C" int xk_pr_add_keyframe(xk_pr *p, const unsigned char *desc, int n_desc, const double *d_payload,
                                  const double *d_tracks, long tag) {
  if (!p) return XK_EINVAL;
  xk_handle *h = p->h;
  // (checked BEFORE the oldest keyframe is dropped: a rejected call leaves the database as it was)
  if (n_desc < 0 || n_desc > p->max_desc) return fail(h, XK_ECAPACITY, "place recognition: more descriptors than max_desc");
  if (n_desc > 0 && !desc) return fail(h, XK_EINVAL, "place recognition: null descriptors");
  HIPCHK(h, hipSetDevice(h->device));
  // slot: a free one, or the oldest keyframe's (erase(begin()), database.cpp:56-58)
  int slot;
  if (p->live < XK_PR_MAX_KEYFRAMES) {
    bool used[XK_PR_MAX_KEYFRAMES] = {false};
    for (int i = 0; i < p->live; ++i) used[p->order[i]] = true;
    slot = 0;
    while (used[slot]) ++slot;
  } else {
    slot = p->order[0];
    for (int i = 1; i < p->live; ++i) p->order[i - 1] = p->order[i];
    --p->live;
  }
  int rc = pr_vlad_device(p, desc, n_desc, p->d_vlad + (size_t)slot * p->VW);
  if (rc != XK_OK) return rc;
  if (n_desc > 0)
    HIPCHK(h, hipMemcpyAsync(p->d_kfdesc + (size_t)slot * p->max_desc * p->W, p->d_q, (size_t)n_desc * p->W * 4,
                             hipMemcpyDeviceToDevice, h->stream));
  if (d_payload && p->pay_n)
    HIPCHK(h, hipMemcpyAsync(p->d_payload + (size_t)slot * p->pay_n, d_payload, sizeof(double) * p->pay_n, hipMemcpyDeviceToDevice, h->stream));
  if (d_tracks && p->trk_n)
    HIPCHK(h, hipMemcpyAsync(p->d_tracks + (size_t)slot * p->trk_n, d_tracks, sizeof(double) * p->trk_n, hipMemcpyDeviceToDevice, h->stream));
  p->n_desc[slot] = n_desc; p->tag[slot] = tag; p->uav_ids[slot]->clear();
  p->order[p->live++] = slot;
  return XK_OK;
}

extern "C" int xk_pr_find_candidate(xk_pr *p, int uav_id, const unsigned char *query_vlad, double pr_score_thr, int *index,
                                    double *score, long *tag) {
  if (!p || !query_vlad || !index) return XK_EINVAL;
  xk_handle *h = p->h;
  HIPCHK(h, hipSetDevice(h->device));
  *index = -1;
  if (score) *score = 0.0;
  if (tag) *tag = -1;
  if (p->live == 0) return XK_OK;
  memcpy(p->h_words, query_vlad, sizeof(unsigned int) * p->VW);
  HIPCHK(h, hipMemcpyAsync(p->d_qvlad, p->h_words, sizeof(unsigned int) * p->VW, hipMemcpyHostToDevice, h->stream));
  XkVladHamArgs a{p->d_qvlad, p->d_vlad, p->VW, p->d_ham};
  hipLaunchKernelGGL(xk_vlad_hamming, dim3(XK_PR_MAX_KEYFRAMES), dim3(256), 0, h->stream, a);
  HIPCHK(h, hipMemcpyAsync(p->h_int, p->d_ham, sizeof(int) * XK_PR_MAX_KEYFRAMES, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  // the selection loop of Database::findCandidate (database.cpp:32-45), keyframes oldest first
  const double v_length = (double)p->VW * 32.0;
  double best = 0.0;
  int best_pos = -1;
  for (int i = 0; i < p->live; ++i) {
    const int s = p->order[i];
    bool seen = false;
    for (int u : *p->uav_ids[s]) seen |= (u == uav_id);
    if (seen) continue;
    const double sc = (v_length - (double)p->h_int[s]) / v_length;      // vlad.cpp:71
    if (sc > pr_score_thr && sc > best) { best = sc; best_pos = i; }
  }
  if (best_pos >= 0) {
    p->uav_ids[p->order[best_pos]]->push_back(uav_id);
    *index = best_pos;
    if (score) *score = best;
    if (tag) *tag = p->tag[p->order[best_pos]];
  }
  return XK_OK;
}

extern "C" int xk_pr_keyframe(xk_pr *p, int index, const double **d_payload, const double **d_tracks, int *n_desc, long *tag,
                              unsigned char *desc_out) {
  if (!p) return XK_EINVAL;
  xk_handle *h = p->h;
  HIPCHK(h, hipSetDevice(h->device));
  if (index < 0 || index >= p->live) return fail(h, XK_EINVAL, "xk_pr_keyframe: no such keyframe");
  const int s = p->order[index];
  if (d_payload) *d_payload = p->d_payload + (size_t)s * p->pay_n;
  if (d_tracks) *d_tracks = p->d_tracks + (size_t)s * p->trk_n;
  if (n_desc) *n_desc = p->n_desc[s];
  if (tag) *tag = p->tag[s];
  if (desc_out && p->n_desc[s] > 0) {
    HIPCHK(h, hipMemcpyAsync(p->h_words, p->d_kfdesc + (size_t)s * p->max_desc * p->W, (size_t)p->n_desc[s] * p->W * 4,
                             hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    memcpy(desc_out, p->h_words, (size_t)p->n_desc[s] * p->W * 4);
  }
  return XK_OK;
}

extern "C" int xk_pr_copy_keyframe(xk_pr *p, int index, double *d_payload_dst, double *d_tracks_dst) {
  if (!p) return XK_EINVAL;
  xk_handle *h = p->h;
  HIPCHK(h, hipSetDevice(h->device));
  if (index < 0 || index >= p->live) return fail(h, XK_EINVAL, "xk_pr_copy_keyframe: no such keyframe");
  const int s = p->order[index];
  if (d_payload_dst && p->pay_n)
    HIPCHK(h, hipMemcpyAsync(d_payload_dst, p->d_payload + (size_t)s * p->pay_n, sizeof(double) * p->pay_n, hipMemcpyDeviceToDevice, h->stream));
  if (d_tracks_dst && p->trk_n)
    HIPCHK(h, hipMemcpyAsync(d_tracks_dst, p->d_tracks + (size_t)s * p->trk_n, sizeof(double) * p->trk_n, hipMemcpyDeviceToDevice, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return XK_OK;
}

extern "C" int xk_pr_knn_match(xk_pr *p, const unsigned char *query, int nq, const unsigned char *train, int nt, int *idx,
                               int *dist) {
  if (!p || !idx || !dist || nq < 0 || nt < 0) return XK_EINVAL;
  xk_handle *h = p->h;
  HIPCHK(h, hipSetDevice(h->device));
  if (nq > p->max_desc || nt > p->max_desc) return fail(h, XK_ECAPACITY, "xk_pr_knn_match: more descriptors than max_desc");
  if (nq == 0) return XK_OK;
  if (!query || (nt > 0 && !train)) return fail(h, XK_EINVAL, "xk_pr_knn_match: null descriptors");
  const size_t qb = (size_t)nq * p->W * 4, tb = (size_t)nt * p->W * 4;
  memcpy(p->h_words, query, qb);
  if (nt) memcpy(p->h_words + (size_t)nq * p->W, train, tb);
  HIPCHK(h, hipMemcpyAsync(p->d_q, p->h_words, qb, hipMemcpyHostToDevice, h->stream));
  if (nt) HIPCHK(h, hipMemcpyAsync(p->d_t, p->h_words + (size_t)nq * p->W, tb, hipMemcpyHostToDevice, h->stream));
  XkKnnArgs a{p->d_q, p->d_t, nq, nt, p->W, p->d_knn, p->d_knn + 2 * (size_t)p->max_desc};
  hipLaunchKernelGGL(xk_desc_knn2, dim3((nq + XK_KNN_Q - 1) / XK_KNN_Q), dim3(256), 0, h->stream, a);
  HIPCHK(h, hipMemcpyAsync(p->h_int, p->d_knn, sizeof(int) * 2 * (size_t)nq, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipMemcpyAsync(p->h_int + 2 * (size_t)nq, p->d_knn + 2 * (size_t)p->max_desc, sizeof(int) * 2 * (size_t)nq,
                           hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  memcpy(idx, p->h_int, sizeof(int) * 2 * (size_t)nq);
  memcpy(dist, p->h_int + 2 * (size_t)nq, sizeof(int) * 2 * (size_t)nq);
  return XK_OK;
}
