/*
 * xk_lab.h -- what the LAB build of the library (-DXK_LAB: x_multi_agent_amd/lab/libxk.so) exports on top of xk.h.
 *
 * Not part of the drop-in boundary: test hooks, A/B switches, debug exports and probe kernels that tests/ and tools/exp use.
 * The release library (x_multi_agent_amd/libxk.so) has none of them and never calls getenv.
 *
 * Environment switches the lab build reads (once each; defaults in parentheses):
 *   at xk_create     XK_CAQR_RESIDENT (1)  XK_CAQR_RESIDENT_POISON (0)  XK_CAQR_TEST_STALL (0)  XK_CAQR_TALL26 (1)  XK_CAQR_REARM (64)
 *                    XK_PIPE_KALMAN (1)  XK_PIPE_SPLIT (1)  XK_HLITE (1)  XK_QUIET (0)
 *   compression      XK_CAQR_WT (0)  XK_CAQR_ARITY1 (auto)  XK_CAQR_CHALF (8)  XK_CAQR_OVERLAP (1)  XK_CAQR_SKIP_REJECTED (1)
 *                    XK_CAQR_PERSIST_DBG (0)  XK_CAQR_LCHALF (auto)  XK_CAQR_ADAPT (1)  XK_CAQR_CUS (256)  XK_CAQR_M32 (1)  XK_CAQR_STREAM (1)
 *   Kalman stage     XK_GEMM_STRUCT (1)  XK_CHOL_WHOLE (1)  XK_CHOL_SPLIT (1)  XK_SPIN_DONE (1)
 *   CI round         XK_CI_SIDE_STREAMS (1)        replay: XK_GRAPH (0)
 * xk_set_option names of the lab build, besides the release ones: "caqr_poison" (1: raise the abort word before every single
 * launch -- it gives up, the host redoes the update), "caqr_test_stall" (1: one workgroup of the single launch never shows up),
 * "caqr_tall26", "pipe_kalman", "pipe_split" (0: 184 tiles always, 1: the adaptive default, 2: 152 tiles when the nominal rows fit,
 * 3: always), "caqr_hlite" (0: the per-feature kernel writes 64-row tiles of H0 as before round 5; 1, default: factor records for the
 * narrow single launch and the multi-launch schedule, csrc/xk_feature.hip.h XkFeatArgs::Hc; 2: for the wide single launch too -- measured
 * slower, config 2: 1546 -> 1524 updates/s).
 */
#ifndef XK_LAB_H_
#define XK_LAB_H_
#include "xk.h"
#ifdef __cplusplus
extern "C" {
#endif

/* 1 -- the symbol only the lab build has (a loader can tell the two libraries apart). */
int xk_is_lab(void);

/* Wall-clock (100 MHz) stamps of the last single-launch CAQR (XK_CAQR_PERSIST_DBG=1), for tools/exp/pipe_trace.py: per panel k,
 * out[16k ..] = one tile workgroup, out[512 + 16k ..] = one first-level workgroup, out[1024 + 16k ..] = one last-level
 * workgroup (phase by phase), out[1536 ..] = start-up and exit.  n_out <= 256 + 64 * 256. */
int xk_debug_persist_stamps(xk_handle *h, long long *out, int n_out);

/* Micro-benchmark of the fp64 ceiling this path is priced against: a grid of waves issuing independent
 * v_mfma_f64_16x16x4_f64 (use_mfma=1) or v_fma_f64 (use_mfma=0) chains.  Reports sustained TFLOP/s. */
int xk_probe_fp64_peak(xk_handle *h, int use_mfma, double *tflops);

#ifdef __cplusplus
}
#endif
#endif /* XK_LAB_H_ */
