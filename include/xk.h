/*
 * xk.h -- C ABI of the MI355X-native xVIO EKF-update engine.
 *
 * This is the drop-in boundary for ONE path of jpl-x/x_multi_agent: the
 * visual EKF update (MSCKF build -> QR compression -> Kalman update) and the
 * covariance-intersection fusion step.  Each entry point names the reference
 * interface it replaces (paths relative to the reference tree).
 *
 * Conventions
 *   - all matrices are column-major IEEE doubles with an explicit leading
 *     dimension (Eigen::MatrixXd::data() can be passed directly);
 *   - quaternions are (x,y,z,w) as stored in State::q_array_
 *     (src/x/ekf/state.cpp:235-240);
 *   - error-state column map (include/x/common/types.h:39-47,
 *     msckf_update.cpp:412-416): [0,15) core, 15+3i position of window pose
 *     i, 15+3N+3i attitude of pose i, 15+6N+3j SLAM feature j;
 *   - the caller owns every host buffer; the library never keeps a host
 *     pointer after a call returns; device memory belongs to the handle;
 *   - one handle per agent / x::Ekf; calls on one handle are serialised by
 *     the caller (as Updater::update is in the reference, ekf.cpp:186-205),
 *     distinct handles are independent;
 *   - every call is synchronous unless its name ends in _async;
 *   - return value is an xk_status; nothing aborts, nothing throws.
 */
#ifndef XK_H_
#define XK_H_

#ifdef __cplusplus
extern "C" {
#endif

typedef struct xk_handle xk_handle;

typedef enum {
  XK_OK = 0,
  XK_EINVAL = 1,    /* bad dims / CI weights (ci.cpp:59-62,98-101 throw) */
  XK_ESINGULAR = 2, /* innovation covariance not SPD */
  XK_ENAN = 3,
  XK_EDEVICE = 4,   /* HIP runtime error; see xk_last_error */
  XK_ENOMEM = 5,
  XK_ECAPACITY = 6  /* problem larger than the handle was created for */
} xk_status;

#define XK_CORE 15 /* kSizeCoreErr, include/x/common/types.h:45 */

/* ---- lifetime -------------------------------------------------------- */

/* Creates the per-agent engine on HIP device `device`.  Capacities:
 * n_poses_max = sliding-window length N (Params::n_poses_max,
 * include/x/vio/types.h), n_feat_max = SLAM feature slots M, k_max = MSCKF
 * tracks per update.  n = 15 + 6N + 3M. */
int xk_create(int device, int n_poses_max, int n_feat_max, int k_max, xk_handle **out);
int xk_destroy(xk_handle *h);
const char *xk_strerror(int status);
const char *xk_last_error(const xk_handle *h);
int xk_version(void);
/* HIP stream the handle launches on (void* = hipStream_t), for callers that
 * want to order their own device work or time with events. */
void *xk_stream(xk_handle *h);

/* ---- staged (device-resident) visual update -------------------------- */

/* Stage the inputs of one visual update in HBM.  Replaces the arguments of
 * the MsckfUpdate / SlamUpdate constructors as called from
 * VioUpdater::constructUpdate (src/x/vio/vio_updater.cpp:279-305,338-346):
 *   C_q_G [n_poses x 4] xyzw, G_p_C [n_poses x 3]   window lists
 *       (StateManager::convertCamera{Attitudes,Positions}ToList,
 *        src/x/vio/state_manager.cpp:539-584), row per pose;
 *   trk_off [K+1], obs_xy [trk_off[K] x 2]          MSCKF tracks, normalised
 *       image coordinates (Feature::getX/getY); a length-L track observes the
 *       LAST L window poses (msckf_update.cpp:329-331);
 *   P [n x n], ldp                                  prior covariance
 *       (Matrix P = state.getCovariance(), vio_updater.cpp:284);
 *   SLAM (M may be 0): feat [3M] inverse-depth states, anchor_idxs [M],
 *       track_sizes [M] (only used for the chi-square dof,
 *       slam_update.cpp:196), z_last [M x 2] newest observation. */
int xk_stage_window(xk_handle *h, const double *C_q_G, const double *G_p_C, int n_poses);
int xk_stage_tracks(xk_handle *h, const int *trk_off, const double *obs_xy, int K);
/* The same in two halves for a host that builds the CSR lists anyway: *trk_off (K+1 ints) and *obs_xy (2 n_obs doubles)
 * point INTO the handle's pinned staging memory; fill them, then call xk_stage_tracks_end (which validates like
 * xk_stage_tracks and queues the one copy).  Replaces the list -> vector -> staging double copy of a C++ host
 * (host/src/vio_updater.cpp, mirror of the list walk in vio_updater.cpp:267-300). */
int xk_stage_tracks_begin(xk_handle *h, int K, int n_obs, int **trk_off, double **obs_xy);
int xk_stage_tracks_end(xk_handle *h);
int xk_stage_slam(xk_handle *h, const double *feat, const int *anchor_idxs, const int *track_sizes,
                  const double *z_last, int M);
int xk_upload_P(xk_handle *h, const double *P, int ldp, int n);
int xk_download_P(xk_handle *h, double *P, int ldp, int n);

/* Per-feature build on the staged inputs: triangulation (triangulation.cpp:
 * 48-206), Jacobians + observability constraint + left-nullspace projection +
 * chi-square gate (msckf_update.cpp:65-173,283-492), SLAM rows
 * (slam_update.cpp:49-214).  Leaves the projected rows [H0|res0]
 * device-resident.  Outputs (host, optional = NULL): inlier/gamma per track. */
int xk_msckf_build(xk_handle *h, double sigma_img, int *inlier_msckf, double *gamma_msckf,
                   int *inlier_slam, double *gamma_slam);

/* Householder QR (communication-avoiding, panel by panel) of the device-resident stacked [H|res]; replaces
 * VioUpdater::applyQRDecomposition (vio_updater.cpp:487-512).  Optional host
 * outputs: T_H (n x n, ldt; upper-trapezoidal, zero core columns) and z (n).
 * T_H^T T_H and T_H^T z equal the reference's up to rounding; the rows
 * themselves differ by an orthogonal factor. */
int xk_qr_compress(xk_handle *h, double *T_H, int ldt, double *z);

/* Kalman gain + covariance/state-correction on the compressed system with
 * R = sigma_img^2 I; replaces Updater::applyUpdate (src/x/ekf/updater.cpp:
 * 117-141): S = HPH^T+R, K = PH^T S^-1, corr = K(res + H corr_tot) - corr_tot,
 * P = (I-KH)P, P = (P+P^T)/2.  P stays on the device (xk_download_P);
 * correction (n) is returned to the host because State::correct
 * (state.cpp:197-249) runs there.  corr_total may be NULL (= 0). */
int xk_apply_update(xk_handle *h, const double *corr_total, int cov_update, double *correction);

/* xk_msckf_build + xk_qr_compress + xk_apply_update in one call on the staged
 * inputs, no host round trips in between (a5..a12 of SURVEY 8a). */
int xk_visual_update_staged(xk_handle *h, double sigma_img, double *correction, int *inlier_msckf,
                            double *gamma_msckf, int *inlier_slam, double *gamma_slam);

/* Convenience: stage + update + download in one call (host buffers in/out;
 * PCIe inclusive).  P is updated in place. */
int xk_visual_update(xk_handle *h, const double *C_q_G, const double *G_p_C, int n_poses,
                     const int *trk_off, const double *obs_xy, int K, const double *feat,
                     const int *anchor_idxs, const int *track_sizes, const double *z_last, int M,
                     double *P, int ldp, int n, double sigma_img, double *correction,
                     int *inlier_msckf, double *gamma_msckf, int *inlier_slam, double *gamma_slam);

/* ---- dense (unfused) Kalman algebra, reference signatures ------------ */

/* Updater::applyUpdate(state, H, res, R, correction_total, cov_update) with
 * arbitrary dense H (m x n) and diagonal R (updater.cpp:117-141).  P in/out
 * on the host; correction_total (n) in/out (updater.cpp:140). */
int xk_apply_update_dense(xk_handle *h, double *P, int ldp, int n, const double *H, int ldh, int m,
                          const double *res, const double *r_diag, double *correction_total,
                          int cov_update, double *correction);

/* Updater::applyCI(state, ci_P, H, res, S) (updater.cpp:144-161):
 * K = ci_P H^T S^-1, corr = K res, P_out = sym((I-KH) ci_P). */
int xk_apply_ci(xk_handle *h, double *P_out, int ldp, const double *ci_P, int ldc, int n,
                const double *H, int ldh, int m, const double *res, const double *S, int lds,
                double *correction);

/* ---- covariance intersection (fixed weights) ------------------------- */

/* CovarianceIntersection::fuseCI, k-agent MSCKF form (src/x/ekf/ci.cpp:49-92):
 * S = (1/w0) H P H^T + sum_i (1/w) H_i P_i H_i^T, w0 = 1 - k w,
 * *w_result = 1/w0.  w outside (0,1] -> XK_EINVAL (the reference throws for
 * w>1, w==0, w<-1; the NLopt branch -1<=w<0 is out of scope). */
int xk_fuse_ci_msckf(xk_handle *h, const double *P, int ldp, int n, const double *H, int ldh, int m,
                     int k, const double *const *Ps, const int *ns, const double *const *Hs,
                     double w_other, double *S, int lds, double *w_result);

/* Pairwise SLAM form (ci.cpp:94-127): S = P_a/(1-w) + P_b/w in measurement
 * space, *w_result = 1/(1-w). */
int xk_fuse_ci_slam(xk_handle *h, const double *Pa, int lda, int na, const double *Ha, int ldha,
                    const double *Pb, int ldb, int nb, const double *Hb, int ldhb, int m,
                    double w_other, double *S, int lds, double *w_result);

/* MultiSlamUpdate::processOneMatch (src/x/vio/multi_slam_update.cpp:61-246):
 * 3-row landmark-difference residual between own SLAM feature `feature_id`
 * and the other agent's `o_feature_id`, chi2_3(0.9) gate, pairwise CI, and
 * the 3 diagonal 3x3 blocks of P_j scaled by w_result.  Outputs valid iff
 * *inlier: H (3 x n), res (3), S (3x3), P_j (n x n). */
int xk_multi_slam_match(xk_handle *h, const double *C_q_G, const double *G_p_C, int n_poses,
                        const double *feat, int anchor_idx, int feature_id, const double *P, int ldp,
                        int n, int n_poses_max, const double *o_C_q_G, const double *o_G_p_C,
                        int o_n_poses, const double *o_feat, int o_anchor_idx, int o_feature_id,
                        const double *o_P, int ldop, int no, int o_n_poses_max, double sigma_landmark,
                        double ci_slam_w, int *inlier, double *gamma, double *H, int ldh, double *res,
                        double *S, double *P_j, int ldpj);

/* MSCKF-MSCKF CI block of MsckfUpdate::preProcessOneTrack (src/x/vio/msckf_update.cpp:96-279)
 * for ONE of this agent's tracks that k other agents also observed (k <= 7):
 * joint triangulation over all agents' observations (matched agents first, self last,
 * :113-165), own single-agent gate (:172, :457-463), column-space rows of every agent
 * (:201-203, :439-443), null-space projection of the landmark (:207, :494-501),
 * S_j over all agents (:217-237), chi2(2*sum(L) - 3, 0.95) gate (:243-250), fixed-weight
 * fuseCI (:252-255) and the block scaling of P_j (:256-267).
 *   obs [L x 2], own window lists / P as in xk_stage_*;
 *   per matched agent i: m_obs[i] [m_L[i] x 2], m_q[i] [m_nposes[i] x 4], m_p[i]
 *   [m_nposes[i] x 3] (its full window lists; its track sees the LAST m_L[i] poses),
 *   m_P[i] [m_n[i] x m_n[i]] dense column-major, m_n[i] = 15 + 6*m_nposes[i] + 3*M_i.
 * Outputs: *self_inlier, *self_gamma; *has_ci; if *has_ci: H (3k x n, ldh), res (3k),
 * S (3k x 3k, lds), P_j (n x n, ldpj), feed them to xk_apply_ci.  H/res/S are defined up to
 * a common orthogonal factor (basis of the null space), H^T S^-1 H and H^T S^-1 res are not. */
int xk_msckf_ci_track(xk_handle *h, const double *obs, int L, const double *C_q_G, const double *G_p_C,
                      int n_poses, const double *P, int ldp, int n, int n_poses_max, double sigma_img, int k,
                      const double *const *m_obs, const int *m_L, const double *const *m_q,
                      const double *const *m_p, const int *m_nposes, const double *const *m_P,
                      const int *m_n, double ci_msckf_w, int *self_inlier, double *self_gamma, int *has_ci,
                      double *ci_gamma, double *H, int ldh, double *res, double *S, int lds, double *P_j,
                      int ldpj);

/* ---- MSCKF-SLAM update and persistent-feature initialisation ----
 * Tracks whose landmark becomes a persistent (SLAM) feature this frame (VioUpdater::constructUpdate,
 * vio_updater.cpp:311-321 -> MsckfSlamUpdate, msckf_slam_update.cpp:25-267).  Staged like the MSCKF tracks
 * (CSR of normalised observations, a length-L track sees the last L poses); xk_msckf_build then also builds
 * their null-space rows (stacked between the MSCKF and the SLAM rows, vio_updater.cpp:413-419) and keeps the
 * column-space rows H1, H2, r1 and the triangulated inverse-depth features on the device. */
int xk_stage_msckf_slam(xk_handle *h, const int *trk_off, const double *obs_xy, int K2);
/* After xk_msckf_build: per-track gate result and the MsckfSlamMatrices (types.h) -- H1 (3K2 x n, ldh1),
 * H2 (3K2 x 3K2 block diagonal, ldh2), r1 (3K2), features (3K2); any pointer may be NULL.  H1/H2/r1 are defined
 * up to an orthogonal 3x3 factor per track (basis of the column space); H2^-1 H1, H2^-1 r1, H2^-1 H2^-T are not. */
int xk_msckf_slam_results(xk_handle *h, int *inlier, double *gamma, double *H1, int ldh1, double *H2, int ldh2,
                          double *r1, double *features);
/* StateManager::initMsckfSlamFeatures + addFeatureStates (state_manager.cpp:151-174,199-226) after the update:
 * new_features (out, 3K2) = features - H2^-1 H1 correction + H2^-1 r1, and the covariance blocks of feature
 * slots n_features .. n_features + K2 - 1 of the RESIDENT (posterior) covariance are set to -H2^-1 H1 P (cross)
 * and H2^-1 H1 P (H2^-1 H1)^T + sigma_img^2 H2^-1 H2^-T.  XK_ESINGULAR if an H2 block is singular. */
int xk_init_msckf_slam_features(xk_handle *h, int n_features, const double *correction, double sigma_img,
                                double *new_features);
/* StateManager::initStandardSlamFeatures + addFeatureStates (state_manager.cpp:176-226): k uncorrelated new
 * features in slots n_features.., variances sigma_img^2, sigma_img^2, sigma_rho_0^2 (feature values:
 * SlamUpdate::computeInverseDepthsNew, slam_update.cpp:216-242, stay on the host). */
int xk_init_standard_slam_features(xk_handle *h, int n_features, int k, double sigma_img, double sigma_rho_0);

/* ---- StateManager::manage on the resident covariance (state_manager.cpp:31-149) ----
 * Every covariance operation of manage() -- persistent-feature removal (:52-112), anchor re-parametrisation
 * (reparametrizeFeatures, :351-482), window slide (slideWindow, :484-537) and pose augmentation
 * (augmentCovariance, :273-349) -- is  P <- J P J^T  with a J that is an identity / permutation / zero except
 * for a few 3-row blocks.  The reference forms J densely and does two n^3 products per operation; here J is
 * handed over in CSR (n rows, row_ptr[n+1], 0-based col_idx, val) and applied to the handle's RESIDENT
 * covariance in O(nnz(J)^2 / n + n^2).  host/src/state_manager.cpp builds the J's exactly as the reference. */
int xk_cov_congruence(xk_handle *h, const int *row_ptr, const int *col_idx, const double *val, int nnz);

/* Propagator::propagateCovarianceMatrices (src/x/ekf/propagator.cpp:166-205) on the resident covariance, one
 * IMU step (and, called in a loop, Ekf::repropagateFromStateAtIdx, ekf.cpp:227-252):
 *   P_ii <- F_d P_ii F_d^T + Q_d,  P_iv <- F_d P_iv,  P_vi <- P_vi F_d^T,  P_vv unchanged.
 * f_d, q_d: the 15 x 15 discrete transition and process-noise matrices (column-major, ld >= 15) that
 * Propagator::discreteStateTransition / discreteProcessNoiseCov (:110-164, :207-840) compute on the host. */
int xk_cov_propagate(xk_handle *h, const double *f_d, int ldf, const double *q_d, int ldq);

/* Device-resident CI round (MsckfUpdate::preProcessOneTrack CI block, msckf_update.cpp:96-279, followed by
 * Updater::applyCI per fused entry, updater.cpp:90-93,144-161) against the snapshots of the other agents as they
 * sit in the RCCL receive buffer -- no host staging of the n x n covariances.
 *   d_payloads   DEVICE [world][payload_stride]: xk_pack_payload layout of every agent (own slot unused)
 *   d_tracks     DEVICE [world][n_tracks][1 + 2N]: per shared track the length, then the observations (x,y)
 *   track_len    host [world][n_tracks], n_poses_valid host [world]: the lengths / window sizes found in the above
 *   self_track   host [n_tracks]: index of each shared track among this handle's staged tracks
 * The handle's staged window and its RESIDENT covariance are this agent's side.  Every entry is built from the
 * same prior and applyCI overwrites P each time (the reference's behaviour): on return the resident covariance
 * is the posterior of the last fused entry.  corrections (host, optional): [*n_fused][n]. */
int xk_ci_round_device(xk_handle *h, const double *d_payloads, long payload_stride, int world, int self_rank,
                       const double *d_tracks, int n_tracks, const int *track_len, const int *n_poses_valid,
                       const int *self_track, double sigma_img, double ci_msckf_w, int *n_fused, double *corrections);

/* ---- inter-agent payload (SimpleState, include/x/ekf/simple_state.h:33-35,
 * assembled at src/x/vio/vio.cpp:447-450) ------------------------------ */

/* Size in doubles of the fixed all-double payload for (N, M): hdr[8] dyn[16]
 * pos[3N] att[4N] feat[3M] anchors[M] cov[n*n]. */
long xk_payload_doubles(int n_poses_max, int n_feat_max);
/* Packs this agent's outgoing payload from the staged window/SLAM state and
 * the CURRENT device-resident P into d_dst (a DEVICE buffer of
 * xk_payload_doubles doubles owned by the caller, e.g. the RCCL send buffer)
 * or, if d_dst is NULL, into a buffer owned by the handle; *d_payload (if
 * non-NULL) receives the device pointer used.  dyn (16) = p,v,q xyzw,b_w,b_a (State::getDynamicStates,
 * state.cpp:87-99) comes from the host. */
int xk_pack_payload(xk_handle *h, double agent_id, double timestamp, const double *dyn16,
                    double *d_dst, double **d_payload);

/* ---- place-recognition request filter + keyframe store ---------------------------
 * The reference answers another agent's request by scoring the request's binary VLAD against its own keyframe
 * database and returning the best keyframe's SimpleState + tracks (VIO::processOtherRequests, vio.cpp:462-496 ->
 * PlaceRecognition::findPlace, place_recognition.cpp:677-683 -> Database::findCandidate, database.cpp:30-49); the
 * requester then matches the returned descriptors against its own (findCorrespondences, place_recognition.cpp:249).
 * The keyframes (payload in xk_pack_payload layout, packed tracks, descriptors, VLAD) stay in device memory, so a
 * response is sent straight from HBM.  Descriptors are rows of desc_bytes bytes (ORB: 32). */
typedef struct xk_pr xk_pr;

/* Vocabulary = a DBoW3 tree (PRVocabulary, types.h:33): k, L, node descriptors [n_nodes][desc_bytes], children
 * [n_nodes][kmax] (-1 padded, in file order), word_of_node [n_nodes] (-1 for inner nodes), node_of_word [n_words].
 * payload_doubles / tracks_doubles: sizes of the per-keyframe device buffers; max_desc: most descriptors per call. */
int xk_pr_create(xk_handle *h, int k, int L, int n_nodes, int kmax, int desc_bytes, const unsigned char *node_desc,
                 const int *children, const int *word_of_node, const int *node_of_word, int n_words,
                 long payload_doubles, long tracks_doubles, int max_desc, xk_pr **out);
void xk_pr_destroy(xk_pr *p);
int xk_pr_vlad_bytes(const xk_pr *p);   /* k^L * desc_bytes (vlad.cpp:27-31: v_length_ / 8) */
int xk_pr_size(const xk_pr *p);         /* keyframes in the store (<= 15, database.h:70) */

/* VLAD::computeVLAD (vlad.cpp:40-66) / Database::computeVLAD (database.cpp:26-28): desc HOST [n][desc_bytes] ->
 * vlad_out HOST [xk_pr_vlad_bytes]. */
int xk_pr_compute_vlad(xk_pr *p, const unsigned char *desc, int n, unsigned char *vlad_out);

/* Database::addKeyframe (database.cpp:51-61): VLAD of the keyframe's descriptors (Keyframe::getDescriptors order:
 * MSCKF, SLAM, OPP tracks, keyframe.cpp:40-52), stored with the DEVICE payload / tracks buffers (copied; may be NULL)
 * and a caller tag; the oldest keyframe is dropped beyond 15. */
int xk_pr_add_keyframe(xk_pr *p, const unsigned char *desc, int n_desc, const double *d_payload, const double *d_tracks,
                       long tag);

/* Database::findCandidate (database.cpp:30-49): best-scoring keyframe with score > pr_score_thr that has not yet
 * been sent to `uav_id` (Keyframe::findOtherUavId); marks it as sent.  *index = position in the store, oldest
 * first, or -1; *score = (v_length - hamming) / v_length of the winner (VLAD::computeScore, vlad.cpp:68-75). */
int xk_pr_find_candidate(xk_pr *p, int uav_id, const unsigned char *query_vlad, double pr_score_thr, int *index,
                         double *score, long *tag);

/* The stored keyframe `index`: DEVICE pointers of its payload / tracks (what VIO::processOtherRequests hands back,
 * vio.cpp:489-495), its descriptor count and tag; desc_out (HOST, optional) receives the descriptors. */
int xk_pr_keyframe(xk_pr *p, int index, const double **d_payload, const double **d_tracks, int *n_desc, long *tag,
                   unsigned char *desc_out);

/* Copies the stored keyframe's payload / tracks into caller-owned DEVICE buffers (e.g. the RCCL send buffer of the
 * response) and waits for the copy. */
int xk_pr_copy_keyframe(xk_pr *p, int index, double *d_payload_dst, double *d_tracks_dst);

/* matcher_->knnMatch(query = received, train = current, k = 2) with NORM_HAMMING (place_recognition.cpp:68-69,249):
 * idx / dist HOST [nq][2], ascending (distance, train index); idx = -1 where the train set is too short. */
int xk_pr_knn_match(xk_pr *p, const unsigned char *query, int nq, const unsigned char *train, int nt, int *idx, int *dist);

/* xk_msckf_build + xk_qr_compress queued on the handle's stream with NO host synchronisation and no host outputs:
 * together with the non-blocking staging calls and xk_cov_congruence / xk_cov_propagate, a whole frame -- covariance
 * propagation, StateManager::manage, per-feature build, QR compression, Kalman update -- is queued back to back and
 * xk_apply_update's single synchronisation brings the correction, the status and the gate results back.
 * For callers that have something between constructUpdate and applyUpdate that rewrites the covariance -- the applyCI entries of
 * the MULTI_UAV order (updater.cpp:84-97).  [T_H | z] does not depend on the covariance once the gates have read the prior (the
 * per-feature kernel, queued here), so where the single launch can take the Kalman update along (narrow systems, n <= 206) the
 * compression itself is queued by xk_apply_update, behind those entries, with the update inside: one launch there instead of one
 * here and five there.  xk_qr_compress / xk_fetch_flags behave as before: an xk_qr_compress between this call and xk_apply_update
 * runs the compression then and there, and xk_apply_update applies the [T_H | z] it left (no second compression).  No STAGING call
 * (xk_stage_*, xk_msckf_build) may come between the two: it would replace the rows the deferred compression is to read. */
int xk_build_compress_async(xk_handle *h, double sigma_img);

/* xk_build_compress_async with the Kalman update of Updater::applyUpdate(correction_total = 0, cov_update = true)
 * (updater.cpp:117-141) queued as well: inside the compression launch where the geometry allows it (windows of up to 31 poses
 * without persistent features: the update is applied block by block as the panels of the QR complete and costs the launch
 * ~5 us), behind it otherwise.  xk_apply_update(h, NULL, 1, correction) then only waits for the result; any other
 * correction_total / cov_update there is XK_EINVAL.  For the single-agent order with iekf_iter = 1 (updater.cpp:99-110); NOT for
 * the MULTI_UAV order, whose applyCI entries replace the covariance between constructUpdate and applyUpdate (:84-97). */
int xk_build_compress_update_async(xk_handle *h, double sigma_img);

/* The same for ONE PASS of the iterated update (updater.cpp:99-110, iekf_iter > 1): both arguments of the applyUpdate that
 * follows are known when constructUpdate is called -- correction_total is what the passes so far accumulated (:140), cov_update is
 * "this is the last pass" -- so the pass is queued whole: corr = K (res + H corr_total) - corr_total (:126-128), the covariance
 * updated only if cov_update.  corr_total: n doubles on the host, NULL = zeros.  xk_apply_update(h, the same corr_total, the same
 * cov_update, correction) then only waits; other arguments there are XK_EINVAL and leave the queued pass collectable. */
int xk_build_compress_update_pass_async(xk_handle *h, double sigma_img, const double *corr_total, int cov_update);
/* The gate results of the last build (any pointer may be NULL); synchronises the stream if it is still busy. */
int xk_fetch_flags(xk_handle *h, int *inlier_msckf, double *gamma_msckf, int *inlier_slam, double *gamma_slam);

/* Updater::applyCI (updater.cpp:144-161) on the RESIDENT covariance: P <- sym((I - K H) ci_P), K = ci_P H^T S^-1,
 * replaces the handle's covariance and stays on the device; only the n-vector correction comes back.  A compressed
 * [T_H | z] waiting for xk_apply_update is left alone, so the reference's order -- constructUpdate, the applyCI loop,
 * then applyUpdate on the post-CI covariance (updater.cpp:84-97) -- needs no covariance transfer. */
int xk_apply_ci_resident(xk_handle *h, const double *ci_P, int ldc, int n, const double *H, int ldh, int m,
                         const double *res, const double *S, int lds, double *correction);

/* Save (restore = 0) / bring back (restore = 1) a device-side copy of the resident covariance; 2 / 3: the same on a second slot,
 * which the filter loop (x::Ekf with a resident covariance) keeps for itself: the prior of an update that the IMU thread may lap
 * (Ekf::repropagateFromStateAtIdx, ekf.cpp:229-239, discards such an update). */
int xk_snapshot_P(xk_handle *h, int restore);

/* ---- measurement ----------------------------------------------------- */

#define XK_NSTAGE 6
/* stage order: 0 msckf_feature, 1 slam_rows, 2 caqr_panel0 (first per-tile panel launch), 3 caqr_rest,
 * 4 kalman_update, 5 unused */
typedef struct {
  float total_ms;               /* one staged update, HIP events on the handle's stream */
  float stage_ms[XK_NSTAGE];    /* per update, summed over the stage's launches */
  int stage_launches[XK_NSTAGE];
  char stage_name[XK_NSTAGE][32];
  int n, c1, k_tracks, rows_stacked, n_leaf, n_levels;
} xk_timing;

/* Runs `steps` staged visual updates back to back on the handle's stream
 * (each from the same staged prior; results of the last one stay resident)
 * and reports HIP-event timings averaged per update. */
int xk_bench_staged(xk_handle *h, double sigma_img, int warmup, int steps, xk_timing *out);

/* Enqueues `steps` sequential staged visual updates (same staged prior, the
 * posterior of the last one left in the handle's output buffer) and waits for
 * them; nothing crosses PCIe.  This is the timed region of bench.py. */
int xk_run_steps(xk_handle *h, double sigma_img, int steps);

/* Which schedule compressed the last update -- 0 the multi-launch CAQR, 2 the pipelined single launch (1 was round 2's
 * register-resident kernel, no longer built), 3 the multi-launch CAQR for the first panels of a tall system (windows of 34..64
 * poses) and one or two single launches for its last <= 192 columns, 4 no compression at all: the stack was the SLAM features' rows alone
 * (2 M rows against n > 3 M columns) or a small stack whose nominal rows are at most n -- the reference compresses only when rows >
 * columns, vio_updater.cpp:487, and neither does this: the rows go to the update as built -- whether the single-launch path is armed for the next update, how many launches have given up on
 * this handle so far (workgroups not co-resident: another process on the GPU, a CU mask) and the reason code of the last one
 * (2 XCD-local hand-off, 3 uneven XCD placement, 4 / 5 / 6 waiting for the last level / the roots / the tiles, 8 the Kalman role
 * waiting for rows of R, 9 more rows passed the gates than the tiles of the launch hold -- not a co-residency problem: the fast
 * path stays armed for smaller stacks).  A launch that
 * gives up costs one bounded retry (<= 2 ms) and the update is redone by the multi-launch schedule with the same result; the
 * handle tries the fast path again after "caqr_rearm" (64) clean updates, doubling that distance at every further give-up.
 * xk_last_error() carries the same information as text.  Any pointer may be NULL. */
int xk_caqr_status(const xk_handle *h, int *schedule, int *armed, int *giveups, int *last_reason);

/* Operational switches of the compression on a live handle.  "caqr_resident": 0 = the multi-launch schedule serves every update
 * (e.g. a GPU this process knowingly shares), 1 (default) = the single launch where the shape allows it; "caqr_rearm": clean
 * multi-launch updates after which a single-launch path that gave up is tried again (default 64, doubling at every further give-up);
 * "caqr_tail": 0 = tall systems (windows of 34..64 poses) are factored by the multi-launch schedule to the last panel, 1 (default) = their
 * last <= 192 columns by one or two single launches, 2 = their last <= 96 columns by one; "slam_split": 1 (default) = systems with SLAM
 * features and more than 206 error states compress only the tracks' rows, in the pose columns, and append the features' own rows to the
 * compressed system as built, and a stack of SLAM rows alone or of at most n nominal rows goes to the update uncompressed (same posterior;
 * xk_qr_compress keeps returning the whole stack's upper-triangular T_H), 0 = the whole stack is compressed every time.
 * Unknown name: XK_EINVAL.  The release library reads nothing from the environment; the experiment switches, test hooks, debug
 * exports and probe kernels of the lab build are declared in xk_lab.h.  No counterpart in the reference. */
int xk_set_option(xk_handle *h, const char *name, int value);

#ifdef __cplusplus
}
#endif
#endif /* XK_H_ */
