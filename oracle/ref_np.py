"""NumPy/SciPy float64 restatement of the xVIO EKF-update hot path.

TEST INFRASTRUCTURE ONLY.  Nothing outside tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this module; the product path
(x_multi_agent_amd/) never does.

PARITY UNPINNED: the reference (jpl-x/x_multi_agent @ v1) ships no tests, no
golden vectors, and cannot be built here (needs Eigen3, OpenCV, Boost -- none
installed, no network).  This file is one of two independent restatements
(the other is oracle/xk_oracle.c); their mutual agreement plus the
mathematical invariants in tests/ is what stands in for reference fixtures.

Every function cites the reference file:line it follows (paths relative to
/root/reference).  The third-party arithmetic the reference calls into is
restated from its published definition:
  * Eigen3 (unpinned, CMakeLists.txt:181): HouseholderQR, .inverse(),
    Quaternion::toRotationMatrix, AngleAxis -> numpy.linalg.{qr,inv}
  * OpenCV >= 3.3.1 calib3d cv::triangulatePoints (CMakeLists.txt:101) ->
    null vector of the 4x4 DLT system via numpy.linalg.svd
  * Boost.Math >= 1.71 quantile(chi_squared) (CMakeLists.txt:117) ->
    scipy.stats.chi2.ppf
"""
import numpy as np
from scipy.stats import chi2 as _chi2

K_CORE = 15  # kSizeCoreErr, include/x/common/types.h:45
GRAV = np.array([0.0, 0.0, -9.81])  # hard-coded, msckf_update.cpp:393


# ----------------------------------------------------------------------------
# helpers (include/x/vio/tools.h:57-84, Eigen semantics)
# ----------------------------------------------------------------------------
def skew(v):
    """x::Skew, include/x/vio/tools.h:57-65."""
    x, y, z = v
    return np.array([[0.0, -z, y], [z, 0.0, -x], [-y, x, 0.0]])


def quat_to_rot(q_xyzw):
    """q.normalized().toRotationMatrix() for q stored (x,y,z,w).

    Maps camera -> world (msckf_update.cpp:339-340, state.cpp:190,194)."""
    q = np.asarray(q_xyzw, dtype=np.float64)
    q = q / np.sqrt(np.dot(q, q))
    x, y, z, w = q
    tx, ty, tz = 2 * x, 2 * y, 2 * z
    twx, twy, twz = tx * w, ty * w, tz * w
    txx, txy, txz = tx * x, ty * x, tz * x
    tyy, tyz, tzz = ty * y, tz * y, tz * z
    return np.array([
        [1 - (tyy + tzz), txy - twz, txz + twy],
        [txy + twz, 1 - (txx + tzz), tyz - twx],
        [txz - twy, tyz + twx, 1 - (txx + tyy)],
    ])


def chi2inv(p, dof):
    """boost::math::quantile(chi_squared(dof), p)."""
    return float(_chi2.ppf(p, dof))


# ----------------------------------------------------------------------------
# Triangulation (src/x/vision/triangulation.cpp)
# ----------------------------------------------------------------------------
def pose2proj(rot, pos):
    """triangulation.cpp:208-216: [R | -R p] with R = world->camera."""
    return np.hstack([rot, (-rot @ pos).reshape(3, 1)])


def triangulate_dlt(obs1, obs2, proj1, proj2):
    """triangulation.cpp:81-100 + cv::triangulatePoints (DLT null vector)."""
    a = np.empty((4, 4))
    a[0] = obs1[0] * proj1[2] - proj1[0]
    a[1] = obs1[1] * proj1[2] - proj1[1]
    a[2] = obs2[0] * proj2[2] - proj2[0]
    a[3] = obs2[1] * proj2[2] - proj2[1]
    _, _, vt = np.linalg.svd(a)
    x = vt[3]
    return x[:3] / x[3]


def triangulate_gn(quats_xyzw, poss, obs, max_iter=10, term=1e-5):
    """triangulation.cpp:48-79 (ctor) + :102-206 (triangulateGN).

    quats/poss/obs all have the same length (the track's poses).  Returns
    (alpha, beta, rho) anchored in the LAST pose, and the iteration count."""
    n = len(poss)
    rots = [quat_to_rot(q).T for q in quats_xyzw]  # world -> camera, :70
    projs = [pose2proj(rots[i], np.asarray(poss[i], float)) for i in range(n)]
    i2 = n - 1
    i1 = i2 - len(obs) + 1  # == 0
    pt = triangulate_dlt(obs[i1], obs[i2], projs[i1], projs[i2])
    pc2 = projs[i2] @ np.append(pt, 1.0)
    alpha, beta, rho = pc2[0] / pc2[2], pc2[1] / pc2[2], 1.0 / pc2[2]
    rot_a, p_a = rots[i2], np.asarray(poss[i2], float)
    r_norm_last, r_norm, it = 1000.0, 100.0, 0
    while r_norm_last - r_norm > term:  # :149
        it += 1
        if it > max_iter:
            break
        r = np.zeros(2 * n)
        j = np.zeros((2 * n, 3))
        for i in range(i1, i2 + 1):
            rot = rots[i]
            drot = rot @ rot_a.T
            dpos = rot @ p_a - rot @ np.asarray(poss[i], float)
            k = i - i1
            h_i = drot @ np.array([alpha, beta, 1.0]) + rho * dpos
            h = np.array([h_i[0] / h_i[2], h_i[1] / h_i[2]])
            r[2 * k:2 * k + 2] = np.asarray(obs[k]) - h
            j0 = np.column_stack([drot[:, 0], drot[:, 1], dpos])
            j1 = np.array([[-1.0 / h_i[2], 0.0, h_i[0] / h_i[2] ** 2],
                           [0.0, -1.0 / h_i[2], h_i[1] / h_i[2] ** 2]])
            j[2 * k:2 * k + 2] = j1 @ j0
        delta = np.linalg.inv(j.T @ j) @ j.T @ r  # :193-194
        alpha, beta, rho = alpha - delta[0], beta - delta[1], rho - delta[2]
        r_norm_last, r_norm = r_norm, np.linalg.norm(r)
    return np.array([alpha, beta, rho]), it


def global_feature_position(ivd, q_last, p_last):
    """msckf_update.cpp:283-304."""
    a, b, rho = ivd
    return (1.0 / rho) * quat_to_rot(q_last) @ np.array([a, b, 1.0]) + np.asarray(p_last, float)


# ----------------------------------------------------------------------------
# MSCKF per-track measurement (src/x/vio/msckf_update.cpp:306-492)
# ----------------------------------------------------------------------------
def msckf_track_jacobians(obs, C_q_G, G_p_C, n_poses_max, G_p_f, n_cols):
    """Observation loop of processOneTrack, msckf_update.cpp:328-417.

    Returns (jac_j [2L x n_cols], Hf_j [2L x 3], res_j [2L]) or None if a
    non-finite camera-frame point is met (:349-357)."""
    L = len(obs)
    n_poses = len(G_p_C)
    jac = np.zeros((2 * L, n_cols))
    hf = np.zeros((2 * L, 3))
    res = np.zeros(2 * L)
    for i in range(L):
        pos = n_poses - L + i  # :329-331
        R = quat_to_rot(C_q_G[pos])
        p = np.asarray(G_p_C[pos], float)
        c = R.T @ (G_p_f - p)
        if not np.all(c == c):
            return None
        res[2 * i] = obs[i][0] - c[0] / c[2]
        res[2 * i + 1] = obs[i][1] - c[1] / c[2]
        Ji = np.array([[1.0 / c[2], 0.0, -c[0] / c[2] ** 2],
                       [0.0, 1.0 / c[2], -c[1] / c[2] ** 2]])
        Jp = -Ji @ R.T
        Ja = Ji @ skew(c)
        # observability constraint, :393-406
        u = (R @ GRAV).reshape(3, 1)
        Jp = Jp - Jp @ u @ np.linalg.inv(u.T @ u) @ u.T
        u = (skew(G_p_f - p) @ GRAV).reshape(3, 1)
        Ja = Ja - Ja @ u @ np.linalg.inv(u.T @ u) @ u.T
        hf[2 * i:2 * i + 2] = -Jp  # :409
        col = K_CORE + 3 * pos
        jac[2 * i:2 * i + 2, col:col + 3] = Jp
        col += 3 * n_poses_max
        jac[2 * i:2 * i + 2, col:col + 3] = Ja
    return jac, hf, res


def left_nullspace(hf):
    """msckf_update.cpp:423-427: Q = householderQr(Hf).householderQ(); A_up =
    Q[:, :3], A = Q[:, 3:].  (Basis is implementation-defined)"""
    q, _ = np.linalg.qr(hf, mode="complete")
    return q[:, :3], q[:, 3:]


def msckf_process_one_track(obs, P, C_q_G, G_p_C, n_poses_max, var_img, G_p_f):
    """processOneTrack single-agent branch, msckf_update.cpp:306-492.

    Returns dict(valid, inlier, gamma, chi, jac0, res0, A_up_jac, A_up_hf,
    A_up_res)."""
    n = P.shape[1]
    out = msckf_track_jacobians(obs, C_q_G, G_p_C, n_poses_max, G_p_f, n)
    if out is None:
        return dict(valid=False, inlier=False, gamma=np.nan)
    jac, hf, res = out
    a_up, a = left_nullspace(hf)
    res0 = a.T @ res
    jac0 = a.T @ jac
    L = len(obs)
    d = 2 * L - 3
    s = jac0 @ P @ jac0.T + var_img * np.eye(d)  # :457
    gamma = float(res0 @ np.linalg.inv(s) @ res0)
    chi = chi2inv(0.95, 2 * L - 3)  # :459-461
    return dict(valid=True, inlier=gamma < chi, gamma=gamma, chi=chi,
                jac0=jac0, res0=res0,
                up_jac=a_up.T @ jac, up_hf=a_up.T @ hf, up_res=a_up.T @ res)


def msckf_update(tracks, C_q_G, G_p_C, P, n_poses_max, sigma_img):
    """MsckfUpdate ctor, single-agent, msckf_update.cpp:27-63 + :65-173.

    tracks: list of (L_k x 2) arrays of normalised observations; a length-L
    track observes the LAST L poses of the window lists.
    Returns (jac [rows x n], res [rows], cov_diag [rows], info) with rows
    pre-sized for all tracks and trailing zero rows for rejected ones (Q1)."""
    n_trks = len(tracks)
    n_obs = sum(len(t) for t in tracks)
    rows = 2 * n_obs - 3 * n_trks
    n = P.shape[1]
    jac = np.zeros((rows, n))
    cov_diag = np.ones(rows)
    res = np.zeros(rows)
    var_img = sigma_img * sigma_img
    row_h = 0
    inlier = np.zeros(n_trks, dtype=np.int32)
    gamma = np.full(n_trks, np.nan)
    feats = np.zeros((n_trks, 3))
    gn_iters = np.zeros(n_trks, dtype=np.int32)
    for k, trk in enumerate(tracks):
        L = len(trk)
        q_l = C_q_G[len(C_q_G) - L:]
        p_l = G_p_C[len(G_p_C) - L:]
        ivd, it = triangulate_gn(q_l, p_l, trk)
        gn_iters[k] = it
        gpf = global_feature_position(ivd, q_l[-1], p_l[-1])
        feats[k] = gpf
        o = msckf_process_one_track(trk, P, C_q_G, G_p_C, n_poses_max, var_img, gpf)
        gamma[k] = o["gamma"]
        if o["valid"] and o["inlier"]:
            d = 2 * L - 3
            jac[row_h:row_h + d] = o["jac0"]
            res[row_h:row_h + d] = o["res0"]
            cov_diag[row_h:row_h + d] = var_img
            row_h += d
            inlier[k] = 1
    return jac, res, cov_diag, dict(inlier=inlier, gamma=gamma, feats=feats,
                                    gn_iters=gn_iters, rows_used=row_h)


# ----------------------------------------------------------------------------
# SLAM measurement (src/x/vio/slam_update.cpp:25-214)
# ----------------------------------------------------------------------------
def slam_process_one_track(track_size, z_last, j, C_q_G, G_p_C, feat, anchor_idxs, P,
                           n_poses_max, var_img):
    """SlamUpdate::processOneTrack, slam_update.cpp:49-214."""
    n = P.shape[1]
    h = np.zeros((2, n))
    alpha, beta, rho = feat[3 * j:3 * j + 3]
    a = anchor_idxs[j]
    Ra = quat_to_rot(C_q_G[a])
    pa = np.asarray(G_p_C[a], float)
    gpf = (1.0 / rho) * Ra @ np.array([alpha, beta, 1.0]) + pa
    Rn = quat_to_rot(C_q_G[-1])
    pn = np.asarray(G_p_C[-1], float)
    c = Rn.T @ (gpf - pn)
    res = np.array([z_last[0] - c[0] / c[2], z_last[1] - c[1] / c[2]])
    pos = len(C_q_G) - 1
    fcol = K_CORE + (2 * n_poses_max + j) * 3
    if a == pos:  # :120-131
        h[0, fcol] = 1.0
        h[1, fcol + 1] = 1.0
    else:
        Ji = np.array([[1.0 / c[2], 0.0, -c[0] / c[2] ** 2],
                       [0.0, 1.0 / c[2], -c[1] / c[2] ** 2]])
        J_att = Ji @ skew(c)
        J_pos = -Ji @ Rn.T
        J_anchor_att = -1.0 / rho * Ji @ Rn.T @ Ra @ skew([alpha, beta, 1.0])
        J_anchor_pos = -J_pos
        mat = np.eye(3)
        mat[0, 2], mat[1, 2], mat[2, 2] = -alpha / rho, -beta / rho, -1.0 / rho
        Hf = 1.0 / rho * Ji @ Rn.T @ Ra @ mat
        col = K_CORE + 3 * pos
        h[:, col:col + 3] = J_pos
        h[:, col + 3 * n_poses_max:col + 3 * n_poses_max + 3] = J_att
        col = K_CORE + 3 * a
        h[:, col:col + 3] = J_anchor_pos
        h[:, col + 3 * n_poses_max:col + 3 * n_poses_max + 3] = J_anchor_att
        h[:, fcol:fcol + 3] = Hf
    s = h @ P @ h.T + var_img * np.eye(2)
    gamma = float(res @ np.linalg.inv(s) @ res)
    chi = chi2inv(0.9, 2 * track_size)  # :196-197
    return h, res, gamma, gamma < chi


def slam_update(track_sizes, z_last, C_q_G, G_p_C, feat, anchor_idxs, P, n_poses_max, sigma_img):
    """SlamUpdate ctor, slam_update.cpp:25-47."""
    m = len(track_sizes)
    n = P.shape[1]
    jac = np.zeros((2 * m, n))
    res = np.zeros(2 * m)
    cov_diag = np.ones(2 * m)
    var_img = sigma_img ** 2
    row = 0
    inl = np.zeros(m, dtype=np.int32)
    gam = np.zeros(m)
    for j in range(m):
        h, r, g, ok = slam_process_one_track(track_sizes[j], z_last[j], j, C_q_G, G_p_C, feat,
                                             anchor_idxs, P, n_poses_max, var_img)
        gam[j] = g
        if ok:
            jac[row:row + 2] = h
            res[row:row + 2] = r
            cov_diag[row:row + 2] = var_img
            row += 2
            inl[j] = 1
    return jac, res, cov_diag, dict(inlier=inl, gamma=gam)


# ----------------------------------------------------------------------------
# QR compression + Kalman update (vio_updater.cpp:487-512, updater.cpp:117-161)
# ----------------------------------------------------------------------------
def apply_qr_decomposition(h, res, r_diag, sigma_img):
    """VioUpdater::applyQRDecomposition, vio_updater.cpp:487-512.

    r_diag is the measurement-noise DIAGONAL (the reference densifies it to
    rows x rows at vio_updater.cpp:417; result-identical).  Returns
    (h, res, r_diag, did_qr)."""
    rows, cols = h.shape
    if rows > cols + 1:
        hres = np.hstack([h, res.reshape(-1, 1)])
        thz = np.linalg.qr(hres, mode="r")
        return (thz[:cols, :cols].copy(), thz[:cols, cols].copy(),
                np.full(cols, sigma_img ** 2), True)
    return h, res, r_diag, False


def apply_update(P, H, res, r_diag, correction_total=None, cov_update=True):
    """Updater::applyUpdate, updater.cpp:117-141 (covariance + correction;
    State::correct is separate).  Returns (P_new, correction)."""
    n = P.shape[0]
    if correction_total is None:
        correction_total = np.zeros(n)
    S = H @ P @ H.T + np.diag(r_diag)
    K = P @ H.T @ np.linalg.inv(S)
    corr = K @ (res + H @ correction_total) - correction_total
    if cov_update:
        Pn = (np.eye(n) - K @ H) @ P
        Pn = 0.5 * (Pn + Pn.T)
    else:
        Pn = P.copy()
    return Pn, corr


def apply_ci(ci_P, H, res, S):
    """Updater::applyCI, updater.cpp:144-161.  Returns (P_new, correction)."""
    n = ci_P.shape[0]
    K = ci_P @ H.T @ np.linalg.inv(S)
    corr = K @ res
    Pn = (np.eye(n) - K @ H) @ ci_P
    return 0.5 * (Pn + Pn.T), corr


def error_quat(dtheta):
    """State::errorQuatFromSmallAngles, state.cpp:273-283 -> (x,y,z,w)."""
    nrm = np.sqrt(np.dot(dtheta, dtheta))
    if nrm == 0.0:
        return np.array([0.0, 0.0, 0.0, 1.0])
    ax = dtheta / nrm
    s = np.sin(0.5 * nrm)
    return np.array([ax[0] * s, ax[1] * s, ax[2] * s, np.cos(0.5 * nrm)])


def quat_mul(a, b):
    """Hamilton product, both (x,y,z,w) (Eigen operator*)."""
    ax, ay, az, aw = a
    bx, by, bz, bw = b
    return np.array([
        aw * bx + ax * bw + ay * bz - az * by,
        aw * by + ay * bw + az * bx - ax * bz,
        aw * bz + az * bw + ax * by - ay * bx,
        aw * bw - ax * bx - ay * by - az * bz,
    ])


def state_correct(state, corr):
    """State::correct, state.cpp:197-249.  state: dict with p,v,q(xyzw),b_w,
    b_a,p_array(3N),q_array(4N xyzw),f_array(3M).  Returns a new dict."""
    s = {k: np.array(v, dtype=np.float64, copy=True) for k, v in state.items()}
    npa = s["p_array"].size
    nf = s["f_array"].size
    s["p"] += corr[0:3]
    s["v"] += corr[3:6]
    s["b_w"] += corr[9:12]
    s["b_a"] += corr[12:15]
    s["p_array"] += corr[K_CORE:K_CORE + npa]
    s["f_array"] += corr[K_CORE + 2 * npa:K_CORE + 2 * npa + nf]
    q = quat_mul(s["q"], error_quat(corr[6:9]))
    s["q"] = q / np.sqrt(np.dot(q, q))
    dth = corr[K_CORE + npa:K_CORE + 2 * npa]
    for i in range(npa // 3):
        qi = quat_mul(s["q_array"][4 * i:4 * i + 4], error_quat(dth[3 * i:3 * i + 3]))
        s["q_array"][4 * i:4 * i + 4] = qi / np.sqrt(np.dot(qi, qi))
    return s


def visual_update(tracks, C_q_G, G_p_C, P, n_poses_max, sigma_img, slam=None, msckf_slam_tracks=None):
    """constructUpdate + applyUpdate for MSCKF (+ optional MSCKF-SLAM and SLAM) rows,
    vio_updater.cpp:267-423 + updater.cpp:99-110 with iekf_iter = 1 (visual_update_iekf below runs the loop).

    slam: None or dict(track_sizes, z_last, feat, anchor_idxs).
    msckf_slam_tracks: None or list of observation arrays (tracks whose feature becomes persistent);
    their rows are stacked between the MSCKF and the SLAM rows (:413-419).
    Returns dict(P, correction, inlier, gamma, h, res, did_qr, ...)."""
    jac, res, cov, info = msckf_update(tracks, C_q_G, G_p_C, P, n_poses_max, sigma_img)
    out = dict(msckf=info)
    if msckf_slam_tracks:
        jm, rm, cm, minfo, init_mats = msckf_slam_update(msckf_slam_tracks, C_q_G, G_p_C, P, n_poses_max, sigma_img)
        jac = np.vstack([jac, jm])
        res = np.concatenate([res, rm])
        cov = np.concatenate([cov, cm])
        out["msckf_slam"] = minfo
        out["init_mats"] = init_mats
    if slam is not None:
        js, rs, cs, sinfo = slam_update(slam["track_sizes"], slam["z_last"], C_q_G, G_p_C,
                                        slam["feat"], slam["anchor_idxs"], P, n_poses_max,
                                        sigma_img)
        jac = np.vstack([jac, js])
        res = np.concatenate([res, rs])
        cov = np.concatenate([cov, cs])
        out["slam"] = sinfo
    out["h_stack"], out["res_stack"] = jac, res
    h, r, cv, did = apply_qr_decomposition(jac, res, cov, sigma_img)
    out.update(h=h, res=r, r_diag=cv, did_qr=did)
    if h.size > 0:
        Pn, corr = apply_update(P, h, r, cv)
    else:
        Pn, corr = P.copy(), np.zeros(P.shape[0])
    out.update(P=Pn, correction=corr)
    return out


def visual_update_iekf(state, tracks, n_poses, P, n_poses_max, sigma_img, iekf_iter, slam=None):
    """The IEKF loop of Updater::update, updater.cpp:99-110 (single-agent build; MULTI_UAV has no loop, Q9).

    for i in range(iekf_iter):
        constructUpdate(state, h, res, r)   -- vio_updater.cpp:267-423: window lists re-read from the state as
                                               corrected so far (state_manager.cpp:539-584), rows built and gated
                                               against state.getCovariance(), which is the PRIOR until the last pass
        if h.size() > 0: applyUpdate(state, h, res, r, correction, is_last_iter)   -- updater.cpp:117-141:
            corr = K (res + H correction_total) - correction_total; covariance only when is_last_iter;
            state.correct(corr); correction_total += corr

    state: dict p, v, q, b_w, b_a, p_array (3 n_poses_max), q_array (4 n_poses_max, xyzw), f_array (3 M_cap).
    slam: None or dict(track_sizes, z_last, anchor_idxs) -- the feature states are read from state["f_array"].
    Returns dict(P, correction (= correction_total, what postUpdate receives), state, inlier (last pass), passes)."""
    n = P.shape[0]
    st = {k: np.array(v, dtype=np.float64, copy=True) for k, v in state.items()}
    ctot = np.zeros(n)
    Pn = P.copy()
    passes = []
    last = None
    for it in range(iekf_iter):
        is_last = it == iekf_iter - 1
        C_q_G = st["q_array"].reshape(-1, 4)[:n_poses].copy()
        G_p_C = st["p_array"].reshape(-1, 3)[:n_poses].copy()
        jac, res, cov, info = msckf_update(tracks, C_q_G, G_p_C, P, n_poses_max, sigma_img)
        sinfo = None
        if slam is not None:
            js, rs, cs, sinfo = slam_update(slam["track_sizes"], slam["z_last"], C_q_G, G_p_C, st["f_array"],
                                            slam["anchor_idxs"], P, n_poses_max, sigma_img)
            jac = np.vstack([jac, js])
            res = np.concatenate([res, rs])
            cov = np.concatenate([cov, cs])
        h, r, cv, _ = apply_qr_decomposition(jac, res, cov, sigma_img)
        last = dict(msckf=info, slam=sinfo)
        if h.size > 0:
            Pn, corr = apply_update(P, h, r, cv, ctot, is_last)
            st = state_correct(st, corr)
            ctot = ctot + corr
            passes.append(corr)
    return dict(P=Pn, correction=ctot, state=st, inlier=last["msckf"]["inlier"],
                inlier_slam=None if last["slam"] is None else last["slam"]["inlier"], passes=passes)


# ----------------------------------------------------------------------------
# Covariance intersection (src/x/ekf/ci.cpp:49-127, fixed-weight branches)
# ----------------------------------------------------------------------------
def _check_w(w):
    if w > 1.0 or w == 0 or w < -1:
        raise RuntimeError("The CI weights must be lower than 1.0 and larger 0.0")
    if w < 0.0:
        raise NotImplementedError("NLopt weight optimisation (ci.cpp:143-190) is out of scope")


def fuse_ci_msckf(cov_a, H_curr, other_covs, Hs, w_other):
    """CovarianceIntersection::fuseCI (k agents), ci.cpp:49-92."""
    _check_w(w_other)
    k = len(Hs)
    w0 = 1.0 - k * w_other
    S = (1.0 / w0) * H_curr @ cov_a @ H_curr.T
    for Pi, Hi in zip(other_covs, Hs):
        S = S + (1.0 / w_other) * Hi @ Pi @ Hi.T
    return S, 1.0 / w0


def fuse_ci_slam(cov_a, H_a, cov_b, H_b, w_other):
    """CovarianceIntersection::fuseCI (pairwise), ci.cpp:94-127."""
    _check_w(w_other)
    Pa = H_a @ cov_a @ H_a.T
    Pb = H_b @ cov_b @ H_b.T
    return (1.0 / (1.0 - w_other)) * Pa + (1.0 / w_other) * Pb, 1.0 / (1.0 - w_other)


def multi_slam_match(C_q_G, G_p_C, feat, anchor_idx, feature_id, P, n_poses_max,
                     o_C_q_G, o_G_p_C, o_feat, o_anchor_idx, o_feature_id, o_P,
                     o_n_poses_max, sigma_landmark, ci_slam_w):
    """MultiSlamUpdate::processOneMatch, multi_slam_update.cpp:61-246.

    Returns dict(inlier, gamma, H, res, S, P_j) (lists entries if inlier)."""
    if anchor_idx < 0:
        raise RuntimeError("anchor_idx < 0")
    n, no = P.shape[1], o_P.shape[1]
    var_l = sigma_landmark ** 2

    def one(Cq, Gp, f, a, fid, npm, ncols, sign):
        al, be, rho = f[3 * fid:3 * fid + 3]
        if sign > 0 and rho == 0:
            raise RuntimeError("rho = 0")
        Ra = quat_to_rot(Cq[a])
        gpf = (1.0 / rho) * Ra @ np.array([al, be, 1.0]) + np.asarray(Gp[a], float)
        J_att = -(1.0 / rho) * Ra @ skew([al, be, 1.0])
        mat = np.eye(3)
        mat[0, 2], mat[1, 2], mat[2, 2] = -al / rho, -be / rho, -1.0 / rho
        Hf = (1.0 / rho) * Ra @ mat
        h = np.zeros((3, ncols))
        col = K_CORE + 3 * a
        h[:, col:col + 3] = sign * np.eye(3)
        h[:, col + 3 * npm:col + 3 * npm + 3] = sign * J_att
        col = K_CORE + (2 * npm + fid) * 3
        h[:, col:col + 3] = sign * Hf
        return gpf, h

    gpf, h = one(C_q_G, G_p_C, feat, anchor_idx, feature_id, n_poses_max, n, +1.0)
    ogpf, oh = one(o_C_q_G, o_G_p_C, o_feat, o_anchor_idx, o_feature_id, o_n_poses_max, no, -1.0)
    res = -gpf + ogpf  # :131
    S0 = h @ P @ h.T + oh @ o_P @ oh.T + var_l * np.eye(3)
    gamma = float(res @ np.linalg.inv(S0) @ res)
    chi = chi2inv(0.9, 3)  # :216-218
    out = dict(inlier=gamma < chi, gamma=gamma, H=h, res=res)
    if out["inlier"]:
        S, w_res = fuse_ci_slam(P, h, o_P, oh, ci_slam_w)
        S = S + var_l * np.eye(3)
        Pj = P.copy()
        for col in (K_CORE + 3 * anchor_idx, K_CORE + 3 * anchor_idx + 3 * n_poses_max,
                    K_CORE + (2 * n_poses_max + feature_id) * 3):
            Pj[col:col + 3, col:col + 3] *= w_res  # :229-239 (diag blocks only, Q7)
        out.update(S=S, P_j=Pj, w_result=w_res)
    return out


def msckf_ci_track(trk, C_q_G, G_p_C, P, n_poses_max, sigma_img, matches, ci_msckf_w):
    """MSCKF-MSCKF CI block of preProcessOneTrack, msckf_update.cpp:96-279.

    matches: list of dict(obs [L_i x 2], q_list, p_list, P, n_poses_max) for
    the agents that observed this same landmark (ground-truth association).
    Returns dict(self=<single-agent result>, ci=None | dict(S,P_j,H,res))."""
    var_img = sigma_img ** 2
    L = len(trk)
    # concatenated lists, matched agents FIRST then self (:113-149)
    q_all, p_all, o_all = [], [], []
    for m in matches:
        Li = len(m["obs"])
        q_all += list(m["q_list"][len(m["q_list"]) - Li:])
        p_all += list(m["p_list"][len(m["p_list"]) - Li:])
        o_all += list(m["obs"])
    q_all += list(C_q_G[len(C_q_G) - L:])
    p_all += list(G_p_C[len(G_p_C) - L:])
    o_all += list(trk)
    ivd, _ = triangulate_gn(q_all, p_all, o_all)
    gpf = global_feature_position(ivd, q_all[-1], p_all[-1])
    own = msckf_process_one_track(trk, P, C_q_G, G_p_C, n_poses_max, var_img, gpf)
    out = dict(self=own, gpf=gpf, ci=None)
    if not (own["valid"] and own["inlier"] and matches):
        return out
    k = len(matches)
    sizes = [P.shape[1]] + [m["P"].shape[1] for m in matches]
    jx = np.zeros((3 * (k + 1), sum(sizes)))
    jf = np.zeros((3 * (k + 1), 3))
    rp = np.zeros(3 * (k + 1))
    jx[0:3, 0:sizes[0]] = own["up_jac"]
    jf[0:3] = own["up_hf"]
    rp[0:3] = own["up_res"]
    col = sizes[0]
    for i, m in enumerate(matches):
        o = msckf_process_one_track(m["obs"], m["P"], m["q_list"], m["p_list"],
                                    m["n_poses_max"], var_img, gpf)
        if not o["valid"]:
            # reference returns early leaving zero rows (:349-357); keep zeros
            col += sizes[i + 1]
            continue
        r0 = 3 * (i + 1)
        jx[r0:r0 + 3, col:col + sizes[i + 1]] = o["up_jac"]
        jf[r0:r0 + 3] = o["up_hf"]
        rp[r0:r0 + 3] = o["up_res"]
        col += sizes[i + 1]
    _, a = left_nullspace(jf)  # nullSpaceProjection, :494-501
    jx = a.T @ jx
    rp = a.T @ rp
    h_j = jx[:, :sizes[0]]
    Hs, col = [], sizes[0]
    S = h_j @ P @ h_j.T
    for i, m in enumerate(matches):
        Hs.append(jx[:, col:col + sizes[i + 1]])
        col += sizes[i + 1]
        S = S + Hs[i] @ m["P"] @ Hs[i].T
    S = S + var_img * np.eye(S.shape[0])
    gamma = float(rp @ np.linalg.inv(S) @ rp)
    chi = chi2inv(0.95, 2 * len(o_all) - 3)  # :245-247
    out["ci_gamma"] = gamma
    if gamma < chi:
        S, w_res = fuse_ci_msckf(P, h_j, [m["P"] for m in matches], Hs, ci_msckf_w)
        S = S + var_img * np.eye(S.shape[0])  # :255 (Q8)
        Pj = P.copy()
        n_poses = len(C_q_G)
        for i in range(L):
            pos = n_poses - L + i
            c = K_CORE + 3 * pos
            Pj[c:c + 3, c:c + 3] *= w_res
            c += 3 * n_poses_max
            Pj[c:c + 3, c:c + 3] *= w_res
        out["ci"] = dict(S=S, P_j=Pj, H=h_j, res=rp, w_result=w_res)
    return out


def multi_uav_update(tracks, matches_per_track, C_q_G, G_p_C, P, n_poses_max, sigma_img, ci_msckf_w, slam=None,
                     msckf_slam_tracks=None):
    """The MULTI_UAV branch of Updater::update (updater.cpp:84-97) around VioUpdater::constructUpdate
    (vio_updater.cpp:295-305, 267-423): every list entry AND the stacked h / res are built from the prior P
    (MsckfUpdate receives cov_s = P once, msckf_update.cpp:27-63); then applyCI runs per entry -- each one
    REPLACING the state covariance by (I - K H) P_j, with P_j a scaled copy of the prior (Q6) -- and finally
    applyUpdate on the covariance the last applyCI left, with h / res still linearised at the prior.

    matches_per_track: dict track index -> list of match dicts (see msckf_ci_track).
    Returns dict(P, corrections [one per applyCI, then the update's], n_ci, ci [entries], visual [visual_update dict])."""
    entries = []
    for i, trk in enumerate(tracks):
        m = matches_per_track.get(i)
        if not m:
            continue
        o = msckf_ci_track(trk, C_q_G, G_p_C, P, n_poses_max, sigma_img, m, ci_msckf_w)
        if o["ci"] is not None:
            entries.append(o["ci"])
    vis = visual_update(tracks, C_q_G, G_p_C, P, n_poses_max, sigma_img, slam=slam, msckf_slam_tracks=msckf_slam_tracks)
    P_cur, corrections = np.asarray(P, float).copy(), []
    for c in entries:                                          # updater.cpp:90-93
        P_cur, corr = apply_ci(c["P_j"], c["H"], c["res"], c["S"])
        corrections.append(corr)
    if vis["h"].size > 0:                                      # :94-97
        P_cur, corr = apply_update(P_cur, vis["h"], vis["res"], vis["r_diag"])
        corrections.append(corr)
    return dict(P=P_cur, corrections=corrections, n_ci=len(entries), ci=entries, visual=vis)


# ----------------------------------------------------------------------------
# StateManager::manage (src/x/vio/state_manager.cpp:31-149): the step that runs immediately before the
# visual update every frame -- persistent-feature removal, anchor re-parametrisation + window slide when
# the window is full, pose augmentation.  Restated AS WRITTEN: dense n x n Jacobians and products.
#   sm    dict(n_poses, n_features, n_poses_max, n_features_max, anchor_idxs[list], filled_before[bool])
#   state dict(p[3], q[4 xyzw], q_ic[4 xyzw], p_ic[3], q_array[4N], p_array[3N], f_array[3M], cov[n,n])
# Both are copied; returns (sm', state').
# ----------------------------------------------------------------------------
def _rot(q_xyzw):
    return quat_to_rot(np.asarray(q_xyzw, float))


def _unit(q):
    q = np.asarray(q, float)
    return q / np.sqrt(np.dot(q, q))


quat_mul_xyzw = quat_mul


def sm_augment_covariance(sm, state, pos, cov):
    """StateManager::augmentCovariance, state_manager.cpp:273-349."""
    N, M = sm["n_poses_max"], sm["n_features_max"]
    n = 15 + 6 * N + 3 * M
    J = np.eye(n) if sm["filled_before"] else np.zeros((n, n))           # :276-284
    k = 15 + (pos + 1) * 3
    J[:k, :k] = np.eye(k)                                                 # :288-291
    a0 = 15 + 3 * N
    J[a0:a0 + 3 * (pos + 1), a0:a0 + 3 * (pos + 1)] = np.eye(3 * (pos + 1))   # :294-297
    f0 = 15 + 6 * N
    nf = sm["n_features"]
    J[f0:f0 + 3 * nf, f0:f0 + 3 * nf] = np.eye(3 * nf)                    # :300-303
    J[15 + 3 * pos:15 + 3 * pos + 3, 0:3] = np.eye(3)                     # :307-308
    J[15 + 3 * pos:15 + 3 * pos + 3, 6:9] = -_rot(state["q"]) @ skew(state["p_ic"])   # :312-315
    qic = np.asarray(state["q_ic"], float)
    qic_conj = np.array([-qic[0], -qic[1], -qic[2], qic[3]])
    J[a0 + 3 * pos:a0 + 3 * pos + 3, 6:9] = _rot(qic_conj)               # :318-322
    Pc = cov.copy()
    for s0 in (15 + 3 * pos, a0 + 3 * pos):                               # :326-339
        Pc[s0:s0 + 3, :] = 0.0
        Pc[:, s0:s0 + 3] = 0.0
    if pos + 1 == N:                                                      # :342-343
        sm["filled_before"] = True
    return J @ Pc @ J.T                                                   # :346-347


def sm_reparametrize_features(sm, atts, poss, feats, cov):
    """StateManager::reparametrizeFeatures, state_manager.cpp:351-482 (modifies feats, sm['anchor_idxs'])."""
    N, M = sm["n_poses_max"], sm["n_features_max"]
    n = 15 + 6 * N + 3 * M
    R_old = _rot(atts[0:4])
    p_old = poss[0:3].copy()
    idx1 = N - 1
    R_new = _rot(atts[4 * idx1:4 * idx1 + 4])
    p_new = poss[3 * idx1:3 * idx1 + 3].copy()
    J = np.eye(n)
    for j in [i for i in range(sm["n_features"]) if sm["anchor_idxs"][i] == 0]:
        al, be, rho = feats[3 * j:3 * j + 3]
        v = np.array([al, be, 1.0])
        new_params = R_new.T @ (-p_new + p_old + (1.0 / rho) * R_old @ v)       # Eq. 38, :402-406
        rho_n = 1.0 / new_params[2]
        al_n, be_n = new_params[0] * rho_n, new_params[1] * rho_n
        feats[3 * j:3 * j + 3] = [al_n, be_n, rho_n]
        sm["anchor_idxs"][j] = idx1
        J_att_old = -(1.0 / rho) * R_new.T @ R_old @ skew(v)                     # :424-427
        J_att_new = skew(new_params)                                            # :430-436
        J_pos_old = R_new.T                                                     # :439-440
        J_pos_new = -R_new.T                                                    # :443-444
        mat = np.eye(3)
        mat[0, 2], mat[1, 2], mat[2, 2] = -al / rho, -be / rho, -1.0 / rho
        J_feat_old = (1.0 / rho) * R_new.T @ R_old @ mat                        # :447-453
        A = np.zeros((3, n))
        A[:, 15 + 3 * idx1:15 + 3 * idx1 + 3] = J_pos_new                       # :458-460
        A[:, 15 + 3 * idx1 + 3 * N:15 + 3 * idx1 + 3 * N + 3] = J_att_new       # :462-463
        A[:, 15:18] = J_pos_old                                                 # :465-466
        A[:, 15 + 3 * N:15 + 3 * N + 3] = J_att_old                             # :468-469
        A[:, 15 + 6 * N + 3 * j:15 + 6 * N + 3 * j + 3] = J_feat_old            # :471-473
        mat = np.eye(3)
        mat[0, 2], mat[1, 2], mat[2, 2] = -al_n, -be_n, -rho_n
        n1 = 15 + 6 * N + 3 * j
        J[n1:n1 + 3, :] = rho_n * mat @ A                                       # :476-482
    return J @ cov @ J.T


def sm_slide_window(sm, atts, poss, cov):
    """StateManager::slideWindow, state_manager.cpp:484-537 (modifies atts, poss, sm)."""
    N, M = sm["n_poses_max"], sm["n_features_max"]
    atts[:4 * (N - 1)] = atts[4:].copy()
    poss[:3 * (N - 1)] = poss[3:].copy()
    atts[4 * (N - 1):] = 0.0
    poss[3 * (N - 1):] = 0.0
    n = cov.shape[0]
    left = np.zeros((n, n))
    left[:15, :15] = np.eye(15)
    if M:
        f0 = 15 + 6 * N
        left[f0:f0 + 3 * M, f0:f0 + 3 * M] = np.eye(3 * M)
    right = left.copy()
    w = 3 * (N - 1)
    left[15:15 + w, 18:18 + w] = np.eye(w)
    left[15 + 3 * N:15 + 3 * N + w, 18 + 3 * N:18 + 3 * N + w] = np.eye(w)
    right[18:18 + w, 15:15 + w] = np.eye(w)
    right[18 + 3 * N:18 + 3 * N + w, 15 + 3 * N:15 + 3 * N + w] = np.eye(w)
    out = left @ cov @ right
    for i in range(sm["n_features"]):
        sm["anchor_idxs"][i] -= 1
    sm["n_poses"] -= 1
    return out


def state_manage(sm, state, del_feat_idx=()):
    """StateManager::manage, state_manager.cpp:31-149."""
    sm = dict(sm, anchor_idxs=list(sm["anchor_idxs"]))
    st = {k: (np.array(v, dtype=float, copy=True) if not np.isscalar(v) else v) for k, v in state.items()}
    N, M = sm["n_poses_max"], sm["n_features_max"]
    att, pos, feats, cov = st["q_array"], st["p_array"], st["f_array"], st["cov"]
    cae = quat_mul_xyzw(_unit(st["q"]), _unit(st["q_ic"]))               # State::computeCameraOrientation, state.cpp:184-187
    cpe = st["p"] + _rot(st["q"]) @ st["p_ic"]                           # computeCameraPosition, :189-191
    n = cov.shape[0]
    for idx in sorted(del_feat_idx, reverse=True):                        # :52-112
        n1 = sm["n_features"] - idx - 1
        feats[3 * idx:3 * idx + 3 * n1] = feats[3 * (idx + 1):3 * (idx + 1) + 3 * n1].copy()
        feats[3 * (sm["n_features"] - 1):3 * sm["n_features"]] = 0.0
        del sm["anchor_idxs"][idx]
        sm["anchor_idxs"].append(-1)
        idx0 = 15 + 6 * N + 3 * idx
        idx1 = idx0 + 3
        dim0 = 3 * (M - idx - 1)
        cols_after = cov[:, idx1:idx1 + dim0].copy()
        cols_after[idx0:idx0 + dim0, :] = cols_after[idx1:idx1 + dim0, :].copy()
        rows_after = cov[idx1:idx1 + dim0, :idx0].copy()
        cov[:, idx0:idx0 + dim0] = cols_after
        cov[idx0:idx0 + dim0, :idx0] = rows_after
        cov[:, n - 3:] = 0.0
        cov[n - 3:, :] = 0.0
        sm["n_features"] -= 1
    if sm["n_poses"] == N:                                                # :119-125
        cov = sm_reparametrize_features(sm, att, pos, feats, cov)
        cov = sm_slide_window(sm, att, pos, cov)
    p_i = sm["n_poses"]
    att[4 * p_i:4 * p_i + 4] = cae                                        # :133
    pos[3 * p_i:3 * p_i + 3] = cpe                                        # :134
    cov = sm_augment_covariance(sm, st, p_i, cov)                        # :137
    sm["n_poses"] += 1
    st.update(q_array=att, p_array=pos, f_array=feats, cov=cov)
    return sm, st


def propagate_covariance_matrices(cov_0, f_d, q_d):
    """Propagator::propagateCovarianceMatrices, propagator.cpp:166-205 (block form, vi computed on its own)."""
    cov_0 = np.asarray(cov_0, float)
    k = 15
    cov_1 = np.empty_like(cov_0)
    cov_1[:k, :k] = f_d @ cov_0[:k, :k] @ f_d.T + q_d      # :194
    cov_1[:k, k:] = f_d @ cov_0[:k, k:]                    # :195
    cov_1[k:, :k] = cov_0[k:, :k] @ f_d.T                  # :203
    cov_1[k:, k:] = cov_0[k:, k:]                          # :204
    return cov_1


# ----------------------------------------------------------------------------
# IMU propagation closed forms (host side): src/x/ekf/propagator.cpp
# ----------------------------------------------------------------------------
def omega_matrix(w):
    """Vector3::toOmegaMatrix (eigen_matrix_base_plugin.h:55-63): quaternion kinematic matrix, (x,y,z,w) order."""
    x, y, z = w
    return np.array([[0.0, z, -y, x], [-z, 0.0, x, y], [y, -x, 0.0, z], [-x, -y, -z, 0.0]])


def quaternion_integrator(e_w_0, e_w_1, dt):
    """Propagator::quaternionIntegrator, propagator.cpp:73-97: 4th-order Taylor series of exp(Omega(mean) dt / 2) plus
    the first-order commutator term (Trawny & Roumeliotis eq. 130-131)."""
    om1, om0 = omega_matrix(e_w_1), omega_matrix(e_w_0)
    a = omega_matrix((np.asarray(e_w_1) + np.asarray(e_w_0)) / 2.0) * 0.5 * dt
    a_k, mat_exp, fac = a.copy(), np.eye(4), 1
    for k in range(1, 5):
        fac *= k
        mat_exp = mat_exp + a_k / fac
        a_k = a_k @ a
    return mat_exp + 1.0 / 48.0 * (om1 @ om0 - om0 @ om1) * dt * dt


def propagate_state(s0, s1, g):
    """Propagator::propagateState, propagator.cpp:30-51.  s0 / s1: dicts with time, p, v, q (xyzw), b_w, b_a, w_m, a_m;
    s1 arrives with time / w_m / a_m set (State::setImu) and leaves with p, v, q, biases filled."""
    s1["b_w"], s1["b_a"] = s0["b_w"].copy(), s0["b_a"].copy()                      # setStaticStatesFrom (window arrays too)
    e_w_1, e_a_1 = s1["w_m"] - s1["b_w"], s1["a_m"] - s1["b_a"]
    e_w_0, e_a_0 = s0["w_m"] - s0["b_w"], s0["a_m"] - s0["b_a"]
    dt = s1["time"] - s0["time"]
    q1 = quaternion_integrator(e_w_0, e_w_1, dt) @ s0["q"]
    s1["q"] = q1 / np.linalg.norm(q1)
    dv = (quat_to_rot(s1["q"]) @ e_a_1 + quat_to_rot(s0["q"]) @ e_a_0) / 2.0
    s1["v"] = s0["v"] + (dv + g) * dt
    s1["p"] = s0["p"] + (s1["v"] + s0["v"]) / 2.0 * dt
    return s1


def discrete_state_transition(dt, e_w, e_a, q_xyzw):
    """Propagator::discreteStateTransition, propagator.cpp:99-164 (15 x 15, error-state order p v theta b_w b_a)."""
    w_x, a_x, eye3 = skew(e_w), skew(e_a), np.eye(3)
    c_q = quat_to_rot(q_xyzw)
    dt_2_f2 = dt * dt * 0.5
    dt_3_f3 = dt_2_f2 * dt / 3.0
    dt_4_f4 = dt_3_f3 * dt * 0.25
    dt_5_f5 = dt_4_f4 * dt * 0.2
    c_q_a_x = c_q @ a_x
    a = c_q_a_x @ (-dt_2_f2 * eye3 + dt_3_f3 * w_x - dt_4_f4 * w_x @ w_x)
    b = c_q_a_x @ (dt_3_f3 * eye3 - dt_4_f4 * w_x + dt_5_f5 * w_x @ w_x)
    d = -a
    e = eye3 - dt * w_x + dt_2_f2 * w_x @ w_x
    f = -dt * eye3 + dt_2_f2 * w_x - dt_3_f3 * (w_x @ w_x)
    c = c_q_a_x @ f
    F = np.eye(15)
    F[0:3, 3:6] = dt * eye3
    F[0:3, 6:9] = a
    F[0:3, 9:12] = b
    F[0:3, 12:15] = -c_q * dt_2_f2
    F[3:6, 6:9] = c
    F[3:6, 9:12] = d
    F[3:6, 12:15] = -c_q * dt
    F[6:9, 6:9] = e
    F[6:9, 9:12] = f
    return F


def process_noise_model(dt, e_w, e_a, q_xyzw, n_w, n_bw, n_a, n_ba):
    """A first-principles discrete process noise  Q_d = int_0^dt F_d(t) G Q_c G^T F_d(t)^T dt  (Weiss 2012, eq. 2.33)
    with the reference's own F_d(t) and G = [v: -C(q) n_a, theta: -n_w, b_w: n_bw, b_a: n_ba], integrated exactly
    (the integrand is a polynomial in t: 12-point Gauss-Legendre).

    NOT the reference's q_d: Propagator::discreteProcessNoiseCov (propagator.cpp:207-840) is 630 lines of
    machine-generated scalar code that assigns 147 of the 225 entries, is not symmetric, and carries first-order gyro-noise
    terms in the velocity block that no derivation of this form produces (tests/test_oracle_propagator.py measures the
    difference on the reference's own outputs, tests/golden/propagator_qd.npz).  It cannot be restated without copying
    it, and it is host-side 15 x 15 arithmetic, so a drop-in keeps the reference's function and hands its q_d to
    xk_cov_propagate; this model is what the mirror's own examples and the frame-loop benchmark use."""
    G = np.zeros((15, 12))
    G[3:6, 0:3] = -quat_to_rot(q_xyzw)
    G[6:9, 6:9] = -np.eye(3)
    G[9:12, 9:12] = np.eye(3)
    G[12:15, 3:6] = np.eye(3)
    Qc = np.diag([n_a ** 2] * 3 + [n_ba ** 2] * 3 + [n_w ** 2] * 3 + [n_bw ** 2] * 3)
    x, wq = np.polynomial.legendre.leggauss(12)
    Q = np.zeros((15, 15))
    for xi, wi in zip(x, wq):
        t = 0.5 * dt * (xi + 1.0)
        Ft = discrete_state_transition(t, e_w, e_a, q_xyzw)
        Q += 0.5 * dt * wi * (Ft @ G @ Qc @ G.T @ Ft.T)
    return 0.5 * (Q + Q.T)


# ----------------------------------------------------------------------------
# MSCKF-SLAM update + persistent-feature initialisation
#   MsckfSlamUpdate            src/x/vio/msckf_slam_update.cpp:25-267
#   initMsckfSlamFeatures      src/x/vio/state_manager.cpp:151-174
#   initStandardSlamFeatures   src/x/vio/state_manager.cpp:176-197
#   addFeatureStates           src/x/vio/state_manager.cpp:199-226
#   computeInverseDepthsNew    src/x/vio/slam_update.cpp:216-242
# ----------------------------------------------------------------------------
def msckf_slam_process_one_track(obs, C_q_G, G_p_C, P, n_poses_max, var_img):
    """MsckfSlamUpdate::processOneTrack, msckf_slam_update.cpp:64-267: the feature is parametrised by inverse
    depth in the LAST pose of the window (its future anchor), so every earlier observation also has Jacobian
    blocks on that pose; no observability constraint is applied.  Returns dict(inlier, gamma, jac0, res0, H1,
    H2, r1, feature, gn_iters)."""
    L = len(obs)
    n = P.shape[1]
    npz = len(C_q_G)
    q_l, p_l = C_q_G[npz - L:], G_p_C[npz - L:]
    ivd, it = triangulate_gn(q_l, p_l, obs)                                   # :82
    al, be, rho = ivd
    R_n = quat_to_rot(C_q_G[-1])                                              # anchor = last pose, :88-94
    p_n = np.asarray(G_p_C[-1], float)
    gpf = (1.0 / rho) * R_n @ np.array([al, be, 1.0]) + p_n                   # :97-99
    h = np.zeros((2 * L, n))
    hf = np.zeros((2 * L, 3))
    res = np.zeros(2 * L)
    N3 = 3 * n_poses_max
    for i in range(L):
        pos = npz - L + i                                                     # :105
        R_i = quat_to_rot(C_q_G[pos])
        p_i = np.asarray(G_p_C[pos], float)
        c = R_i.T @ (gpf - p_i)                                               # :111-113
        res[2 * i:2 * i + 2] = np.asarray(obs[i], float) - np.array([c[0] / c[2], c[1] / c[2]])   # :116-127
        if i == L - 1:                                                        # :133-142
            hf[2 * i:2 * i + 2] = np.array([[1.0, 0, 0], [0, 1.0, 0]])
            continue
        Ji = np.array([[1.0 / c[2], 0.0, -c[0] / c[2] ** 2], [0.0, 1.0 / c[2], -c[1] / c[2] ** 2]])   # :145-153
        J_att = Ji @ skew(c)                                                  # :156-160
        J_pos = -Ji @ R_i.T                                                   # :163-164
        J_anchor_att = -(1.0 / rho) * Ji @ R_i.T @ R_n @ skew([al, be, 1.0])  # :167-170
        J_anchor_pos = -J_pos                                                 # :173
        mat = np.eye(3)
        mat[0, 2], mat[1, 2], mat[2, 2] = -al / rho, -be / rho, -1.0 / rho
        hf[2 * i:2 * i + 2] = (1.0 / rho) * Ji @ R_i.T @ R_n @ mat            # :176-183
        h[2 * i:2 * i + 2, 15 + 3 * pos:15 + 3 * pos + 3] = J_pos             # :189-190
        h[2 * i:2 * i + 2, 15 + 3 * pos + N3:15 + 3 * pos + N3 + 3] = J_att   # :192-193
        a = npz - 1
        h[2 * i:2 * i + 2, 15 + 3 * a:15 + 3 * a + 3] = J_anchor_pos          # :195-196
        h[2 * i:2 * i + 2, 15 + 3 * a + N3:15 + 3 * a + N3 + 3] = J_anchor_att   # :198-199
    a_up, a_null = left_nullspace(hf)                                         # :206-207
    res0, h0 = a_null.T @ res, a_null.T @ h                                   # :210-211
    H1, H2, r1 = a_up.T @ h, a_up.T @ hf, a_up.T @ res                        # :224-233
    d = 2 * L - 3
    S = h0 @ P @ h0.T + var_img * np.eye(d)                                   # :243
    gamma = float(res0 @ np.linalg.inv(S) @ res0)
    return dict(inlier=gamma < chi2inv(0.95, d), gamma=gamma, jac0=h0, res0=res0, H1=H1, H2=H2, r1=r1,
                feature=np.asarray(ivd, float), gn_iters=it)


def msckf_slam_update(tracks, C_q_G, G_p_C, P, n_poses_max, sigma_img):
    """MsckfSlamUpdate ctor, msckf_slam_update.cpp:25-62.  Returns (jac, res, cov_diag, info, init_mats);
    init_mats (H1, H2 block diagonal, r1, features) is filled for EVERY track, gated out or not."""
    k = len(tracks)
    n = P.shape[1]
    rows0 = 2 * sum(len(t) for t in tracks) - 3 * k
    jac, cov_diag, res = np.zeros((rows0, n)), np.ones(rows0), np.zeros(rows0)
    H1, H2, r1, feats = np.zeros((3 * k, n)), np.zeros((3 * k, 3 * k)), np.zeros(3 * k), np.zeros(3 * k)
    var_img = sigma_img * sigma_img
    inlier, gamma = np.zeros(k, np.int32), np.zeros(k)
    row_h = 0
    for j, trk in enumerate(tracks):
        o = msckf_slam_process_one_track(trk, C_q_G, G_p_C, P, n_poses_max, var_img)
        H1[3 * j:3 * j + 3], H2[3 * j:3 * j + 3, 3 * j:3 * j + 3], r1[3 * j:3 * j + 3] = o["H1"], o["H2"], o["r1"]
        feats[3 * j:3 * j + 3] = o["feature"]
        gamma[j] = o["gamma"]
        if o["inlier"]:
            d = 2 * len(trk) - 3
            jac[row_h:row_h + d], res[row_h:row_h + d], cov_diag[row_h:row_h + d] = o["jac0"], o["res0"], var_img
            row_h += d
            inlier[j] = 1
    return jac, res, cov_diag, dict(inlier=inlier, gamma=gamma, rows_used=row_h), dict(H1=H1, H2=H2, r1=r1, features=feats)


def sm_add_feature_states(sm, state, new_features, cov_new, cross):
    """StateManager::addFeatureStates, state_manager.cpp:199-226 (modifies sm and state)."""
    N = sm["n_poses_max"]
    k3 = len(new_features)
    nf = sm["n_features"]
    state["f_array"][3 * nf:3 * nf + k3] = new_features
    P = state["cov"]
    ns = 15 + 6 * N + 3 * nf
    P[ns:ns + k3, :] = cross
    P[:, ns:ns + k3] = cross.T
    P[ns:ns + k3, ns:ns + k3] = cov_new
    for i in range(k3 // 3):
        sm["anchor_idxs"][nf + i] = sm["n_poses"] - 1
    sm["n_features"] = nf + k3 // 3


def sm_init_msckf_slam_features(sm, state, init_mats, correction, sigma_img):
    """StateManager::initMsckfSlamFeatures, state_manager.cpp:151-174."""
    sm = dict(sm, anchor_idxs=list(sm["anchor_idxs"]))
    st = {k: (np.array(v, dtype=float, copy=True) if not np.isscalar(v) else v) for k, v in state.items()}
    P = st["cov"]
    H2_inv = np.linalg.inv(init_mats["H2"])
    G = H2_inv @ init_mats["H1"]
    new_features = init_mats["features"] - G @ correction + H2_inv @ init_mats["r1"]
    var_img = sigma_img * sigma_img
    P_cross = -G @ P
    P_diag = G @ P @ G.T + var_img * H2_inv @ H2_inv.T
    sm_add_feature_states(sm, st, new_features, P_diag, P_cross)
    return sm, st


def sm_init_standard_slam_features(sm, state, last_obs, rho_0, sigma_img, sigma_rho_0):
    """computeInverseDepthsNew (slam_update.cpp:216-242) + initStandardSlamFeatures (state_manager.cpp:176-197)."""
    sm = dict(sm, anchor_idxs=list(sm["anchor_idxs"]))
    st = {k: (np.array(v, dtype=float, copy=True) if not np.isscalar(v) else v) for k, v in state.items()}
    k = len(last_obs)
    ivds = np.zeros(3 * k)
    for j, z in enumerate(last_obs):
        ivds[3 * j:3 * j + 3] = [z[0], z[1], rho_0]
    n = st["cov"].shape[0]
    P_diag = sigma_img ** 2 * np.eye(3 * k)
    for j in range(k):
        P_diag[3 * j + 2, 3 * j + 2] = sigma_rho_0 ** 2
    sm_add_feature_states(sm, st, ivds, P_diag, np.zeros((3 * k, n)))
    return sm, st
