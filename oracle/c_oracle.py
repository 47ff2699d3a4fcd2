"""ctypes binding of oracle/libxk_oracle.so (the C restatement).

TEST INFRASTRUCTURE ONLY -- see the header of oracle/xk_oracle.c.  Only
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

c_dp = C.POINTER(C.c_double)
c_ip = C.POINTER(C.c_int)


def build(march="x86-64-v3", out=None, force=False, extra_flags=()):
    out = out or os.path.join(_HERE, "libxk_oracle.so")
    src = os.path.join(_HERE, "xk_oracle.c")
    if force or not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O3", f"-march={march}", "-std=gnu11", "-fPIC", "-Wall",
                               "-Wno-unused-function", *extra_flags, "-shared", "-o", out, src, "-lm"])
    return out


def lib(path=None):
    global _LIB
    if path is not None:
        return C.CDLL(path)
    if _LIB is None:
        p = os.path.join(_HERE, "libxk_oracle.so")
        if not os.path.exists(p):
            build()
        _LIB = C.CDLL(p)
        _LIB.xo_gemm_probe.restype = C.c_double
    return _LIB


def _d(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(c_dp)


def _i(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(c_ip)


def _f(a):
    """column-major copy"""
    a = np.asfortranarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(c_dp)


def _chk(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what} failed with status {rc}")


def chi2inv(p, dof):
    out = C.c_double()
    _chk(lib().xo_chi2inv(C.c_double(p), C.c_int(dof), C.byref(out)), "xo_chi2inv")
    return out.value


def triangulate_gn(q, p, obs, max_iter=10, term=1e-5):
    q, qp = _d(q)
    p, pp = _d(p)
    obs, op = _d(obs)
    ivd = np.zeros(3)
    it = C.c_int()
    _chk(lib().xo_triangulate_gn(qp, pp, op, C.c_int(len(obs)), C.c_int(max_iter), C.c_double(term),
                                 ivd.ctypes.data_as(c_dp), C.byref(it)), "xo_triangulate_gn")
    return ivd, it.value


def msckf_update(sc_or_tracks, C_q_G=None, G_p_C=None, P=None, n_poses_max=None, sigma_img=None):
    """Either msckf_update(scenario_dict) or explicit arrays with (trk_off, obs_xy) tuple."""
    if isinstance(sc_or_tracks, dict):
        sc = sc_or_tracks
        trk_off, obs_xy = sc["trk_off"], sc["obs_xy"]
        C_q_G, G_p_C, P = sc["C_q_G"], sc["G_p_C"], sc["P"]
        n_poses_max, sigma_img = sc["n_poses_max"], sc["sigma_img"]
    else:
        trk_off, obs_xy = sc_or_tracks
    K = len(trk_off) - 1
    n = P.shape[0]
    rows = 2 * int(trk_off[-1] - trk_off[0]) - 3 * K
    q, qp = _d(C_q_G)
    p, pp = _d(G_p_C)
    to, top = _i(trk_off)
    ob, obp = _d(obs_xy)
    Pf, Pp = _f(P)
    jac = np.zeros((rows, n), order="F")
    res = np.zeros(rows)
    cov = np.zeros(rows)
    inl = np.zeros(K, dtype=np.int32)
    gam = np.zeros(K)
    feats = np.zeros((K, 3))
    its = np.zeros(K, dtype=np.int32)
    used = C.c_int()
    _chk(lib().xo_msckf_update(qp, pp, C.c_int(len(G_p_C)), top, obp, C.c_int(K), Pp, C.c_int(n),
                               C.c_int(n_poses_max), C.c_double(sigma_img), jac.ctypes.data_as(c_dp),
                               res.ctypes.data_as(c_dp), cov.ctypes.data_as(c_dp), C.c_int(rows),
                               inl.ctypes.data_as(c_ip), gam.ctypes.data_as(c_dp),
                               feats.ctypes.data_as(c_dp), its.ctypes.data_as(c_ip), C.byref(used)),
         "xo_msckf_update")
    return jac, res, cov, dict(inlier=inl, gamma=gam, feats=feats, gn_iters=its, rows_used=used.value)


def slam_update(C_q_G, G_p_C, feat, anchor_idxs, track_sizes, z_last, P, n_poses_max, sigma_img):
    M = len(anchor_idxs)
    n = P.shape[0]
    q, qp = _d(C_q_G)
    p, pp = _d(G_p_C)
    f, fp = _d(feat)
    a, ap = _i(anchor_idxs)
    ts, tsp = _i(track_sizes)
    z, zp = _d(z_last)
    Pf, Pp = _f(P)
    jac = np.zeros((2 * M, n), order="F")
    res = np.zeros(2 * M)
    cov = np.zeros(2 * M)
    inl = np.zeros(M, dtype=np.int32)
    gam = np.zeros(M)
    used = C.c_int()
    _chk(lib().xo_slam_update(qp, pp, C.c_int(len(G_p_C)), fp, ap, tsp, zp, C.c_int(M), Pp, C.c_int(n),
                              C.c_int(n_poses_max), C.c_double(sigma_img), jac.ctypes.data_as(c_dp),
                              res.ctypes.data_as(c_dp), cov.ctypes.data_as(c_dp),
                              inl.ctypes.data_as(c_ip), gam.ctypes.data_as(c_dp), C.byref(used)),
         "xo_slam_update")
    return jac, res, cov, dict(inlier=inl, gamma=gam, rows_used=used.value)


def qr_compress(h, res, sigma_img):
    rows, cols = h.shape
    hf, hp = _f(h)
    r, rp = _d(res)
    ho = np.zeros((cols, cols), order="F")
    ro = np.zeros(cols)
    co = np.zeros(cols)
    did = C.c_int()
    _chk(lib().xo_qr_compress(hp, C.c_int(rows), C.c_int(cols), rp, C.c_double(sigma_img),
                              ho.ctypes.data_as(c_dp), ro.ctypes.data_as(c_dp),
                              co.ctypes.data_as(c_dp), C.byref(did)), "xo_qr_compress")
    if did.value:
        return ho, ro, co, True
    return h, res, None, False


def apply_update(P, H, res, r_diag, correction_total=None, cov_update=True):
    n = P.shape[0]
    m = H.shape[0]
    Pf = np.array(P, dtype=np.float64, order="F", copy=True)
    Hf, Hp = _f(H)
    r, rp = _d(res)
    rd, rdp = _d(r_diag)
    ct = np.zeros(n) if correction_total is None else np.array(correction_total, dtype=np.float64)
    corr = np.zeros(n)
    _chk(lib().xo_apply_update(Pf.ctypes.data_as(c_dp), C.c_int(n), Hp, C.c_int(m), rp, rdp,
                               ct.ctypes.data_as(c_dp), C.c_int(int(cov_update)),
                               corr.ctypes.data_as(c_dp)), "xo_apply_update")
    return np.ascontiguousarray(Pf), corr


def apply_ci(ci_P, H, res, S):
    n = ci_P.shape[0]
    m = H.shape[0]
    Pf, Pp = _f(ci_P)
    Hf, Hp = _f(H)
    r, rp = _d(res)
    Sf, Sp = _f(S)
    Po = np.zeros((n, n), order="F")
    corr = np.zeros(n)
    _chk(lib().xo_apply_ci(Po.ctypes.data_as(c_dp), Pp, C.c_int(n), Hp, C.c_int(m), rp, Sp,
                           corr.ctypes.data_as(c_dp)), "xo_apply_ci")
    return np.ascontiguousarray(Po), corr


def state_correct(state, corr):
    s = {k: np.array(v, dtype=np.float64, copy=True) for k, v in state.items()}
    N = s["p_array"].size // 3
    M = s["f_array"].size // 3
    c, cp = _d(corr)
    if M == 0:
        s["f_array"] = np.zeros(0)
    fa = s["f_array"] if M else np.zeros(1)
    _chk(lib().xo_state_correct(*(s[k].ctypes.data_as(c_dp) for k in ("p", "v", "q", "b_w", "b_a",
                                                                        "p_array", "q_array")),
                                fa.ctypes.data_as(c_dp), C.c_int(N), C.c_int(M), cp), "xo_state_correct")
    return s


def visual_update(sc, library=None):
    """Full as-written update on a scenario dict.  Returns dict(P, correction, inlier, gamma, ...)."""
    L = library or lib()
    K = len(sc["trk_off"]) - 1
    n = sc["P"].shape[0]
    M = len(sc["slam_anchor_idxs"]) if "slam_anchor_idxs" in sc else 0
    q, qp = _d(sc["C_q_G"])
    p, pp = _d(sc["G_p_C"])
    to, top = _i(sc["trk_off"])
    ob, obp = _d(sc["obs_xy"])
    Pf = np.array(sc["P"], dtype=np.float64, order="F", copy=True)
    corr = np.zeros(n)
    inl = np.zeros(max(K, 1), dtype=np.int32)
    gam = np.zeros(max(K, 1))
    inls = np.zeros(max(M, 1), dtype=np.int32)
    gams = np.zeros(max(M, 1))
    did = C.c_int()
    if M:
        f, fp = _d(sc["slam_feat"])
        a, ap = _i(sc["slam_anchor_idxs"])
        ts, tsp = _i(sc["slam_track_sizes"])
        z, zp = _d(sc["slam_z_last"])
    else:
        fp = zp = c_dp()
        ap = tsp = c_ip()
    _chk(L.xo_visual_update(qp, pp, C.c_int(len(sc["G_p_C"])), top, obp, C.c_int(K), fp, ap, tsp, zp,
                            C.c_int(M), Pf.ctypes.data_as(c_dp), C.c_int(n), C.c_int(sc["n_poses_max"]),
                            C.c_double(sc["sigma_img"]), corr.ctypes.data_as(c_dp),
                            inl.ctypes.data_as(c_ip), gam.ctypes.data_as(c_dp),
                            inls.ctypes.data_as(c_ip), gams.ctypes.data_as(c_dp), C.byref(did)),
         "xo_visual_update")
    return dict(P=np.ascontiguousarray(Pf), correction=corr, inlier=inl[:K], gamma=gam[:K],
                inlier_slam=inls[:M], gamma_slam=gams[:M], did_qr=bool(did.value))


def visual_update_iekf(sc, state, iekf_iter, library=None):
    """The IEKF loop of Updater::update (updater.cpp:99-110) on a scenario + full state dict (p, v, q, b_w, b_a,
    p_array 3N, q_array 4N xyzw, f_array 3M_cap).  Returns dict(P, correction (= correction_total), state, inlier,
    gamma, inlier_slam, gamma_slam); flags are the last iteration's."""
    L = library or lib()
    K = len(sc["trk_off"]) - 1
    n = sc["P"].shape[0]
    M = len(sc["slam_anchor_idxs"]) if "slam_anchor_idxs" in sc else 0
    N = sc["n_poses_max"]
    s = {k: np.array(v, dtype=np.float64, copy=True) for k, v in state.items()}
    Mcap = (n - 15 - 6 * N) // 3
    assert s["p_array"].size == 3 * N and s["q_array"].size == 4 * N and s["f_array"].size == 3 * Mcap
    fa = s["f_array"] if Mcap else np.zeros(1)
    to, top = _i(sc["trk_off"])
    ob, obp = _d(sc["obs_xy"])
    Pf = np.array(sc["P"], dtype=np.float64, order="F", copy=True)
    ctot = np.zeros(n)
    inl = np.zeros(max(K, 1), dtype=np.int32)
    gam = np.zeros(max(K, 1))
    inls = np.zeros(max(M, 1), dtype=np.int32)
    gams = np.zeros(max(M, 1))
    if M:
        a, ap = _i(sc["slam_anchor_idxs"])
        ts, tsp = _i(sc["slam_track_sizes"])
        z, zp = _d(sc["slam_z_last"])
    else:
        zp = c_dp()
        ap = tsp = c_ip()
    _chk(L.xo_visual_update_iekf(*(s[k].ctypes.data_as(c_dp) for k in ("p", "v", "q", "b_w", "b_a", "p_array", "q_array")),
                                 fa.ctypes.data_as(c_dp), C.c_int(len(sc["G_p_C"])), top, obp, C.c_int(K), ap, tsp, zp,
                                 C.c_int(M), Pf.ctypes.data_as(c_dp), C.c_int(n), C.c_int(N), C.c_double(sc["sigma_img"]),
                                 C.c_int(iekf_iter), ctot.ctypes.data_as(c_dp), inl.ctypes.data_as(c_ip),
                                 gam.ctypes.data_as(c_dp), inls.ctypes.data_as(c_ip), gams.ctypes.data_as(c_dp)),
         "xo_visual_update_iekf")
    return dict(P=np.ascontiguousarray(Pf), correction=ctot, state=s, inlier=inl[:K], gamma=gam[:K],
                inlier_slam=inls[:M], gamma_slam=gams[:M])


def fuse_ci_slam(Pa, Ha, Pb, Hb, w):
    m = Ha.shape[0]
    Paf, Pap = _f(Pa)
    Haf, Hap = _f(Ha)
    Pbf, Pbp = _f(Pb)
    Hbf, Hbp = _f(Hb)
    S = np.zeros((m, m), order="F")
    wr = C.c_double()
    rc = lib().xo_fuse_ci_slam(Pap, C.c_int(Pa.shape[0]), Hap, Pbp, C.c_int(Pb.shape[0]), Hbp, C.c_int(m),
                               C.c_double(w), S.ctypes.data_as(c_dp), C.byref(wr))
    if rc != 0:
        raise RuntimeError("The CI weights must be lower than 1.0 and larger than 0.0")
    return np.ascontiguousarray(S), wr.value


def fuse_ci_msckf(P, H, Ps, Hs, w):
    m = H.shape[0]
    k = len(Ps)
    Pf, Pp = _f(P)
    Hf, Hp = _f(H)
    keep = [_f(x) for x in Ps] + [_f(x) for x in Hs]
    PsA = (c_dp * max(k, 1))(*[keep[i][1] for i in range(k)])
    HsA = (c_dp * max(k, 1))(*[keep[k + i][1] for i in range(k)])
    ns, nsp = _i([x.shape[0] for x in Ps] or [0])
    S = np.zeros((m, m), order="F")
    wr = C.c_double()
    rc = lib().xo_fuse_ci_msckf(Pp, C.c_int(P.shape[0]), Hp, C.c_int(m), C.c_int(k), PsA, nsp, HsA,
                                C.c_double(w), S.ctypes.data_as(c_dp), C.byref(wr))
    if rc != 0:
        raise RuntimeError("The CI weights must be lower than 1.0 and larger 0.0")
    return np.ascontiguousarray(S), wr.value


def multi_slam_match(C_q_G, G_p_C, feat, anchor_idx, feature_id, P, n_poses_max, o_C_q_G, o_G_p_C,
                     o_feat, o_anchor_idx, o_feature_id, o_P, o_n_poses_max, sigma_landmark, ci_slam_w):
    n, no = P.shape[0], o_P.shape[0]
    a = [_d(C_q_G), _d(G_p_C), _d(feat), _f(P), _d(o_C_q_G), _d(o_G_p_C), _d(o_feat), _f(o_P)]
    inl = C.c_int()
    gam = C.c_double()
    H = np.zeros((3, n), order="F")
    res = np.zeros(3)
    S = np.zeros((3, 3), order="F")
    Pj = np.zeros((n, n), order="F")
    rc = lib().xo_multi_slam_match(a[0][1], a[1][1], a[2][1], C.c_int(anchor_idx), C.c_int(feature_id),
                                   a[3][1], C.c_int(n), C.c_int(n_poses_max), a[4][1], a[5][1], a[6][1],
                                   C.c_int(o_anchor_idx), C.c_int(o_feature_id), a[7][1], C.c_int(no),
                                   C.c_int(o_n_poses_max), C.c_double(sigma_landmark), C.c_double(ci_slam_w),
                                   C.byref(inl), C.byref(gam), H.ctypes.data_as(c_dp),
                                   res.ctypes.data_as(c_dp), S.ctypes.data_as(c_dp), Pj.ctypes.data_as(c_dp))
    if rc != 0:
        raise RuntimeError(f"xo_multi_slam_match status {rc}")
    out = dict(inlier=bool(inl.value), gamma=gam.value, H=np.ascontiguousarray(H), res=res)
    if out["inlier"]:
        out.update(S=np.ascontiguousarray(S), P_j=np.ascontiguousarray(Pj))
    return out


def msckf_ci_track(trk, C_q_G, G_p_C, P, n_poses_max, sigma_img, matches, ci_msckf_w):
    k = len(matches)
    n = P.shape[0]
    base = [_d(trk), _d(C_q_G), _d(G_p_C), _f(P)]
    mo = [_d(m["obs"]) for m in matches]
    mq = [_d(m["q_list"]) for m in matches]
    mp = [_d(m["p_list"]) for m in matches]
    mP = [_f(m["P"]) for m in matches]
    arr = lambda xs: (c_dp * max(k, 1))(*[x[1] for x in xs])
    mL, mLp = _i([len(m["obs"]) for m in matches] or [0])
    mnp, mnpp = _i([len(m["p_list"]) for m in matches] or [0])
    mn, mnp_ = _i([m["P"].shape[0] for m in matches] or [0])
    si, sg, hc, cg = C.c_int(), C.c_double(), C.c_int(), C.c_double()
    m3 = max(3 * k, 1)
    H = np.zeros((m3, n), order="F")
    res = np.zeros(m3)
    S = np.zeros((m3, m3), order="F")
    Pj = np.zeros((n, n), order="F")
    gpf = np.zeros(3)
    rc = lib().xo_msckf_ci_track(base[0][1], C.c_int(len(trk)), base[1][1], base[2][1], C.c_int(len(G_p_C)),
                                 base[3][1], C.c_int(n), C.c_int(n_poses_max), C.c_double(sigma_img), C.c_int(k),
                                 arr(mo), mLp, arr(mq), arr(mp), mnpp, arr(mP), mnp_, C.c_double(ci_msckf_w),
                                 C.byref(si), C.byref(sg), C.byref(hc), C.byref(cg), H.ctypes.data_as(c_dp),
                                 res.ctypes.data_as(c_dp), S.ctypes.data_as(c_dp), Pj.ctypes.data_as(c_dp),
                                 gpf.ctypes.data_as(c_dp))
    if rc != 0:
        raise RuntimeError(f"xo_msckf_ci_track status {rc}")
    out = dict(self_inlier=bool(si.value), self_gamma=sg.value, gpf=gpf, ci=None, ci_gamma=cg.value)
    if hc.value:
        out["ci"] = dict(S=np.ascontiguousarray(S), P_j=np.ascontiguousarray(Pj), H=np.ascontiguousarray(H), res=res)
    return out


def track_jacobians(sc, k, gpf):
    """jac (2L x n), hf (2L x 3), res (2L) of track k BEFORE the null-space projection (msckf_update.cpp:328-417), nan flag."""
    off = sc["trk_off"]
    L = int(off[k + 1] - off[k])
    n = sc["P"].shape[0]
    ob, obp = _d(sc["obs_xy"][off[k]:off[k + 1]])
    q, qp = _d(sc["C_q_G"])
    p, pp = _d(sc["G_p_C"])
    g, gp = _d(gpf)
    jac = np.zeros((2 * L, n), order="F")
    hf = np.zeros((2 * L, 3), order="F")
    res = np.zeros(2 * L)
    bad = C.c_int()
    _chk(lib().xo_track_jacobians(obp, C.c_int(L), C.c_int(n), qp, pp, C.c_int(len(sc["G_p_C"])), C.c_int(sc["n_poses_max"]), gp,
                                  jac.ctypes.data_as(c_dp), hf.ctypes.data_as(c_dp), res.ctypes.data_as(c_dp), C.byref(bad)),
         "xo_track_jacobians")
    return jac, hf, res, bool(bad.value)


def eigen_variant(sc, reps=3, workdir="/tmp"):
    """the true-Eigen variant of a9 / a11 / a12 (oracle/eigen_variant.cpp) on this scenario (MSCKF tracks only),
    when the box has Eigen3; {"eigen": "absent"} otherwise.  Returns the program's JSON as a dict."""
    import json
    import subprocess
    src = os.path.join(_HERE, "eigen_variant.cpp")
    exe = os.path.join(workdir, "xk_eigen_variant")
    inc = [a for d in ("/usr/include/eigen3", "/usr/local/include/eigen3", "/opt/eigen3") if os.path.isdir(d) for a in ("-I", d)]
    r = subprocess.run(["g++", "-std=c++17", "-O3", "-march=native", "-DNDEBUG"] + inc + ["-o", exe, src], capture_output=True, text=True)
    if r.returncode != 0:
        return {"eigen": "present, did not build", "error": r.stderr[-400:]}
    probe = subprocess.run([exe], capture_output=True, text=True)
    if '"absent"' in probe.stdout:
        return json.loads(probe.stdout.strip().splitlines()[-1])
    if "slam_anchor_idxs" in sc and len(sc["slam_anchor_idxs"]):
        return {"eigen": "present", "error": "the variant covers MSCKF tracks only"}
    _, _, _, info = msckf_update(sc)
    ref = visual_update(sc)
    n = sc["P"].shape[0]
    K = len(sc["trk_off"]) - 1
    parts = [np.array([n, K, sc["sigma_img"] ** 2, reps], float), np.asfortranarray(sc["P"]).ravel(order="F")]
    for k in range(K):
        jac, hf, res, bad = track_jacobians(sc, k, info["feats"][k])
        L = len(res) // 2
        parts += [np.array([L, chi2inv(0.95, 2 * L - 3), float(bad)]), jac.ravel(order="F"), hf.ravel(order="F"), res]
    parts += [np.asfortranarray(ref["P"]).ravel(order="F"), ref["correction"]]
    dump = os.path.join(workdir, "xk_eigen_dump.bin")
    np.concatenate(parts).astype("<f8").tofile(dump)
    r = subprocess.run([exe, dump, str(reps)], capture_output=True, text=True, timeout=600)
    os.remove(dump)
    try:
        return json.loads(r.stdout.strip().splitlines()[-1])
    except Exception:
        return {"eigen": "present", "error": (r.stdout + r.stderr)[-300:]}
