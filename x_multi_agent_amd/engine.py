"""ctypes binding of libxk.so -- the C ABI declared in include/xk.h.

This is plumbing for tests and bench.py; the product is the shared library.
There is NO fallback: if libxk.so (the hand-written HIP kernels for gfx950) is
missing or fails to load, importing this module's `lib()` raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("XK_LIB_PATH", os.path.join(_HERE, "libxk.so"))  # override: experiments only
LAB_LIB_PATH = os.path.join(_HERE, "lab", "libxk.so")   # the same sources with -DXK_LAB: test hooks, env switches, probes (include/xk_lab.h)
STRICT_LIB_PATH = os.path.join(_HERE, "libxk_strict.so")   # -DXK_SYNC_STRICT=1: release / acquire hand-offs, the bit-compare reference
_LIBS = {}

c_dp = C.POINTER(C.c_double)
c_ip = C.POINTER(C.c_int)

XK_NSTAGE = 6
STATUS = {0: "XK_OK", 1: "XK_EINVAL", 2: "XK_ESINGULAR", 3: "XK_ENAN", 4: "XK_EDEVICE", 5: "XK_ENOMEM",
          6: "XK_ECAPACITY"}

# every symbol include/xk.h declares (tests check the library exports them all)
SYMBOLS = [
    "xk_create", "xk_destroy", "xk_strerror", "xk_last_error", "xk_version", "xk_stream",
    "xk_stage_window", "xk_stage_tracks", "xk_stage_tracks_begin", "xk_stage_tracks_end", "xk_stage_slam", "xk_upload_P", "xk_download_P",
    "xk_msckf_build", "xk_qr_compress", "xk_apply_update", "xk_visual_update_staged", "xk_visual_update",
    "xk_apply_update_dense", "xk_apply_ci", "xk_fuse_ci_msckf", "xk_fuse_ci_slam", "xk_multi_slam_match", "xk_msckf_ci_track",
    "xk_ci_round_device", "xk_cov_congruence", "xk_cov_propagate",
    "xk_stage_msckf_slam", "xk_msckf_slam_results", "xk_init_msckf_slam_features", "xk_init_standard_slam_features",
    "xk_payload_doubles", "xk_pack_payload", "xk_bench_staged", "xk_run_steps",
    "xk_apply_ci_resident", "xk_snapshot_P", "xk_caqr_status", "xk_set_option", "xk_build_compress_async", "xk_build_compress_update_async", "xk_build_compress_update_pass_async", "xk_fetch_flags",
    "xk_pr_create", "xk_pr_destroy", "xk_pr_vlad_bytes", "xk_pr_size", "xk_pr_compute_vlad", "xk_pr_add_keyframe",
    "xk_pr_find_candidate", "xk_pr_keyframe", "xk_pr_copy_keyframe", "xk_pr_knn_match",
]


# what include/xk_lab.h adds (lab build only; the release library must NOT export them)
LAB_SYMBOLS = ["xk_is_lab", "xk_probe_fp64_peak", "xk_debug_persist_stamps"]


class XkTiming(C.Structure):
    _fields_ = [("total_ms", C.c_float), ("stage_ms", C.c_float * XK_NSTAGE),
                ("stage_launches", C.c_int * XK_NSTAGE), ("stage_name", (C.c_char * 32) * XK_NSTAGE),
                ("n", C.c_int), ("c1", C.c_int), ("k_tracks", C.c_int), ("rows_stacked", C.c_int),
                ("n_leaf", C.c_int), ("n_levels", C.c_int)]


class XkError(RuntimeError):
    def __init__(self, status, what, detail=""):
        super().__init__(f"{what}: {STATUS.get(status, status)} {detail}".strip())
        self.status = status


def _load(path):
    if not os.path.exists(path):
        raise ImportError(f"{path} not found: build it with `python -m x_multi_agent_amd.build` "
                          "(hipcc --offload-arch=gfx950); there is no CPU fallback")
    try:
        # PyTorch bundles its own HIP runtime; whichever libamdhip64 is loaded first serves the
        # whole process, so when torch is installed (it is the device-memory / RCCL plumbing of
        # bench.py and the tests) load it first -- the other order leaves torch without a GPU.
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(path)
    L.xk_strerror.restype = C.c_char_p
    L.xk_last_error.restype = C.c_char_p
    L.xk_last_error.argtypes = [C.c_void_p]
    L.xk_stream.restype = C.c_void_p
    L.xk_stream.argtypes = [C.c_void_p]
    L.xk_payload_doubles.restype = C.c_long
    return L


def lib(lab=False, path=None):
    """Load libxk.so (lab=True: lab/libxk.so, the -DXK_LAB build with the test hooks; path: any other build of the same ABI,
    e.g. STRICT_LIB_PATH) or raise -- never falls back to anything else.  Several builds can be loaded in one process."""
    p = os.path.abspath(path if path else (LAB_LIB_PATH if lab else LIB_PATH))
    if p not in _LIBS:
        _LIBS[p] = _load(p)
    return _LIBS[p]


def _d(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(c_dp)


def _i(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(c_ip)


def _f(a):
    a = np.asfortranarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(c_dp)


class Engine:
    """One xk_handle (= one agent / one x::Ekf)."""

    def __init__(self, n_poses_max, n_feat_max, k_max, device=0, lab=False, lib_path=None):
        self.L = lib(lab, lib_path)
        self.h = C.c_void_p()
        rc = self.L.xk_create(C.c_int(device), C.c_int(n_poses_max), C.c_int(n_feat_max), C.c_int(k_max),
                              C.byref(self.h))
        if rc != 0:
            raise XkError(rc, "xk_create")
        self.N, self.M, self.K = n_poses_max, n_feat_max, k_max
        self.n = 15 + 6 * n_poses_max + 3 * n_feat_max

    def close(self):
        if self.h:
            self.L.xk_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc, what):
        if rc != 0:
            raise XkError(rc, what, (self.L.xk_last_error(self.h) or b"").decode())

    # ---- staging -----------------------------------------------------
    def stage(self, sc):
        """Stage a synth.make_scenario dict in HBM."""
        q, qp = _d(sc["C_q_G"])
        p, pp = _d(sc["G_p_C"])
        self._chk(self.L.xk_stage_window(self.h, qp, pp, C.c_int(len(p))), "xk_stage_window")
        to, top = _i(sc["trk_off"])
        ob, obp = _d(sc["obs_xy"])
        self._chk(self.L.xk_stage_tracks(self.h, top, obp, C.c_int(len(to) - 1)), "xk_stage_tracks")
        M = len(sc["slam_anchor_idxs"]) if "slam_anchor_idxs" in sc else 0
        if M:
            f, fp = _d(sc["slam_feat"])
            a, ap = _i(sc["slam_anchor_idxs"])
            ts, tsp = _i(sc["slam_track_sizes"])
            z, zp = _d(sc["slam_z_last"])
            self._chk(self.L.xk_stage_slam(self.h, fp, ap, tsp, zp, C.c_int(M)), "xk_stage_slam")
        else:
            self._chk(self.L.xk_stage_slam(self.h, None, None, None, None, C.c_int(0)), "xk_stage_slam")
        self.upload_P(sc["P"])
        self._K, self._M = len(to) - 1, M

    def upload_P(self, P):
        Pf, Pp = _f(P)
        self._chk(self.L.xk_upload_P(self.h, Pp, C.c_int(self.n), C.c_int(self.n)), "xk_upload_P")

    def download_P(self):
        P = np.zeros((self.n, self.n), order="F")
        self._chk(self.L.xk_download_P(self.h, P.ctypes.data_as(c_dp), C.c_int(self.n), C.c_int(self.n)),
                  "xk_download_P")
        return np.ascontiguousarray(P)

    def snapshot_P(self, restore=False):
        """Keep (restore=False) / bring back (restore=True) a device-side copy of the resident covariance (xk_snapshot_P)."""
        self._chk(self.L.xk_snapshot_P(self.h, C.c_int(1 if restore else 0)), "xk_snapshot_P")

    # ---- staged path -------------------------------------------------
    def _flag_bufs(self):
        K, M = max(self._K, 1), max(self._M, 1)
        return (np.zeros(K, dtype=np.int32), np.zeros(K), np.zeros(M, dtype=np.int32), np.zeros(M))

    def msckf_build(self, sigma_img):
        inl, gam, inls, gams = self._flag_bufs()
        self._chk(self.L.xk_msckf_build(self.h, C.c_double(sigma_img), inl.ctypes.data_as(c_ip),
                                        gam.ctypes.data_as(c_dp), inls.ctypes.data_as(c_ip),
                                        gams.ctypes.data_as(c_dp)), "xk_msckf_build")
        return dict(inlier=inl[:self._K], gamma=gam[:self._K], inlier_slam=inls[:self._M],
                    gamma_slam=gams[:self._M])

    def qr_compress(self, want=True):
        if not want:
            self._chk(self.L.xk_qr_compress(self.h, None, C.c_int(0), None), "xk_qr_compress")
            return None, None
        T = np.zeros((self.n, self.n), order="F")
        z = np.zeros(self.n)
        self._chk(self.L.xk_qr_compress(self.h, T.ctypes.data_as(c_dp), C.c_int(self.n), z.ctypes.data_as(c_dp)),
                  "xk_qr_compress")
        return np.ascontiguousarray(T), z

    def build_compress_update_pass_async(self, sigma_img, corr_total=None, cov_update=True):
        """One pass of the iterated update queued whole (include/xk.h); collect it with apply_update(the same arguments)."""
        ctp = None
        if corr_total is not None:
            ct, ctp = _d(corr_total)
        self._chk(self.L.xk_build_compress_update_pass_async(self.h, C.c_double(sigma_img), ctp, C.c_int(int(cov_update))),
                  "xk_build_compress_update_pass_async")

    def apply_update(self, corr_total=None, cov_update=True):
        corr = np.zeros(self.n)
        ctp = None
        if corr_total is not None:
            ct, ctp = _d(corr_total)
        self._chk(self.L.xk_apply_update(self.h, ctp, C.c_int(int(cov_update)), corr.ctypes.data_as(c_dp)),
                  "xk_apply_update")
        return corr

    def visual_update_staged(self, sigma_img):
        corr = np.zeros(self.n)
        inl, gam, inls, gams = self._flag_bufs()
        self._chk(self.L.xk_visual_update_staged(self.h, C.c_double(sigma_img), corr.ctypes.data_as(c_dp),
                                                 inl.ctypes.data_as(c_ip), gam.ctypes.data_as(c_dp),
                                                 inls.ctypes.data_as(c_ip), gams.ctypes.data_as(c_dp)),
                  "xk_visual_update_staged")
        return dict(correction=corr, inlier=inl[:self._K], gamma=gam[:self._K], inlier_slam=inls[:self._M],
                    gamma_slam=gams[:self._M])

    def visual_update(self, sc):
        """Host-buffer convenience call (PCIe inclusive); returns dict with posterior P."""
        q, qp = _d(sc["C_q_G"])
        p, pp = _d(sc["G_p_C"])
        to, top = _i(sc["trk_off"])
        ob, obp = _d(sc["obs_xy"])
        K = len(to) - 1
        M = len(sc["slam_anchor_idxs"]) if "slam_anchor_idxs" in sc else 0
        if M:
            f, fp = _d(sc["slam_feat"])
            a, ap = _i(sc["slam_anchor_idxs"])
            ts, tsp = _i(sc["slam_track_sizes"])
            z, zp = _d(sc["slam_z_last"])
        else:
            fp = zp = None
            ap = tsp = None
        P = np.array(sc["P"], dtype=np.float64, order="F", copy=True)
        corr = np.zeros(self.n)
        self._K, self._M = K, M
        inl, gam, inls, gams = self._flag_bufs()
        self._chk(self.L.xk_visual_update(self.h, qp, pp, C.c_int(len(p)), top, obp, C.c_int(K), fp, ap, tsp, zp,
                                          C.c_int(M), P.ctypes.data_as(c_dp), C.c_int(self.n), C.c_int(self.n),
                                          C.c_double(sc["sigma_img"]), corr.ctypes.data_as(c_dp),
                                          inl.ctypes.data_as(c_ip), gam.ctypes.data_as(c_dp),
                                          inls.ctypes.data_as(c_ip), gams.ctypes.data_as(c_dp)),
                  "xk_visual_update")
        return dict(P=np.ascontiguousarray(P), correction=corr, inlier=inl[:K], gamma=gam[:K],
                    inlier_slam=inls[:M], gamma_slam=gams[:M])

    # ---- dense algebra -----------------------------------------------
    def apply_update_dense(self, P, H, res, r_diag, correction_total=None, cov_update=True):
        m = H.shape[0]
        Pf = np.array(P, dtype=np.float64, order="F", copy=True)
        Hf, Hp = _f(H)
        r, rp = _d(res)
        rd, rdp = _d(r_diag)
        corr = np.zeros(self.n)
        ct = None if correction_total is None else np.array(correction_total, dtype=np.float64)
        ctp = None if ct is None else ct.ctypes.data_as(c_dp)
        self._chk(self.L.xk_apply_update_dense(self.h, Pf.ctypes.data_as(c_dp), C.c_int(self.n), C.c_int(self.n),
                                               Hp, C.c_int(m), C.c_int(m), rp, rdp, ctp,
                                               C.c_int(int(cov_update)), corr.ctypes.data_as(c_dp)),
                  "xk_apply_update_dense")
        return np.ascontiguousarray(Pf), corr, ct

    def apply_ci(self, ci_P, H, res, S):
        m = H.shape[0]
        Pf, Pp = _f(ci_P)
        Hf, Hp = _f(H)
        r, rp = _d(res)
        Sf, Sp = _f(S)
        Po = np.zeros((self.n, self.n), order="F")
        corr = np.zeros(self.n)
        self._chk(self.L.xk_apply_ci(self.h, Po.ctypes.data_as(c_dp), C.c_int(self.n), Pp, C.c_int(self.n),
                                     C.c_int(self.n), Hp, C.c_int(m), C.c_int(m), rp, Sp, C.c_int(m),
                                     corr.ctypes.data_as(c_dp)), "xk_apply_ci")
        return np.ascontiguousarray(Po), corr

    def fuse_ci_slam(self, Pa, Ha, Pb, Hb, w):
        m = Ha.shape[0]
        Paf, Pap = _f(Pa)
        Haf, Hap = _f(Ha)
        Pbf, Pbp = _f(Pb)
        Hbf, Hbp = _f(Hb)
        S = np.zeros((m, m), order="F")
        wr = C.c_double()
        self._chk(self.L.xk_fuse_ci_slam(self.h, Pap, C.c_int(Pa.shape[0]), C.c_int(Pa.shape[0]), Hap, C.c_int(m),
                                         Pbp, C.c_int(Pb.shape[0]), C.c_int(Pb.shape[0]), Hbp, C.c_int(m),
                                         C.c_int(m), C.c_double(w), S.ctypes.data_as(c_dp), C.c_int(m),
                                         C.byref(wr)), "xk_fuse_ci_slam")
        return np.ascontiguousarray(S), wr.value

    def fuse_ci_msckf(self, P, H, Ps, Hs, w):
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
        self._chk(self.L.xk_fuse_ci_msckf(self.h, Pp, C.c_int(P.shape[0]), C.c_int(P.shape[0]), Hp, C.c_int(m),
                                          C.c_int(m), C.c_int(k), PsA, nsp, HsA, C.c_double(w),
                                          S.ctypes.data_as(c_dp), C.c_int(m), C.byref(wr)), "xk_fuse_ci_msckf")
        return np.ascontiguousarray(S), wr.value

    def multi_slam_match(self, C_q_G, G_p_C, feat, anchor_idx, feature_id, P, n_poses_max, o_C_q_G, o_G_p_C,
                         o_feat, o_anchor_idx, o_feature_id, o_P, o_n_poses_max, sigma_landmark, ci_slam_w):
        n, no = P.shape[0], o_P.shape[0]
        a = [_d(C_q_G), _d(G_p_C), _d(feat), _f(P), _d(o_C_q_G), _d(o_G_p_C), _d(o_feat), _f(o_P)]
        inl = C.c_int()
        gam = C.c_double()
        H = np.zeros((3, n), order="F")
        res = np.zeros(3)
        S = np.zeros((3, 3), order="F")
        Pj = np.zeros((n, n), order="F")
        self._chk(self.L.xk_multi_slam_match(
            self.h, a[0][1], a[1][1], C.c_int(len(a[1][0])), a[2][1], C.c_int(anchor_idx), C.c_int(feature_id),
            a[3][1], C.c_int(n), C.c_int(n), C.c_int(n_poses_max), a[4][1], a[5][1], C.c_int(len(a[5][0])),
            a[6][1], C.c_int(o_anchor_idx), C.c_int(o_feature_id), a[7][1], C.c_int(no), C.c_int(no),
            C.c_int(o_n_poses_max), C.c_double(sigma_landmark), C.c_double(ci_slam_w), C.byref(inl),
            C.byref(gam), H.ctypes.data_as(c_dp), C.c_int(3), res.ctypes.data_as(c_dp), S.ctypes.data_as(c_dp),
            Pj.ctypes.data_as(c_dp), C.c_int(n)), "xk_multi_slam_match")
        out = dict(inlier=bool(inl.value), gamma=gam.value, H=np.ascontiguousarray(H), res=res)
        if out["inlier"]:
            out.update(S=np.ascontiguousarray(S), P_j=np.ascontiguousarray(Pj))
        return out

    def msckf_ci_track(self, trk, C_q_G, G_p_C, P, n_poses_max, sigma_img, matches, ci_msckf_w):
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
        self._chk(self.L.xk_msckf_ci_track(
            self.h, base[0][1], C.c_int(len(trk)), base[1][1], base[2][1], C.c_int(len(G_p_C)), base[3][1],
            C.c_int(n), C.c_int(n), C.c_int(n_poses_max), C.c_double(sigma_img), C.c_int(k), arr(mo), mLp, arr(mq),
            arr(mp), mnpp, arr(mP), mnp_, C.c_double(ci_msckf_w), C.byref(si), C.byref(sg), C.byref(hc), C.byref(cg),
            H.ctypes.data_as(c_dp), C.c_int(m3), res.ctypes.data_as(c_dp), S.ctypes.data_as(c_dp), C.c_int(m3),
            Pj.ctypes.data_as(c_dp), C.c_int(n)), "xk_msckf_ci_track")
        out = dict(self_inlier=bool(si.value), self_gamma=sg.value, ci=None, ci_gamma=cg.value)
        if hc.value:
            out["ci"] = dict(S=np.ascontiguousarray(S), P_j=np.ascontiguousarray(Pj), H=np.ascontiguousarray(H), res=res)
        return out

    # ---- MSCKF-SLAM tracks / persistent-feature initialisation --------
    def stage_msckf_slam(self, tracks):
        """tracks: list of (L x 2) observation arrays (possibly empty)."""
        k = len(tracks)
        self.K2 = k
        if k == 0:
            self._chk(self.L.xk_stage_msckf_slam(self.h, None, None, C.c_int(0)), "xk_stage_msckf_slam")
            return
        off, offp = _i(np.concatenate([[0], np.cumsum([len(t) for t in tracks])]))
        obs, obsp = _d(np.concatenate([np.asarray(t, float).reshape(-1, 2) for t in tracks]))
        self._chk(self.L.xk_stage_msckf_slam(self.h, offp, obsp, C.c_int(k)), "xk_stage_msckf_slam")

    def msckf_slam_results(self):
        k = getattr(self, "K2", 0)
        m = max(3 * k, 1)
        inl = np.zeros(max(k, 1), np.int32)
        gam = np.zeros(max(k, 1))
        H1 = np.zeros((m, self.n), order="F")
        H2 = np.zeros((m, m), order="F")
        r1, f = np.zeros(m), np.zeros(m)
        self._chk(self.L.xk_msckf_slam_results(self.h, inl.ctypes.data_as(c_ip), gam.ctypes.data_as(c_dp),
                                               H1.ctypes.data_as(c_dp), C.c_int(m), H2.ctypes.data_as(c_dp), C.c_int(m),
                                               r1.ctypes.data_as(c_dp), f.ctypes.data_as(c_dp)), "xk_msckf_slam_results")
        return dict(inlier=inl[:k], gamma=gam[:k], H1=np.ascontiguousarray(H1[:3 * k]), H2=np.ascontiguousarray(H2[:3 * k, :3 * k]),
                    r1=r1[:3 * k], features=f[:3 * k])

    def init_msckf_slam_features(self, n_features, correction, sigma_img):
        k = getattr(self, "K2", 0)
        corr, corrp = _d(correction)
        out = np.zeros(max(3 * k, 1))
        self._chk(self.L.xk_init_msckf_slam_features(self.h, C.c_int(n_features), corrp, C.c_double(sigma_img),
                                                     out.ctypes.data_as(c_dp)), "xk_init_msckf_slam_features")
        return out[:3 * k]

    def init_standard_slam_features(self, n_features, k, sigma_img, sigma_rho_0):
        self._chk(self.L.xk_init_standard_slam_features(self.h, C.c_int(n_features), C.c_int(k), C.c_double(sigma_img),
                                                        C.c_double(sigma_rho_0)), "xk_init_standard_slam_features")

    def cov_congruence(self, J):
        """Resident P <- J P J^T; J dense (n x n, mostly zeros) or a (row_ptr, col_idx, val) CSR triple."""
        if isinstance(J, tuple):
            rp, ci, v = J
        else:
            J = np.asarray(J, dtype=np.float64)
            nz = [np.nonzero(row)[0] for row in J]
            rp = np.concatenate([[0], np.cumsum([len(z) for z in nz])])
            ci = np.concatenate(nz) if len(nz) else np.zeros(0, int)
            v = np.concatenate([J[i, z] for i, z in enumerate(nz)]) if len(nz) else np.zeros(0)
        rp_, rpp = _i(rp)
        ci_, cip = _i(ci if len(ci) else [0])
        v_, vp = _d(v if len(v) else [0.0])
        self._chk(self.L.xk_cov_congruence(self.h, rpp, cip, vp, C.c_int(int(rp_[-1]))), "xk_cov_congruence")

    def cov_propagate(self, f_d, q_d):
        """Resident P <- blkdiag(F_d, I) P blkdiag(F_d, I)^T + blkdiag(Q_d, 0) (propagateCovarianceMatrices)."""
        f, fp = _f(f_d)
        q, qp = _f(q_d)
        self._chk(self.L.xk_cov_propagate(self.h, fp, C.c_int(15), qp, C.c_int(15)), "xk_cov_propagate")

    def ci_round_device(self, payloads_ptr, payload_stride, world, self_rank, tracks_ptr, n_tracks, track_len,
                        n_poses_valid, self_track, sigma_img, ci_msckf_w, want_corrections=False):
        """CI round against gathered payloads that already sit in device memory (RCCL receive buffer).
        payloads_ptr / tracks_ptr are device addresses (e.g. torch tensor.data_ptr()).  Returns
        (n_fused, corrections or None); the resident covariance becomes the last fused posterior."""
        tl, tlp = _i(np.asarray(track_len, dtype=np.int32).ravel())
        nv, nvp = _i(np.asarray(n_poses_valid, dtype=np.int32).ravel())
        st, stp = _i(np.asarray(self_track, dtype=np.int32).ravel())
        nf = C.c_int()
        corr = np.zeros((max(n_tracks, 1), self.n)) if want_corrections else None
        self._chk(self.L.xk_ci_round_device(
            self.h, C.cast(C.c_void_p(payloads_ptr), c_dp), C.c_long(payload_stride), C.c_int(world), C.c_int(self_rank),
            C.cast(C.c_void_p(tracks_ptr), c_dp), C.c_int(n_tracks), tlp, nvp, stp, C.c_double(sigma_img),
            C.c_double(ci_msckf_w), C.byref(nf), corr.ctypes.data_as(c_dp) if want_corrections else None),
            "xk_ci_round_device")
        return nf.value, (corr[:nf.value] if want_corrections else None)

    # ---- measurement -------------------------------------------------
    def bench_staged(self, sigma_img, warmup, steps):
        t = XkTiming()
        self._chk(self.L.xk_bench_staged(self.h, C.c_double(sigma_img), C.c_int(warmup), C.c_int(steps),
                                         C.byref(t)), "xk_bench_staged")
        return dict(total_ms=t.total_ms,
                    stages={t.stage_name[s].value.decode(): dict(ms=t.stage_ms[s], launches=t.stage_launches[s])
                            for s in range(XK_NSTAGE)},
                    n=t.n, c1=t.c1, k_tracks=t.k_tracks, rows_stacked=t.rows_stacked, n_leaf=t.n_leaf,
                    n_levels=t.n_levels)

    def caqr_status(self):
        """xk_caqr_status: schedule of the last compression (0 multi-launch, 1 resident, 2 pipelined), whether the
        single-launch path is armed, give-ups so far, reason of the last one."""
        v = [C.c_int() for _ in range(4)]
        self._chk(self.L.xk_caqr_status(self.h, *[C.byref(x) for x in v]), "xk_caqr_status")
        return dict(schedule=v[0].value, armed=bool(v[1].value), giveups=v[2].value, last_reason=v[3].value)

    def set_option(self, name, value):
        """xk_set_option: "caqr_resident", "caqr_rearm" (release library); the lab build (Engine(..., lab=True)) adds the test hooks and
        A/B switches "caqr_poison", "caqr_test_stall", "caqr_tall26", "pipe_kalman" (include/xk_lab.h)."""
        self._chk(self.L.xk_set_option(self.h, name.encode(), C.c_int(int(value))), "xk_set_option")

    def probe_fp64_peak(self, use_mfma=True):
        t = C.c_double()
        self._chk(self.L.xk_probe_fp64_peak(self.h, C.c_int(int(use_mfma)), C.byref(t)), "xk_probe_fp64_peak")
        return t.value

    def payload_doubles(self):
        return int(self.L.xk_payload_doubles(C.c_int(self.N), C.c_int(self.M)))

    def pack_payload_into(self, agent_id, timestamp, dyn16, device_ptr=None):
        """Pack the SimpleState payload into a caller-owned DEVICE buffer (e.g. a torch tensor's data_ptr)."""
        d, dp = _d(dyn16)
        ptr = c_dp()
        dst = C.cast(C.c_void_p(device_ptr), c_dp) if device_ptr else None
        self._chk(self.L.xk_pack_payload(self.h, C.c_double(agent_id), C.c_double(timestamp), dp, dst,
                                         C.byref(ptr)), "xk_pack_payload")
        return C.cast(ptr, C.c_void_p).value

    def run_steps(self, sigma_img, steps):
        self._chk(self.L.xk_run_steps(self.h, C.c_double(sigma_img), C.c_int(steps)), "xk_run_steps")


def LabEngine(*a, **kw):
    """An Engine on the lab build of the library (x_multi_agent_amd/lab/libxk.so, -DXK_LAB): what tests and tools/exp use when they
    need a test hook, an environment switch or a probe kernel.  Both libraries can be loaded in one process."""
    return Engine(*a, lab=True, **kw)
