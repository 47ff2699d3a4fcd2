"""Place-recognition request filter and keyframe store on the device -- ctypes binding of the xk_pr_* entry
points (include/xk.h).  Mirrors the reference's `Database` / `VLAD` / `Keyframe`
(src/x/place_recognition/{database,vlad,keyframe}.cpp) and the matching front half of
`PlaceRecognition::findCorrespondences` (place_recognition.cpp:137-390).

No fallback: everything numeric runs in libxk.so's HIP kernels; the host part here is the same bookkeeping the
reference does on the host (ratio test, duplicate removal, MSCKF / SLAM / OPP classification)."""
import ctypes as C
import os

import numpy as np

from .engine import XkError, c_dp, c_ip

c_ub = C.POINTER(C.c_ubyte)
DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")


def load_vocabulary(name="visual", path=None):
    """A DBoW3 vocabulary as unpacked arrays (k, L, desc, children, word_of_node, node_of_word).
    path: an .npz a deployment unpacked from its own Vocabulary/*.yaml (tools in tests/golden/make_vocab_fixture.py);
    default: the package's own copy of the reference's vocabulary `name` (Vocabulary/<name>_voc_3_4_dbow3.yaml),
    x_multi_agent_amd/data/vocab_<name>.npz."""
    return dict(np.load(path if path else os.path.join(DATA, f"vocab_{name}.npz")))


def _u8(a):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    return a, a.ctypes.data_as(c_ub)


class Database:
    """x::Database on one agent's GPU: the keyframes (payload, tracks, descriptors, VLAD) live in HBM."""

    def __init__(self, eng, voc, pr_score_thr, payload_doubles=0, tracks_doubles=0, max_desc=1024):
        self.eng, self.L = eng, eng.L
        self.thr = float(pr_score_thr)
        self.k, self.Lv = int(voc["k"]), int(voc["L"])
        desc, dp = _u8(voc["desc"])
        ch = np.ascontiguousarray(voc["children"], np.int32)
        won = np.ascontiguousarray(voc["word_of_node"], np.int32)
        now = np.ascontiguousarray(voc["node_of_word"], np.int32)
        self.desc_bytes = desc.shape[1]
        self.p = C.c_void_p()
        rc = self.L.xk_pr_create(eng.h, C.c_int(self.k), C.c_int(self.Lv), C.c_int(desc.shape[0]), C.c_int(ch.shape[1]),
                                 C.c_int(self.desc_bytes), dp, ch.ctypes.data_as(c_ip), won.ctypes.data_as(c_ip),
                                 now.ctypes.data_as(c_ip), C.c_int(len(now)), C.c_long(payload_doubles),
                                 C.c_long(tracks_doubles), C.c_int(max_desc), C.byref(self.p))
        if rc != 0:
            raise XkError(rc, "xk_pr_create", (self.L.xk_last_error(eng.h) or b"").decode())
        self.vlad_bytes = int(self.L.xk_pr_vlad_bytes(self.p))
        self.clusters = self.vlad_bytes // self.desc_bytes
        self.max_desc = max_desc

    def close(self):
        if self.p:
            self.L.xk_pr_destroy(self.p)
            self.p = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc, what):
        if rc != 0:
            raise XkError(rc, what, (self.L.xk_last_error(self.eng.h) or b"").decode())

    def __len__(self):
        return int(self.L.xk_pr_size(self.p))

    def compute_vlad(self, descriptors):
        """VLAD::computeVLAD -> uint8 [clusters, desc_bytes]."""
        d, dp = _u8(np.asarray(descriptors, np.uint8).reshape(-1, self.desc_bytes))
        out = np.zeros(self.vlad_bytes, np.uint8)
        self._chk(self.L.xk_pr_compute_vlad(self.p, dp, C.c_int(d.shape[0]), out.ctypes.data_as(c_ub)), "xk_pr_compute_vlad")
        return out.reshape(self.clusters, self.desc_bytes)

    def add_keyframe(self, descriptors, payload_ptr=None, tracks_ptr=None, tag=0):
        d, dp = _u8(np.asarray(descriptors, np.uint8).reshape(-1, self.desc_bytes))
        pp = C.cast(C.c_void_p(payload_ptr), c_dp) if payload_ptr else None
        tp = C.cast(C.c_void_p(tracks_ptr), c_dp) if tracks_ptr else None
        self._chk(self.L.xk_pr_add_keyframe(self.p, dp, C.c_int(d.shape[0]), pp, tp, C.c_long(tag)), "xk_pr_add_keyframe")

    def find_candidate(self, uav_id, query_vlad):
        """Database::findCandidate -> (index or -1, score, tag)."""
        q, qp = _u8(np.asarray(query_vlad, np.uint8).ravel())
        if q.size != self.vlad_bytes:
            raise ValueError("query VLAD has the wrong size")
        idx, sc, tag = C.c_int(-1), C.c_double(0.0), C.c_long(-1)
        self._chk(self.L.xk_pr_find_candidate(self.p, C.c_int(uav_id), qp, C.c_double(self.thr), C.byref(idx), C.byref(sc),
                                              C.byref(tag)), "xk_pr_find_candidate")
        return idx.value, sc.value, tag.value

    def keyframe(self, index, want_descriptors=False):
        """-> dict(payload_ptr, tracks_ptr, n_desc, tag[, descriptors]) of the stored keyframe (device pointers)."""
        pp, tp = c_dp(), c_dp()
        nd, tag = C.c_int(0), C.c_long(0)
        buf = np.zeros((self.max_desc, self.desc_bytes), np.uint8) if want_descriptors else None
        self._chk(self.L.xk_pr_keyframe(self.p, C.c_int(index), C.byref(pp), C.byref(tp), C.byref(nd), C.byref(tag),
                                        buf.ctypes.data_as(c_ub) if want_descriptors else None), "xk_pr_keyframe")
        out = dict(payload_ptr=C.cast(pp, C.c_void_p).value, tracks_ptr=C.cast(tp, C.c_void_p).value, n_desc=nd.value,
                   tag=tag.value)
        if want_descriptors:
            out["descriptors"] = buf[:nd.value].copy()
        return out

    def copy_keyframe(self, index, payload_dst_ptr=None, tracks_dst_ptr=None):
        """The stored keyframe's payload / tracks into caller-owned DEVICE buffers (the response's send buffer)."""
        pp = C.cast(C.c_void_p(payload_dst_ptr), c_dp) if payload_dst_ptr else None
        tp = C.cast(C.c_void_p(tracks_dst_ptr), c_dp) if tracks_dst_ptr else None
        self._chk(self.L.xk_pr_copy_keyframe(self.p, C.c_int(index), pp, tp), "xk_pr_copy_keyframe")

    def knn_match(self, query, train):
        """BFMatcher(NORM_HAMMING).knnMatch(query, train, 2) -> (idx [nq,2], dist [nq,2])."""
        q, qp = _u8(np.asarray(query, np.uint8).reshape(-1, self.desc_bytes))
        t, tp = _u8(np.asarray(train, np.uint8).reshape(-1, self.desc_bytes))
        idx = np.full((q.shape[0], 2), -1, np.int32)
        dist = np.zeros((q.shape[0], 2), np.int32)
        self._chk(self.L.xk_pr_knn_match(self.p, qp, C.c_int(q.shape[0]), tp, C.c_int(t.shape[0]), idx.ctypes.data_as(c_ip),
                                         dist.ctypes.data_as(c_ip)), "xk_pr_knn_match")
        return idx, dist


def good_matches(idx, dist, min_distance, ratio_thr, inlier_mask=None):
    """Host half of findCorrespondences (place_recognition.cpp:252-301): distance + ratio test on the device's
    2-NN result, optional RANSAC inlier mask (the essential-matrix filter is the caller's), duplicate removal."""
    good = []
    for q in range(len(idx)):
        if idx[q, 1] < 0:
            continue
        d0, d1 = np.float32(dist[q, 0]), np.float32(dist[q, 1])
        if d0 < min_distance and d0 < d1 * ratio_thr:
            good.append((q, int(idx[q, 0])))
    if not good:
        return []
    if inlier_mask is not None:
        good = [m for m, keep in zip(good, inlier_mask) if keep]
    remove_ids = []
    for i in range(len(good)):
        for j in range(i, len(good)):
            if i != j and (good[i][0] == good[j][0] or good[i][1] == good[j][1]):
                remove_ids.append(j)
                break
    corr = 0
    for r in remove_ids:
        pos = r - corr
        if 0 <= pos < len(good):
            del good[pos]
        corr += 1
    return good


def classify(good, n_cur_msckf, n_cur_slam, n_rec_msckf, n_rec_slam):
    """place_recognition.cpp:311-388: which kind of collaborative match each (received, current) pair is."""
    max_cur_msckf, max_cur_slam = n_cur_msckf, n_cur_msckf + n_cur_slam
    max_rec_msckf, max_rec_slam = n_rec_msckf, n_rec_msckf + n_rec_slam
    out = []
    for q, t in good:
        if q < max_rec_msckf and t >= max_cur_slam:
            out.append(("msckf", t - max_cur_slam, q))
        if max_rec_msckf <= q < max_rec_slam:
            if max_cur_msckf <= t < max_cur_slam:
                out.append(("slam", t - max_cur_msckf, q - max_rec_msckf))
            if t >= max_cur_slam:
                out.append(("opp_slam", t - max_cur_slam, q - max_rec_msckf))
        if q >= max_rec_slam and t >= max_cur_slam:
            out.append(("opp_opp", t - max_cur_slam, q - max_rec_slam))
    return out
