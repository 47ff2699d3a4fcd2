"""TEST INFRASTRUCTURE -- CPU restatement of the reference's place-recognition request filter and
descriptor matching.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg may import this module; the product path never does.

What is restated (reference file:line):
  VLAD::computeVLAD / computeScore            src/x/place_recognition/vlad.cpp:40-75
  Database::addKeyframe / findCandidate       src/x/place_recognition/database.cpp:30-61, database.h:70
  Keyframe (descriptor order, uav-id set)     src/x/place_recognition/keyframe.cpp:23-56
  PlaceRecognition::findCorrespondences       src/x/place_recognition/place_recognition.cpp:137-390 (non-GT branch:
      2-NN Hamming match, distance + ratio test, duplicate removal, MSCKF/SLAM/OPP classification; the RANSAC
      essential-matrix filter of :268-281 is an OpenCV call and enters here as an optional inlier mask)
Third-party pieces that are NOT part of /root/reference's own sources and are restated from their published
behaviour:
  DBoW3 (vendored, third_party/DBow3 @ the reference's pinned copy): Vocabulary::transform for binary descriptors
      (src/Vocabulary.cpp:880-914: greedy descent, children in file order, strict '<' so the first minimum wins),
      DescManip::distance_8uc1 (include/DBow3/DescManip.h:72-97: popcount of the XOR), getWord (Vocabulary.cpp:599).
  OpenCV BFMatcher(NORM_HAMMING)::knnMatch(k=2): for every query the two smallest distances in ascending
      (distance, train index) order -- the brute-force matcher keeps the earlier index on ties (strict comparisons
      while inserting into the top-k list).  OpenCV is absent from this image: this ordering is the documented /
      observed behaviour, not something run here.
PARITY PINNING: the vocabulary is pinned -- x_multi_agent_amd/data/vocab_*.npz are the reference's own Vocabulary/*.yaml data
files unpacked by tests/golden/make_vocab_fixture.py (with the reference's own QuickLZ decoder).  The reference has
no tests or golden vectors for VLAD / Database / findCorrespondences and cannot be built here (OpenCV, DBoW3 need
OpenCV), so the OUTPUTS of this module are "parity unpinned": they follow the cited lines, nothing more.
"""
import numpy as np

_POP8 = np.array([bin(i).count("1") for i in range(256)], dtype=np.int64)
MAX_KEYFRAMES = 15          # database.h:70


def hamming(a, b):
    """Number of differing bits of two uint8 arrays (DescManip::distance_8uc1 / cv::norm(.., NORM_HAMMING))."""
    return int(_POP8[np.bitwise_xor(np.asarray(a, np.uint8), np.asarray(b, np.uint8))].sum())


class Vocabulary:
    """The tree of a DBoW3 binary vocabulary, from an x_multi_agent_amd/data/vocab_*.npz file (or any dict of arrays)."""

    def __init__(self, v):
        self.k, self.L = int(v["k"]), int(v["L"])
        self.desc = np.asarray(v["desc"], np.uint8)
        self.children = np.asarray(v["children"], np.int32)
        self.word_of_node = np.asarray(v["word_of_node"], np.int32)
        self.node_of_word = np.asarray(v["node_of_word"], np.int32)
        self.d_length = self.desc.shape[1]                      # getDescritorSize(): bytes
        self.clusters_n = int(self.k ** self.L)                 # vlad.cpp:27-28
        self.v_length = self.clusters_n * self.d_length * 8     # vlad.cpp:29-31

    def transform(self, d):
        node = 0
        while True:
            best, nxt = None, -1
            for c in self.children[node]:
                if c < 0:
                    break
                dist = hamming(d, self.desc[c])
                if best is None or dist < best:
                    best, nxt = dist, int(c)
            node = nxt
            if self.children[node][0] < 0:                      # isLeaf()
                return int(self.word_of_node[node])

    def get_word(self, w):
        return self.desc[self.node_of_word[w]]


def compute_vlad(voc, x):
    """VLAD::computeVLAD (vlad.cpp:40-66): per descriptor, XOR with its closest centroid, OR into that cluster's row."""
    x = np.asarray(x, np.uint8).reshape(-1, voc.d_length)
    vlad = np.zeros((voc.clusters_n, voc.d_length), np.uint8)
    for t in range(x.shape[0]):
        w = voc.transform(x[t])
        vlad[w] |= np.bitwise_xor(x[t], voc.get_word(w))
    return vlad


def compute_score(voc, x, y):
    """VLAD::computeScore (vlad.cpp:68-75)."""
    return (voc.v_length - hamming(x, y)) / voc.v_length


class Keyframe:
    def __init__(self, descriptors, payload=None, tag=None):
        self.descriptors = np.asarray(descriptors, np.uint8)
        self.payload, self.tag = payload, tag
        self.vlad = None
        self.uav_ids = set()


class Database:
    """Database (database.cpp): at most 15 keyframes, oldest dropped first."""

    def __init__(self, voc, pr_score_thr):
        self.voc, self.thr = voc, float(pr_score_thr)
        self.keyframes = []

    def add_keyframe(self, kf):
        kf.vlad = compute_vlad(self.voc, kf.descriptors)        # database.cpp:52-54
        self.keyframes.append(kf)
        if len(self.keyframes) > MAX_KEYFRAMES:                 # :56-58
            self.keyframes.pop(0)

    def find_candidate(self, uav_id, query_vlad):
        """database.cpp:30-49 -> (keyframe or None, index in the store or -1, score of the winner or 0.0)."""
        score, best, best_i = 0.0, None, -1
        for i, kf in enumerate(self.keyframes):
            if uav_id in kf.uav_ids:
                continue
            s = compute_score(self.voc, query_vlad, kf.vlad)
            if s > self.thr and s > score:
                score, best, best_i = s, kf, i
        if best is not None:
            best.uav_ids.add(uav_id)
        return best, best_i, score


def knn2(query, train):
    """BFMatcher(NORM_HAMMING).knnMatch(query, train, k=2): (idx[nq,2], dist[nq,2]); -1 / huge where train is short."""
    query = np.asarray(query, np.uint8)
    train = np.asarray(train, np.uint8)
    nq, nt = len(query), len(train)
    idx = np.full((nq, 2), -1, np.int32)
    dist = np.full((nq, 2), 1 << 30, np.int32)
    for q in range(nq):
        d = _POP8[np.bitwise_xor(train, query[q][None, :])].sum(axis=1) if nt else np.zeros(0, np.int64)
        order = np.lexsort((np.arange(nt), d))[:2]
        for r, t in enumerate(order):
            idx[q, r], dist[q, r] = t, d[t]
    return idx, dist


def good_matches(idx, dist, min_distance, ratio_thr, inlier_mask=None):
    """place_recognition.cpp:252-301: distance + ratio test (float compares, as cv::DMatch::distance is float),
    optional RANSAC mask, then the reference's duplicate-removal loops, quirks included.  -> list of (query, train)."""
    good = []
    for q in range(len(idx)):
        if idx[q, 1] < 0:
            continue                                            # fewer than two neighbours: m[1] does not exist
        d0, d1 = np.float32(dist[q, 0]), np.float32(dist[q, 1])
        if d0 < min_distance and d0 < d1 * ratio_thr:           # :254-255 (float * double -> double)
            good.append((q, int(idx[q, 0])))
    if not good:
        return []
    if inlier_mask is not None:                                 # :275-281
        good = [m for m, keep in zip(good, inlier_mask) if keep]
    remove_ids = []                                             # :283-296
    for i in range(len(good)):
        for j in range(i, len(good)):
            if i != j and (good[i][0] == good[j][0] or good[i][1] == good[j][1]):
                remove_ids.append(j)
                break
    corr = 0                                                    # :297-301
    for r in remove_ids:
        pos = r - corr
        if 0 <= pos < len(good):
            del good[pos]
        # (an out-of-range erase is undefined behaviour in the reference; it cannot occur when every train index
        #  is claimed at most twice, which is what the tests feed)
        corr += 1
    return good


def classify(good, n_cur_msckf, n_cur_slam, n_rec_msckf, n_rec_slam):
    """place_recognition.cpp:311-388 -> list of (kind, current index within its list, received index within its list).
    kind: 'msckf' (received MSCKF x current OPP), 'slam' (SLAM x SLAM), 'opp_slam' (received SLAM x current OPP),
    'opp_opp' (OPP x OPP, becomes an MsckfMatch)."""
    max_cur_msckf, max_cur_slam = n_cur_msckf, n_cur_msckf + n_cur_slam
    max_rec_msckf, max_rec_slam = n_rec_msckf, n_rec_msckf + n_rec_slam
    out = []
    for q, t in good:
        if q < max_rec_msckf:
            if t >= max_cur_slam:
                out.append(("msckf", t - max_cur_slam, q))
        if max_rec_msckf <= q < max_rec_slam:
            if max_cur_msckf <= t < max_cur_slam:
                out.append(("slam", t - max_cur_msckf, q - max_rec_msckf))
            if t >= max_cur_slam:
                out.append(("opp_slam", t - max_cur_slam, q - max_rec_msckf))
        if q >= max_rec_slam:
            if t >= max_cur_slam:
                out.append(("opp_opp", t - max_cur_slam, q - max_rec_slam))
    return out


def is_keyframe(frames_since_last, position, last_position, inverse_depths, n_tracks):
    """Keyframe selection rule (vio_updater.cpp:452-470): more than 10 frames since the last one, parallax
    |dp| / mean depth > 0.15 and more than 10 tracks.  inverse_depths = the feature array (alpha, beta, rho per
    feature); the reference's loop starts at element 3 with stride 3 and divides by the feature count."""
    if frames_since_last <= 10:
        return False
    f = np.asarray(inverse_depths, float).ravel()
    med = 0.0
    for i in range(3, len(f), 3):
        if f[i] > 0.001:
            med += abs(1.0 / f[i])
    if len(f) == 0:
        return False
    med /= len(f) / 3.0
    diff = np.linalg.norm(np.asarray(position, float) - np.asarray(last_position, float))
    return bool(med > 0.0 and diff / med > 0.15 and n_tracks > 10)
