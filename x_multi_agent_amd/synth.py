"""Deterministic synthetic workloads for the xVIO EKF-update path.

Sliding window of N camera poses on a circle (radius 5 m, 1 m/s, 30 Hz
frames), K MSCKF landmarks seen in every window frame (+ M persistent SLAM
features), ideal pinhole observations + N(0, sigma_img^2), window estimate =
truth minus an error drawn from the prior P, and a prior P grown by a small
clone-and-propagate covariance recursion so that consecutive clones are
strongly correlated, as in a running filter.

The PRNG is a counter-mode splitmix64 (vectorised, platform independent) with
Box-Muller normals, so the same seed gives the same bits everywhere.
Seed convention: 0x5EED0000 + 1000*config + agent_id.
"""
import numpy as np

K_CORE = 15
_M64 = (1 << 64) - 1


class SplitMix:
    """Counter-mode splitmix64: value i = mix(seed + (i+1)*golden)."""

    def __init__(self, seed):
        self.seed = np.uint64(seed & _M64)
        self.ctr = 0

    def u64(self, n):
        with np.errstate(over="ignore"):
            i = np.arange(self.ctr + 1, self.ctr + n + 1, dtype=np.uint64)
            z = self.seed + i * np.uint64(0x9E3779B97F4A7C15)
            z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
            z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
            z = z ^ (z >> np.uint64(31))
        self.ctr += n
        return z

    def uniform(self, n):
        return (self.u64(n) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)

    def normal(self, n):
        m = (n + 1) // 2
        u1 = 1.0 - self.uniform(m)  # (0,1]
        u2 = self.uniform(m)
        r = np.sqrt(-2.0 * np.log(u1))
        out = np.concatenate([r * np.cos(2 * np.pi * u2), r * np.sin(2 * np.pi * u2)])
        return out[:n]


def seed_for(config, agent_id=0):
    return 0x5EED0000 + 1000 * int(config) + int(agent_id)


def _rot_to_quat_xyzw(R):
    """Rotation matrix (camera->world) to unit quaternion (x,y,z,w), w >= 0."""
    t = np.trace(R)
    if t > 0:
        s = np.sqrt(t + 1.0) * 2
        w, x, y, z = 0.25 * s, (R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s
    elif R[0, 0] > R[1, 1] and R[0, 0] > R[2, 2]:
        s = np.sqrt(1.0 + R[0, 0] - R[1, 1] - R[2, 2]) * 2
        w, x, y, z = (R[2, 1] - R[1, 2]) / s, 0.25 * s, (R[0, 1] + R[1, 0]) / s, (R[0, 2] + R[2, 0]) / s
    elif R[1, 1] > R[2, 2]:
        s = np.sqrt(1.0 + R[1, 1] - R[0, 0] - R[2, 2]) * 2
        w, x, y, z = (R[0, 2] - R[2, 0]) / s, (R[0, 1] + R[1, 0]) / s, 0.25 * s, (R[1, 2] + R[2, 1]) / s
    else:
        s = np.sqrt(1.0 + R[2, 2] - R[0, 0] - R[1, 1]) * 2
        w, x, y, z = (R[1, 0] - R[0, 1]) / s, (R[0, 2] + R[2, 0]) / s, (R[1, 2] + R[2, 1]) / s, 0.25 * s
    q = np.array([x, y, z, w])
    if w < 0:
        q = -q
    return q / np.linalg.norm(q)


def _quat_to_rot(q):
    x, y, z, w = q / np.linalg.norm(q)
    return np.array([
        [1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
        [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
        [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)],
    ])


def _small_rot(dth):
    a = np.linalg.norm(dth)
    if a == 0:
        return np.eye(3)
    k = dth / a
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(a) * K + (1 - np.cos(a)) * K @ K


def true_poses(n_frames, phase0=0.3, agent_offset=0.0):
    """Camera on a circle looking radially outward; returns (R_list cam->world, p_list)."""
    radius, speed, fps = 5.0, 1.0, 30.0
    Rs, ps = [], []
    for i in range(n_frames):
        t = i / fps
        th = phase0 + agent_offset + speed * t / radius
        p = np.array([radius * np.cos(th), radius * np.sin(th), 1.5 + 0.05 * np.sin(2.0 * t)])
        zc = np.array([np.cos(th), np.sin(th), 0.0])
        xc = np.array([np.sin(th), -np.cos(th), 0.0])
        yc = np.cross(zc, xc)
        R = np.column_stack([xc, yc, zc])
        R = R @ _small_rot(np.array([0.02 * np.sin(3.0 * t), 0.015 * np.cos(2.5 * t), 0.01 * np.sin(1.7 * t)]))
        Rs.append(R)
        ps.append(p)
    return Rs, ps


def prior_covariance(n_poses_max, n_feat, rng, kind="filter", scale=1.0):
    """SPD prior over n = 15 + 6N + 3M error states.

    kind="filter": clone-and-propagate recursion (realistic cross terms);
    kind="stress": sigma^2 (B B^T + I) with random B."""
    N, M = n_poses_max, n_feat
    n = K_CORE + 6 * N + 3 * M
    if kind == "stress":
        B = rng.normal(n * n).reshape(n, n)
        return (1e-4 * scale) * (B @ B.T / n + np.eye(n))
    P = np.zeros((n, n))
    sig = np.array([0.02] * 3 + [0.02] * 3 + [0.004] * 3 + [2e-4] * 3 + [2e-3] * 3) * np.sqrt(scale)
    P[:15, :15] = np.diag(sig ** 2)
    dt = 1.0 / 30.0
    F = np.eye(15)
    F[0:3, 3:6] = dt * np.eye(3)
    A = rng.normal(9).reshape(3, 3)
    F[3:6, 6:9] = dt * 3.0 * (A - A.T)          # -R [a]x dt (skew, order 10 m/s^2 * dt)
    F[3:6, 12:15] = -dt * np.eye(3)
    F[6:9, 9:12] = -dt * np.eye(3)
    q = np.array([1e-8] * 3 + [2e-5] * 3 + [4e-7] * 3 + [1e-10] * 3 + [1e-8] * 3) * scale
    for i in range(N):
        # propagate core block and its cross terms with the clones so far
        nv = K_CORE + 6 * N
        P[:15, :nv] = F @ P[:15, :nv]
        P[:nv, :15] = P[:nv, :15] @ F.T
        P[:15, :15] += np.diag(q)
        # clone pose i (position <- core p, attitude <- core theta)
        cp, ca = K_CORE + 3 * i, K_CORE + 3 * N + 3 * i
        for dst, src in ((cp, 0), (ca, 6)):
            P[dst:dst + 3, :] = P[src:src + 3, :]
            P[:, dst:dst + 3] = P[:, src:src + 3]
            P[dst:dst + 3, dst:dst + 3] = P[src:src + 3, src:src + 3]
        P = 0.5 * (P + P.T)
    if M:
        nx = K_CORE + 6 * N
        C = 0.05 * rng.normal(3 * M * nx).reshape(3 * M, nx)
        Pxx = P[:nx, :nx]
        D = np.diag(np.tile(np.array([2e-3, 2e-3, 5e-3]) ** 2, M)) * scale
        P[nx:, :nx] = C @ Pxx
        P[:nx, nx:] = (C @ Pxx).T
        P[nx:, nx:] = C @ Pxx @ C.T + D
    # tiny jitter keeps it strictly PD (clone of a clone is otherwise singular)
    P += 1e-12 * scale * np.eye(n)
    return 0.5 * (P + P.T)


def _sample_error(P, rng):
    """Draw e ~ N(0, P) via eigen-decomposition (P may be near-singular)."""
    w, V = np.linalg.eigh(P)
    w = np.clip(w, 0.0, None)
    return V @ (np.sqrt(w) * rng.normal(P.shape[0]))


def make_scenario(n_poses_max, n_msckf, n_slam=0, seed=0, sigma_img=1.0 / 500.0,
                  n_poses=None, track_len=None, outlier_frac=0.05, prior_kind="filter",
                  prior_scale=1.0, agent_offset=0.0, landmarks=None, err_scale=1.0):
    """Build one visual-update problem.

    Returns a dict of plain numpy arrays:
      C_q_G [n_poses,4] xyzw, G_p_C [n_poses,3]  -- estimated window lists
      trk_off [K+1] int32, obs_xy [sum L_k, 2]   -- ragged MSCKF tracks
      P [n,n] prior, n_poses_max, sigma_img
      slam_* (if n_slam): feat [3M], anchor_idxs [M], z_last [M,2], track_sizes [M]
      landmarks_true [K+M,3]
    err_scale: the window / feature error is err_scale x a draw from the prior P (1.0: a draw from P itself).  The gate's
      S = H0 P H0^T + sigma^2 I is built from the observability-constrained Jacobians (msckf_update.cpp:393-406), which do not
      model the error components they project out: with a full-size draw only ~86 % of the CLEAN full-window tracks pass the
      95 % gate at the headline size; at 0.3 (a conservatively tuned filter) 97 % do -- the "nominal work" scenario of bench.py.
    """
    rng = SplitMix(seed)
    N = n_poses_max
    npz = N if n_poses is None else n_poses
    K, M = n_msckf, n_slam
    n = K_CORE + 6 * N + 3 * M
    Rs, ps = true_poses(npz, agent_offset=agent_offset)
    mid = npz // 2
    # landmarks in the mid camera's field of view, 4..20 m deep
    if landmarks is None:
        u = rng.uniform(3 * (K + M)).reshape(K + M, 3)
        depth = 4.0 + 16.0 * u[:, 2]
        pc = np.column_stack([(u[:, 0] - 0.5) * depth, (u[:, 1] - 0.5) * 0.8 * depth, depth])
        lm = (Rs[mid] @ pc.T).T + ps[mid]
    else:
        lm = np.asarray(landmarks, float)
        rng.uniform(3 * (K + M))  # keep the stream position independent of the branch
    P = prior_covariance(N, M, rng, kind=prior_kind, scale=prior_scale)
    err = _sample_error(P, rng)
    if err_scale != 1.0:
        err = err_scale * err
    # estimated window = truth (-) error   (true = est (+) err)
    q_est = np.zeros((npz, 4))
    p_est = np.zeros((npz, 3))
    for i in range(npz):
        dp = err[K_CORE + 3 * i:K_CORE + 3 * i + 3]
        dth = err[K_CORE + 3 * N + 3 * i:K_CORE + 3 * N + 3 * i + 3]
        p_est[i] = ps[i] - dp
        q_est[i] = _rot_to_quat_xyzw(Rs[i] @ _small_rot(dth).T)
    # MSCKF tracks: last L_k poses
    if track_len is None:
        lens = np.full(K, npz, dtype=np.int64)
    elif np.isscalar(track_len):
        lens = np.full(K, int(track_len), dtype=np.int64)
    else:
        lo, hi = track_len
        lens = lo + (rng.uniform(K) * (hi - lo + 1)).astype(np.int64)
        lens = np.clip(lens, lo, hi)
    trk_off = np.zeros(K + 1, dtype=np.int32)
    trk_off[1:] = np.cumsum(2 * 0 + lens)
    obs = np.zeros((int(trk_off[-1]), 2))
    noise = rng.normal(2 * obs.shape[0]).reshape(-1, 2) * sigma_img
    bad = rng.uniform(K) < outlier_frac
    bad_mag = 15.0 + 30.0 * rng.uniform(K)
    for k in range(K):
        L = int(lens[k])
        for i in range(L):
            pos = npz - L + i
            c = Rs[pos].T @ (lm[k] - ps[pos])
            obs[trk_off[k] + i] = c[:2] / c[2]
        if bad[k]:
            # gross mismatch on the second half of the track
            noise[trk_off[k] + L // 2:trk_off[k] + L] += bad_mag[k] * sigma_img
    obs += noise
    out = dict(C_q_G=q_est, G_p_C=p_est, trk_off=trk_off, obs_xy=obs, P=P,
               n_poses_max=N, n_poses=npz, sigma_img=float(sigma_img), n=n,
               landmarks_true=lm, seed=int(seed), R_true=np.array(Rs), p_true=np.array(ps))
    if M:
        # SLAM features: inverse-depth in an anchor pose, estimate = truth - error
        anchors = (rng.uniform(M) * npz).astype(np.int32)
        anchors = np.clip(anchors, 0, npz - 1)
        feat = np.zeros(3 * M)
        z_last = np.zeros((M, 2))
        zn = rng.normal(2 * M).reshape(M, 2) * sigma_img
        for j in range(M):
            a = int(anchors[j])
            c = Rs[a].T @ (lm[K + j] - ps[a])
            true_ivd = np.array([c[0] / c[2], c[1] / c[2], 1.0 / c[2]])
            feat[3 * j:3 * j + 3] = true_ivd - err[K_CORE + 6 * N + 3 * j:K_CORE + 6 * N + 3 * j + 3]
            cl = Rs[-1].T @ (lm[K + j] - ps[-1])
            z_last[j] = cl[:2] / cl[2] + zn[j]
        out.update(slam_feat=feat, slam_anchor_idxs=anchors, slam_z_last=z_last,
                   slam_track_sizes=np.full(M, npz, dtype=np.int32))
    return out


def tracks_as_list(sc):
    """Ragged obs -> list of [L_k,2] arrays (oracle-side convenience)."""
    off = sc["trk_off"]
    return [sc["obs_xy"][off[k]:off[k + 1]] for k in range(len(off) - 1)]


CONFIGS = {
    # BASELINE.json configs: (n_poses_max, n_msckf, n_slam)
    1: (10, 50, 0),
    2: (30, 200, 50),
    3: (50, 800, 0),
    4: (30, 400, 0),
    5: (30, 400, 0),
}


def make_config(cfg, agent_id=0, **kw):
    N, K, M = CONFIGS[cfg]
    return make_scenario(N, K, M, seed=seed_for(cfg, agent_id), **kw)


def make_manage_sequence(n_poses_max, n_feat_max, n_steps, seed, start_poses=0, start_features=0, removals=None):
    """Inputs of a run of StateManager::manage calls (state_manager.cpp:31-149): an IMU/camera rig moving along
    the usual circle, a covariance that starts as a dense SPD matrix, `start_features` persistent features
    anchored in the first poses of a partly filled window, and per-step lists of features to delete.
    Returns dict(N, M, init=dict(cov, q_array, p_array, f_array, sm=dict(...)), steps=[dict(p, q, q_ic, p_ic, del)])."""
    N, M = n_poses_max, n_feat_max
    n = 15 + 6 * N + 3 * M
    rng = SplitMix(seed)
    removals = removals or {}
    A = rng.normal(n * n).reshape(n, n)
    cov = (A @ A.T) / n * 1e-2 + np.diag(1e-3 + 1e-2 * rng.uniform(n))
    Rs, ps = true_poses(start_poses + n_steps + 1)
    q_ic = _rot_to_quat_xyzw(_small_rot(np.array([0.03, -0.02, 0.05])) @ np.array([[0.0, 0, 1], [-1, 0, 0], [0, -1, 0]]))
    p_ic = np.array([0.05, -0.02, 0.01])
    R_ic = _quat_to_rot(q_ic)
    q_array, p_array, f_array = np.zeros(4 * N), np.zeros(3 * N), np.zeros(3 * M)
    for i in range(start_poses):            # window slots already occupied: camera poses of the first frames
        q_array[4 * i:4 * i + 4] = _rot_to_quat_xyzw(Rs[i])
        p_array[3 * i:3 * i + 3] = ps[i]
    anchors = [-1] * M
    for j in range(start_features):
        anchors[j] = j % max(start_poses, 1) if j % 2 == 0 else 0
        f_array[3 * j:3 * j + 3] = [0.1 * (rng.uniform(1)[0] - 0.5), 0.1 * (rng.uniform(1)[0] - 0.5), 1.0 / (3.0 + 4.0 * rng.uniform(1)[0])]
    init = dict(cov=cov, q_array=q_array, p_array=p_array, f_array=f_array,
                sm=dict(n_poses=start_poses, n_features=start_features, n_poses_max=N, n_features_max=M,
                        anchor_idxs=anchors, filled_before=False))
    steps = []
    for s in range(n_steps):
        Rc, pc = Rs[start_poses + s], ps[start_poses + s]      # camera pose; IMU pose follows from the extrinsics
        R_i = Rc @ R_ic.T
        steps.append(dict(p=pc - R_i @ p_ic, q=_rot_to_quat_xyzw(R_i), q_ic=q_ic, p_ic=p_ic, **{"del": list(removals.get(s, []))}))
    return dict(N=N, M=M, init=init, steps=steps)


MANAGE_SEQUENCES = {
    # window fills from empty (zero-Jacobian rows of the never-filled slots), then slides
    "manage_empty_n4_m0": dict(n_poses_max=4, n_feat_max=0, n_steps=7, seed=0x5EED3001),
    # running filter: partly filled window, features anchored in the oldest poses, removals along the way
    "manage_feats_n5_m4": dict(n_poses_max=5, n_feat_max=4, n_steps=8, seed=0x5EED3002, start_poses=3, start_features=4,
                               removals={1: [1], 4: [0, 2]}),
    # headline-sized state
    "manage_n30_m0": dict(n_poses_max=30, n_feat_max=0, n_steps=3, seed=0x5EED3003, start_poses=29),
}


# ---- place recognition: synthetic binary descriptors and vocabularies ------------------
def make_vocabulary(k, L, desc_bytes=32, seed=7, prune=0.0):
    """A random DBoW3-shaped vocabulary tree: every inner node has up to k children (a fraction `prune` of the
    non-first children is dropped, so child counts differ), depth L, random binary node descriptors; leaves are the
    words, numbered in depth-first file order like Vocabulary::toStream writes them."""
    rng = np.random.default_rng(seed)
    parent, depth = [-1], [0]
    children = [[]]
    stack = [0]
    while stack:                                   # depth-first, the order DBoW3 saves nodes in
        pid = stack.pop()
        if depth[pid] == L:
            continue
        kids = []
        for c in range(k):
            if c > 0 and rng.random() < prune:
                continue
            nid = len(parent)
            parent.append(pid); depth.append(depth[pid] + 1); children.append([])
            kids.append(nid)
        children[pid] = kids
        stack.extend(reversed(kids))
    nn = len(parent)
    desc = rng.integers(0, 256, size=(nn, desc_bytes), dtype=np.uint8)
    ch = np.full((nn, k), -1, np.int32)
    for i, c in enumerate(children):
        ch[i, :len(c)] = c
    leaves = [i for i in range(1, nn) if not children[i]]
    word_of_node = np.full(nn, -1, np.int32)
    word_of_node[leaves] = np.arange(len(leaves), dtype=np.int32)
    return dict(k=np.int32(k), L=np.int32(L), desc=desc, children=ch, word_of_node=word_of_node,
                node_of_word=np.asarray(leaves, np.int32), parent=np.asarray(parent, np.int32))


def make_descriptors(n, desc_bytes=32, seed=11):
    """n random binary descriptors (one per landmark)."""
    return np.random.default_rng(seed).integers(0, 256, size=(n, desc_bytes), dtype=np.uint8)


def observe_descriptors(base, flip_bits, seed):
    """What another view of the same landmarks extracts: every descriptor with `flip_bits` random bits flipped."""
    rng = np.random.default_rng(seed)
    out = np.array(base, dtype=np.uint8, copy=True)
    nbits = out.shape[1] * 8
    for i in range(out.shape[0]):
        for b in rng.choice(nbits, size=flip_bits, replace=False):
            out[i, b >> 3] ^= np.uint8(1 << (b & 7))
    return out
