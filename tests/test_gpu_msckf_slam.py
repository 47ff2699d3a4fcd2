"""MSCKF-SLAM rows inside the visual update and the initialisation of the new persistent features on the GPU
 against the golden vectors / the oracle."""
import os

import numpy as np
import pytest

from helpers import GOLDEN_DIR, rel
from oracle import ref_np
from x_multi_agent_amd import synth

pytestmark = pytest.mark.gpu


def _stage(xk, g, N, M, n_ms):
    tr_off = g["trk_off"]
    obs = g["obs_xy"]
    tracks = [obs[tr_off[k]:tr_off[k + 1]] for k in range(len(tr_off) - 1)]
    eng = xk.Engine(N, M, 16)
    sc = dict(C_q_G=g["C_q_G"], G_p_C=g["G_p_C"], trk_off=np.concatenate([[0], np.cumsum([len(t) for t in tracks[:10]])]),
              obs_xy=np.concatenate(tracks[:10]), P=g["P_full"], n_poses_max=N, sigma_img=float(g["sigma_img"]),
              slam_feat=g["feat_full"][:9], slam_anchor_idxs=g["slam_anchor_idxs"], slam_z_last=g["slam_z_last"],
              slam_track_sizes=g["slam_track_sizes"])
    eng.stage(sc)
    eng.stage_msckf_slam(tracks[10:10 + n_ms])
    return eng, tracks


def test_update_with_msckf_slam_rows_and_feature_init(xk):
    g = np.load(os.path.join(GOLDEN_DIR, "msckf_slam_n8.npz"))
    N, M, sigma = int(g["n_poses_max"]), int(g["n_feat_max"]), float(g["sigma_img"])
    eng, tracks = _stage(xk, g, N, M, 3)
    r = eng.visual_update_staged(sigma)
    assert np.array_equal(r["inlier"], g["exp_inlier"]) and np.array_equal(r["inlier_slam"], g["exp_inlier_slam"])
    ms = eng.msckf_slam_results()
    assert np.array_equal(ms["inlier"], g["exp_inlier_ms"])
    assert rel(ms["gamma"], g["exp_gamma_ms"]) <= 1e-8
    assert rel(ms["features"], g["exp_features"]) <= 1e-9
    # what initMsckfSlamFeatures consumes is independent of the column-space basis
    H2i = np.linalg.inv(ms["H2"])
    assert rel(H2i @ ms["H1"], g["exp_G"]) <= 1e-9
    # (r1 vanishes at the Gauss-Newton optimum -- Hf^T res = 0 -- so this term is rounding noise: absolute bound)
    assert np.abs(H2i @ ms["r1"] - g["exp_g"]).max() <= 1e-10
    assert rel(H2i @ H2i.T, g["exp_HH"]) <= 1e-9
    assert np.all(ms["H2"][np.kron(np.eye(3), np.ones((3, 3))) == 0] == 0)      # block diagonal
    assert rel(r["correction"], g["exp_correction"]) <= 1e-8
    P_post = eng.download_P()
    assert rel(P_post, g["exp_P"]) <= 1e-9
    # feature initialisation on the resident posterior
    newf = eng.init_msckf_slam_features(3, r["correction"], sigma)
    assert rel(newf, g["exp_new_features"][9:18]) <= 1e-8
    P_init = eng.download_P()
    assert rel(P_init, g["exp_P_init"]) <= 1e-9
    assert np.array_equal(P_init[:33 + 6 * N - 18, :15], P_post[:33 + 6 * N - 18, :15])   # existing entries are carried over
    # the standard (uncorrelated) initialisation from the same posterior
    eng.upload_P(P_post)
    eng.init_standard_slam_features(3, 2, sigma, 0.4)
    assert rel(eng.download_P(), g["exp_P_std"]) <= 1e-12
    eng.close()


def test_rejected_msckf_slam_track_contributes_nothing(xk):
    """A gated-out MSCKF-SLAM track leaves zero rows (the reference pre-sizes the stack) but its
    column-space rows are still produced."""
    g = np.load(os.path.join(GOLDEN_DIR, "msckf_slam_n8.npz"))
    N, M, sigma = int(g["n_poses_max"]), int(g["n_feat_max"]), float(g["sigma_img"])
    eng, tracks = _stage(xk, g, N, M, 0)
    bad = tracks[10].copy()
    bad[::2] += 0.08                      # gross outlier observations
    eng.stage_msckf_slam([bad, tracks[11]])
    r = eng.visual_update_staged(sigma)
    ms = eng.msckf_slam_results()
    P1 = eng.download_P()
    slam = dict(track_sizes=g["slam_track_sizes"], z_last=g["slam_z_last"], feat=g["feat_full"], anchor_idxs=g["slam_anchor_idxs"])
    ref = ref_np.visual_update(tracks[:10], g["C_q_G"], g["G_p_C"], g["P_full"], N, sigma, slam=slam,
                               msckf_slam_tracks=[bad, tracks[11]])
    assert list(ms["inlier"]) == list(ref["msckf_slam"]["inlier"]) == [0, 1]
    assert rel(P1, ref["P"]) <= 1e-9 and rel(r["correction"], ref["correction"]) <= 1e-8
    eng.close()


def test_cpp_mirror_update_then_post_update_initialises_features(tmp_path):
    """Ekf::processUpdateMeasurement -> Updater::update -> constructUpdate (MSCKF + MSCKF-SLAM + SLAM rows) ->
    applyUpdate -> State::correct -> VioUpdater::postUpdate -> StateManager::initMsckfSlamFeatures, in C++."""
    import subprocess
    PKG = os.path.join(os.path.dirname(__file__), "..", "x_multi_agent_amd")
    exe = os.path.join(PKG, "xk_host_example")
    if not os.path.exists(exe):
        from x_multi_agent_amd import build
        build.build_host()
    g = np.load(os.path.join(GOLDEN_DIR, "msckf_slam_n8.npz"))
    N, M, sigma = int(g["n_poses_max"]), int(g["n_feat_max"]), float(g["sigma_img"])
    off, obs = g["trk_off"], g["obs_xy"]
    tracks = [obs[off[k]:off[k + 1]] for k in range(len(off) - 1)]
    K, K2, M_used = 10, 3, 3
    anchors = np.full(M, -1.0); anchors[:M_used] = g["slam_anchor_idxs"]
    zl = np.zeros((M, 2)); zl[:M_used] = g["slam_z_last"]
    tsz = np.ones(M); tsz[:M_used] = g["slam_track_sizes"]
    parts = [np.array([N, M, K, N, sigma], float), g["C_q_G"].ravel(), g["G_p_C"].ravel(),
             np.array([len(t) for t in tracks[:K]], float), np.concatenate(tracks[:K]).ravel(),
             g["feat_full"], anchors, zl.ravel(), tsz, np.asfortranarray(g["P_full"]).ravel(order="F"),
             np.array([K2, M_used], float), np.array([len(t) for t in tracks[K:K + K2]], float),
             np.concatenate(tracks[K:K + K2]).ravel()]
    fin, fout = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    np.concatenate(parts).astype("<f8").tofile(fin)
    env = dict(os.environ, LD_LIBRARY_PATH=PKG + ":/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", ""))
    r = subprocess.run([exe, fin, fout], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    out = np.fromfile(fout, dtype="<f8")
    n = 15 + 6 * N + 3 * M
    P = out[:n * n].reshape(n, n, order="F")
    f_arr = out[n * n + 7 * N:n * n + 7 * N + 3 * M]
    assert rel(P, g["exp_P_init"]) <= 1e-9
    assert rel(f_arr[9:18], g["exp_new_features"][9:18]) <= 1e-8
    # the features that already existed received their share of the correction (State::correct)
    assert rel(f_arr[:9], g["feat_full"][:9] + g["exp_correction"][15 + 6 * N:15 + 6 * N + 9]) <= 1e-9


def test_update_with_only_msckf_slam_tracks(xk):
    """K = 0, M = 0, K2 > 0: the MSCKF-SLAM rows alone are a non-empty h (vio_updater.cpp:311-321, 413-419), so the
    Kalman update runs (updater.cpp:106) and the new features can be initialised afterwards."""
    g = np.load(os.path.join(GOLDEN_DIR, "msckf_slam_n8.npz"))
    N, M, sigma = int(g["n_poses_max"]), int(g["n_feat_max"]), float(g["sigma_img"])
    off, obs = g["trk_off"], g["obs_xy"]
    tracks = [obs[off[k]:off[k + 1]] for k in range(len(off) - 1)]
    eng = xk.Engine(N, M, 16)
    sc = dict(C_q_G=g["C_q_G"], G_p_C=g["G_p_C"], trk_off=np.zeros(1, np.int32), obs_xy=np.zeros((0, 2)), P=g["P_full"],
              n_poses_max=N, sigma_img=sigma)
    eng.stage(sc)
    eng.stage_msckf_slam(tracks[10:13])
    r = eng.visual_update_staged(sigma)
    ref = ref_np.visual_update([], g["C_q_G"], g["G_p_C"], g["P_full"], N, sigma, msckf_slam_tracks=tracks[10:13])
    assert np.linalg.norm(ref["correction"]) > 0
    ms = eng.msckf_slam_results()
    assert list(ms["inlier"]) == list(ref["msckf_slam"]["inlier"])
    assert rel(eng.download_P(), ref["P"]) <= 1e-9 and rel(r["correction"], ref["correction"]) <= 1e-8
    eng.init_msckf_slam_features(3, r["correction"], sigma)      # needs the column-space rows of THIS update
    eng.close()


def test_update_without_any_rows_is_a_no_op(xk):
    """No usable track at all: the reference skips applyUpdate (h.size() == 0, updater.cpp:106); the reference-shaped
    call sequence build -> compress -> update must not fail on the empty tile stack either."""
    g = np.load(os.path.join(GOLDEN_DIR, "msckf_slam_n8.npz"))
    N, M, sigma = int(g["n_poses_max"]), int(g["n_feat_max"]), float(g["sigma_img"])
    eng = xk.Engine(N, M, 16)
    sc = dict(C_q_G=g["C_q_G"], G_p_C=g["G_p_C"], trk_off=np.zeros(1, np.int32), obs_xy=np.zeros((0, 2)), P=g["P_full"],
              n_poses_max=N, sigma_img=sigma)
    eng.stage(sc)
    eng.stage_msckf_slam([])
    r = eng.visual_update_staged(sigma)
    assert not r["correction"].any()
    assert np.array_equal(eng.download_P(), g["P_full"])
    eng.msckf_build(sigma)
    T, z = eng.qr_compress()
    assert not T.any() and not z.any()
    corr = eng.apply_update()
    assert not corr.any() and rel(eng.download_P(), g["P_full"]) <= 1e-15
    eng.close()


def test_slam_staging_rejects_stale_anchors(xk):
    g = np.load(os.path.join(GOLDEN_DIR, "msckf_slam_n8.npz"))
    N, M, sigma = int(g["n_poses_max"]), int(g["n_feat_max"]), float(g["sigma_img"])
    eng = xk.Engine(N, M, 16)
    base = dict(C_q_G=g["C_q_G"], G_p_C=g["G_p_C"], trk_off=np.zeros(1, np.int32), obs_xy=np.zeros((0, 2)), P=g["P_full"],
                n_poses_max=N, sigma_img=sigma, slam_feat=g["feat_full"][:9], slam_z_last=g["slam_z_last"],
                slam_track_sizes=g["slam_track_sizes"])
    for bad in (np.array([-1, 0, 1]), np.array([0, N, 1])):        # an unused StateManager slot; past the window
        with pytest.raises(xk.XkError) as e:
            eng.stage(dict(base, slam_anchor_idxs=bad.astype(np.int32)))
        assert e.value.status == 1
    with pytest.raises(xk.XkError) as e:
        eng.stage(dict(base, slam_anchor_idxs=g["slam_anchor_idxs"], slam_track_sizes=np.array([3, 0, 2], np.int32)))
    assert e.value.status == 1
    # anchors are rechecked against the window staged at build time
    short = dict(base, C_q_G=g["C_q_G"][:2], G_p_C=g["G_p_C"][:2], slam_anchor_idxs=np.array([0, 1, 5], np.int32))
    eng.stage(short)
    with pytest.raises(xk.XkError) as e:
        eng.msckf_build(sigma)
    assert e.value.status == 1
    eng.close()
