"""The host-side C++ mirror of the reference API (host/: x::Ekf -> x::Updater::update ->
x::VioUpdater::constructUpdate -> applyUpdate -> State::correct) driven end to end on the GPU and
compared with the golden vectors."""
import os
import subprocess

import numpy as np
import pytest

from helpers import load_case, rel
from oracle import ref_np

pytestmark = pytest.mark.gpu
PKG = os.path.join(os.path.dirname(__file__), "..", "x_multi_agent_amd")


@pytest.mark.parametrize("name", ["cfg1_n10_k50", "slam_n8_k30_m6", "partial_window_n10_p7_k20_m3"])
def test_filter_loop_through_mirrored_api(tmp_path, name):
    exe = os.path.join(PKG, "xk_host_example")
    if not os.path.exists(exe):
        from x_multi_agent_amd import build
        build.build_host()
    sc, exp = load_case(name)
    N = sc["n_poses_max"]
    npz = len(sc["G_p_C"])
    K = len(sc["trk_off"]) - 1
    M = len(sc["slam_anchor_idxs"]) if "slam_anchor_idxs" in sc else 0
    q = np.zeros((N, 4)); q[:, 3] = 1.0; q[:npz] = sc["C_q_G"]
    p = np.zeros((N, 3)); p[:npz] = sc["G_p_C"]
    L = np.diff(sc["trk_off"]).astype(float)
    parts = [np.array([N, M, K, npz, sc["sigma_img"]], float), q.ravel(), p.ravel(), L, sc["obs_xy"].ravel()]
    if M:
        parts += [sc["slam_feat"], sc["slam_anchor_idxs"].astype(float), sc["slam_z_last"].ravel(),
                  sc["slam_track_sizes"].astype(float)]
    parts.append(np.asfortranarray(sc["P"]).ravel(order="F"))
    fin, fout = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    np.concatenate(parts).astype("<f8").tofile(fin)
    env = dict(os.environ, LD_LIBRARY_PATH=PKG + ":/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", ""))
    r = subprocess.run([exe, fin, fout], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    out = np.fromfile(fout, dtype="<f8")
    n = sc["P"].shape[0]
    P = out[:n * n].reshape(n, n, order="F")
    at = n * n
    p_arr, q_arr, f_arr = out[at:at + 3 * N], out[at + 3 * N:at + 7 * N], out[at + 7 * N:at + 7 * N + 3 * M]
    inl = out[at + 7 * N + 3 * M:].astype(int)
    assert np.array_equal(inl, exp["inlier"])
    assert rel(P, exp["P"]) <= 1e-9
    # State::correct with the golden correction (state.cpp:197-249)
    st = dict(p=np.zeros(3), v=np.zeros(3), q=np.array([0, 0, 0, 1.0]), b_w=np.zeros(3), b_a=np.zeros(3),
              p_array=p.ravel(), q_array=q.ravel(), f_array=sc["slam_feat"] if M else np.zeros(0))
    ref = ref_np.state_correct(st, exp["correction"])
    assert rel(p_arr, ref["p_array"]) <= 1e-9 and rel(q_arr, ref["q_array"]) <= 1e-9
    if M:
        assert rel(f_arr, ref["f_array"]) <= 1e-9


def test_config2_through_the_mirrored_api(tmp_path, oracle_c):
    """BASELINE config 2 (window 30, 200 MSCKF tracks, 50 SLAM features, n = 345) through x::Ekf -> x::Updater::update -> x::VioUpdater: the
    compression the mirror queues is the SPLIT one (round 6) -- posterior, gate results and corrected state against the C oracle."""
    from x_multi_agent_amd import synth
    exe = os.path.join(PKG, "xk_host_example")
    sc = synth.make_config(2)
    exp = oracle_c.visual_update(sc)
    N, K, M = synth.CONFIGS[2]
    parts = [np.array([N, M, K, N, sc["sigma_img"]], float), sc["C_q_G"].ravel(), sc["G_p_C"].ravel(), np.diff(sc["trk_off"]).astype(float),
             sc["obs_xy"].ravel(), sc["slam_feat"], sc["slam_anchor_idxs"].astype(float), sc["slam_z_last"].ravel(),
             sc["slam_track_sizes"].astype(float), np.asfortranarray(sc["P"]).ravel(order="F")]
    fin, fout = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    np.concatenate(parts).astype("<f8").tofile(fin)
    env = dict(os.environ, LD_LIBRARY_PATH=PKG + ":/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", ""))
    r = subprocess.run([exe, fin, fout], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    out = np.fromfile(fout, dtype="<f8")
    n = sc["P"].shape[0]
    P = out[:n * n].reshape(n, n, order="F")
    at = n * n
    p_arr, q_arr, f_arr = out[at:at + 3 * N], out[at + 3 * N:at + 7 * N], out[at + 7 * N:at + 7 * N + 3 * M]
    inl = out[at + 7 * N + 3 * M:].astype(int)
    assert np.array_equal(inl[:K], exp["inlier"])
    assert rel(P, exp["P"]) <= 1e-9, rel(P, exp["P"])
    st = dict(p=np.zeros(3), v=np.zeros(3), q=np.array([0, 0, 0, 1.0]), b_w=np.zeros(3), b_a=np.zeros(3),
              p_array=sc["G_p_C"].ravel(), q_array=sc["C_q_G"].ravel(), f_array=sc["slam_feat"])
    ref = ref_np.state_correct(st, exp["correction"])
    assert rel(p_arr, ref["p_array"]) <= 1e-9 and rel(q_arr, ref["q_array"]) <= 1e-9 and rel(f_arr, ref["f_array"]) <= 1e-9
