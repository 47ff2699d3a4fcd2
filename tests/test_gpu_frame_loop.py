"""A whole filter frame through the C++ mirror -- IMU propagation (state on the host, covariance transitions applied on
the device), StateManager::manage, the visual update, State::correct -- with the covariance owned by the State
(reference semantics) and resident on the device (steps composed / one launch per step), against the NumPy oracle."""
import os
import subprocess

import numpy as np
import pytest

from helpers import rel, with_lab
from oracle import ref_np
from x_multi_agent_amd import synth

pytestmark = pytest.mark.gpu
PKG = os.path.join(os.path.dirname(__file__), "..", "x_multi_agent_amd")


def run_frame_loop(tmp_path, sc, frames, imu_per_frame, mode, extra_env=None):
    exe = os.path.join(PKG, "xk_frame_loop_example")
    if not os.path.exists(exe):
        from x_multi_agent_amd import build
        build.build_host()
    N = sc["n_poses_max"]
    off = sc["trk_off"]
    K = len(off) - 1
    parts = [np.array([N, K, frames, imu_per_frame, mode, sc["sigma_img"]], float), sc["C_q_G"].ravel(), sc["G_p_C"].ravel(),
             np.diff(off).astype(float), sc["obs_xy"].ravel(), np.asfortranarray(sc["P"]).ravel(order="F")]
    fin, fout = str(tmp_path / "in.bin"), str(tmp_path / f"out{mode}.bin")
    np.concatenate(parts).astype("<f8").tofile(fin)
    env = dict(os.environ, LD_LIBRARY_PATH=PKG + ":/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", ""))
    if extra_env:                       # environment switches are read by the lab build of the library only
        env = with_lab(env)
        env.update(extra_env)
    r = subprocess.run([exe, fin, fout], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    out = np.fromfile(fout, dtype="<f8")
    n = 15 + 6 * N
    at = n * n
    return dict(P=out[:at].reshape(n, n, order="F"), p_array=out[at:at + 3 * N], q_array=out[at + 3 * N:at + 7 * N],
                core=out[at + 7 * N:at + 7 * N + 16], ms=out[at + 7 * N + 16:], log=r.stdout)


def oracle_frame(sc, imu_per_frame, dt=0.005):
    N = sc["n_poses_max"]
    q, p = sc["C_q_G"], sc["G_p_C"]
    g = np.array([0, 0, -9.81])
    src = [max(i - 1, 0) for i in range(N)]
    st = dict(time=10.0, p=p[N - 1].copy(), v=np.zeros(3), q=q[N - 1].copy(), b_w=np.zeros(3), b_a=np.zeros(3),
              w_m=np.zeros(3), a_m=ref_np.quat_to_rot(q[N - 1]).T @ (-g))
    P = sc["P"].copy()
    noise = (0.0013, 0.00013, 0.0083, 0.00083)             # ImuNoise defaults of the mirror
    for i in range(1, imu_per_frame + 1):
        nxt = dict(time=10.0 + i * dt, w_m=st["w_m"].copy(), a_m=st["a_m"].copy())
        ref_np.propagate_state(st, nxt, g)
        e_w, e_a = nxt["w_m"] - nxt["b_w"], nxt["a_m"] - nxt["b_a"]
        F = ref_np.discrete_state_transition(dt, e_w, e_a, nxt["q"])
        Q = ref_np.process_noise_model(dt, e_w, e_a, nxt["q"], *noise)
        P = ref_np.propagate_covariance_matrices(P, F, Q)
        st = nxt
    sm = dict(n_poses=N, n_features=0, n_poses_max=N, n_features_max=0, anchor_idxs=[], filled_before=True)
    full = dict(p=st["p"], q=st["q"], q_ic=np.array([0, 0, 0, 1.0]), p_ic=np.zeros(3), q_array=q[src].ravel().copy(),
                p_array=p[src].ravel().copy(), f_array=np.zeros(0), cov=P)
    sm, full = ref_np.state_manage(sm, full, ())
    out = ref_np.visual_update(synth.tracks_as_list(sc), full["q_array"].reshape(N, 4), full["p_array"].reshape(N, 3), full["cov"], N,
                               sc["sigma_img"])
    s2 = dict(p=st["p"], v=st["v"], q=st["q"], b_w=st["b_w"], b_a=st["b_a"], p_array=full["p_array"], q_array=full["q_array"],
              f_array=np.zeros(0))
    s2 = ref_np.state_correct(s2, out["correction"])
    return dict(P=out["P"], p_array=s2["p_array"], q_array=s2["q_array"], core=np.concatenate([s2["p"], s2["v"], s2["q"], s2["b_w"], s2["b_a"]]),
                inliers=int(out["msckf"]["inlier"].sum()))


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_frame_matches_oracle(tmp_path, mode):
    sc = synth.make_scenario(8, 24, 0, seed=0x5EED6001)
    ref = oracle_frame(sc, 3)
    assert ref["inliers"] >= 12
    got = run_frame_loop(tmp_path, sc, 2, 3, mode)                    # two frames: the replay set-up really restores the prior
    assert f"inliers={ref['inliers']} " in got["log"]
    assert rel(got["P"], ref["P"]) <= 1e-9
    assert rel(got["p_array"], ref["p_array"]) <= 1e-10 and rel(got["q_array"], ref["q_array"]) <= 1e-10
    assert rel(got["core"], ref["core"]) <= 1e-8


def test_resident_and_state_owned_covariance_agree_at_the_headline_size(tmp_path):
    sc = synth.make_config(4)
    a = run_frame_loop(tmp_path, sc, 2, 7, 0)
    b = run_frame_loop(tmp_path, sc, 2, 7, 1)
    assert rel(b["P"], a["P"]) <= 1e-10 and rel(b["core"], a["core"]) <= 1e-9


def test_frame_with_the_runtime_wait_instead_of_the_completion_marker(tmp_path):
    """XK_SPIN_DONE=0: xk_apply_update waits for the stream instead of polling the marker its last workgroup writes; and
    with the marker, ten frames in a row (the marker's sequence number and the window lists carried by manage()'s congruence)."""
    sc = synth.make_scenario(8, 24, 0, seed=0x5EED6001)
    ref = oracle_frame(sc, 3)
    a = run_frame_loop(tmp_path, sc, 2, 3, 1, {"XK_SPIN_DONE": "0"})
    b = run_frame_loop(tmp_path, sc, 10, 3, 1)
    for got in (a, b):
        assert f"inliers={ref['inliers']} " in got["log"]
        assert rel(got["P"], ref["P"]) <= 1e-9 and rel(got["core"], ref["core"]) <= 1e-8
    assert np.array_equal(a["P"], b["P"])
