"""Updater::update with iekf_iter > 1 (src/x/ekf/updater.cpp:99-110) through the C++ mirror on the GPU.

Every pass re-linearises the rows at the state as corrected so far (window lists and SLAM features re-staged), gates and
compresses them against the PRIOR, and only the last pass touches the covariance: xk_apply_update(cov_update = 0) followed
by a fresh build + compression on the same resident prior -- a sequence no other test runs.  Checked against the NumPy
restatement's fixtures (tests/golden/iekf_*.npz, make_golden.iekf_case) and, live, against the C oracle's loop."""
import os
import subprocess

import numpy as np
import pytest

from helpers import load_iekf_case, rel, with_lab
from oracle import c_oracle

pytestmark = pytest.mark.gpu
PKG = os.path.join(os.path.dirname(__file__), "..", "x_multi_agent_amd")


def run_example(tmp_path, sc, st, iekf_iter, resident, extra_env=None):
    exe = os.path.join(PKG, "xk_host_example")
    N = sc["n_poses_max"]
    npz = len(sc["G_p_C"])
    K = len(sc["trk_off"]) - 1
    M = len(sc["slam_anchor_idxs"]) if "slam_anchor_idxs" in sc else 0
    n = sc["P"].shape[0]
    L = np.diff(sc["trk_off"]).astype(float)
    parts = [np.array([N, M, K, npz, sc["sigma_img"]], float), st["q_array"], st["p_array"], L, sc["obs_xy"].ravel()]
    if M:
        parts += [st["f_array"][:3 * M], sc["slam_anchor_idxs"].astype(float), sc["slam_z_last"].ravel(),
                  sc["slam_track_sizes"].astype(float)]
    parts.append(np.asfortranarray(sc["P"]).ravel(order="F"))
    fin, fout, fcore = str(tmp_path / "in.bin"), str(tmp_path / "out.bin"), str(tmp_path / "core.bin")
    np.concatenate(parts).astype("<f8").tofile(fin)
    np.concatenate([st["p"], st["v"], st["q"], st["b_w"], st["b_a"]]).astype("<f8").tofile(fcore)
    env = dict(os.environ, LD_LIBRARY_PATH=PKG + ":/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", ""))
    if extra_env:                       # environment switches are read by the lab build of the library only
        env = with_lab(env)
        env.update(extra_env)
    r = subprocess.run([exe, fin, fout, str(iekf_iter), str(int(resident)), fcore], capture_output=True, text=True, env=env,
                       timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    out = np.fromfile(fout, dtype="<f8")
    at = n * n
    got = dict(P=out[:at].reshape(n, n, order="F"), p_array=out[at:at + 3 * N], q_array=out[at + 3 * N:at + 7 * N],
               f_array=out[at + 7 * N:at + 7 * N + 3 * M])
    at += 7 * N + 3 * M
    got["inlier"] = out[at:at + K].astype(int)
    dyn = out[at + K:at + K + 16]
    got.update(p=dyn[0:3], v=dyn[3:6], q=dyn[6:10], b_w=dyn[10:13], b_a=dyn[13:16])
    return got, r.stdout


def check(got, e, M, tol=1e-9):
    assert np.array_equal(got["inlier"], e["inlier"].astype(int))
    assert rel(got["P"], e["P"]) <= tol, rel(got["P"], e["P"])
    for k in ("p", "v", "q", "b_w", "b_a", "p_array", "q_array"):
        assert rel(got[k], e["state"][k]) <= tol, k
    if M:
        assert rel(got["f_array"], e["state"]["f_array"][:3 * M]) <= tol


@pytest.mark.parametrize("resident", [0, 1])
@pytest.mark.parametrize("name", ["iekf_n8_k30_m6", "iekf_n10_k50"])
def test_iekf_loop_through_the_cpp_mirror(tmp_path, name, resident):
    """iekf_iter = 2 and 3, MSCKF + SLAM rows (multi-launch compression, 402 rows) and MSCKF rows (single launch, 850
    rows), covariance owned by the State or resident on the device."""
    sc, st, exp = load_iekf_case(name)
    M = len(sc["slam_anchor_idxs"]) if "slam_anchor_idxs" in sc else 0
    for it, e in exp.items():
        got, _ = run_example(tmp_path, sc, st, it, resident)
        check(got, e, M)
        # the C oracle's own loop, live, on the same inputs
        ref = c_oracle.visual_update_iekf(sc, st, it)
        assert rel(got["P"], ref["P"]) <= 1e-9 and np.array_equal(got["inlier"], ref["inlier"])
        assert rel(got["b_a"], ref["state"]["b_a"]) <= 1e-9


def test_iekf_loop_with_the_queued_single_launch_giving_up(tmp_path):
    """Resident order, three passes, and the queued single launch of the FIRST pass made to give up (abort word raised
    before the launch): xk_apply_update(cov_update = 0) redoes rows + compression with the multi-launch schedule against the
    untouched prior, the later passes find the fast path disarmed -- same posterior."""
    sc, st, exp = load_iekf_case("iekf_n10_k50")
    got, _ = run_example(tmp_path, sc, st, 3, 1, {"XK_CAQR_RESIDENT_POISON": "1"})
    check(got, exp[3], 0)


def test_one_iteration_is_the_plain_update(tmp_path):
    sc, st, exp = load_iekf_case("iekf_n10_k50")
    got, _ = run_example(tmp_path, sc, st, 1, 1)
    ref = c_oracle.visual_update(sc)
    assert rel(got["P"], ref["P"]) <= 1e-9 and np.array_equal(got["inlier"], ref["inlier"])
    assert rel(got["P"], exp[2]["P"]) > 1e-6    # (and the two-pass posterior is a different matrix)
