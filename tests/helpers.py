"""Shared helpers for the parity tests."""
import glob
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
PKG_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "x_multi_agent_amd")
LAB_DIR = os.path.join(PKG_DIR, "lab")          # lab/libxk.so: the -DXK_LAB build (test hooks, environment switches; include/xk_lab.h)


def with_lab(env):
    """Environment of a child process that needs a test hook or an environment switch: the C++ examples find lab/libxk.so
    before the release library (LD_LIBRARY_PATH precedes their RUNPATH), Python children load it through XK_LIB_PATH."""
    env = dict(env)
    env["LD_LIBRARY_PATH"] = LAB_DIR + ":" + env.get("LD_LIBRARY_PATH", "")
    env["XK_LIB_PATH"] = os.path.join(LAB_DIR, "libxk.so")
    return env
VISUAL_CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz"))
                      if not os.path.basename(p).startswith(("ci_", "manage_", "msckf_slam_", "vocab_", "multi_uav_", "propagator_", "iekf_")))

# north_star tolerance: <= 1e-6 relative Frobenius on P (BASELINE.json).  The two
# restatements and the GPU path agree far tighter than that; the tests assert the
# tighter bound so regressions are caught long before they reach the 1e-6 bar.
TOL_NORTH_STAR = 1e-6
TOL_TIGHT = 1e-9


def rel(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    nb = np.linalg.norm(b)
    if nb == 0:
        return float(np.linalg.norm(a))
    return float(np.linalg.norm(a - b) / nb)


def load_case(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    sc = {k: z[k] for k in z.files if not k.startswith("exp_")}
    sc["n_poses_max"] = int(sc["n_poses_max"])
    sc["sigma_img"] = float(sc["sigma_img"])
    sc["n"] = sc["P"].shape[0]
    exp = {k[4:]: z[k] for k in z.files if k.startswith("exp_")}
    return sc, exp


def check_visual(got, exp, tol=TOL_TIGHT):
    """got: dict(P, correction, inlier, gamma[, inlier_slam, gamma_slam])"""
    assert np.array_equal(np.asarray(got["inlier"]).astype(int), exp["inlier"].astype(int)), "MSCKF inlier mask"
    fin = np.isfinite(exp["gamma"])
    assert rel(np.asarray(got["gamma"])[fin], exp["gamma"][fin]) <= 1e-8, "gamma"
    if "inlier_slam" in exp:
        assert np.array_equal(np.asarray(got["inlier_slam"]).astype(int), exp["inlier_slam"].astype(int))
        assert rel(got["gamma_slam"], exp["gamma_slam"]) <= 1e-8
    rp, rc = rel(got["P"], exp["P"]), rel(got["correction"], exp["correction"])
    assert rp <= tol, f"posterior covariance rel error {rp}"
    assert rc <= max(tol, 1e-8), f"correction rel error {rc}"
    return rp, rc


def load_iekf_case(name):
    """-> (scenario dict, state dict, {iekf_iter: expected dict})"""
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    sc = {k: z[k] for k in z.files if not k.startswith(("exp", "st_", "iters"))}
    sc["n_poses_max"] = int(sc["n_poses_max"])
    sc["sigma_img"] = float(sc["sigma_img"])
    st = {k[3:]: z[k] for k in z.files if k.startswith("st_")}
    exp = {}
    for it in z["iters"]:
        pre = f"exp{int(it)}_"
        e = {k[len(pre):]: z[k] for k in z.files if k.startswith(pre) and not k.startswith(pre + "st_")}
        e["state"] = {k[len(pre) + 3:]: z[k] for k in z.files if k.startswith(pre + "st_")}
        exp[int(it)] = e
    return sc, st, exp
