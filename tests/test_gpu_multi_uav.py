"""MULTI_UAV form of Updater::update (updater.cpp:84-97) through the C++ mirror: constructUpdate builds the stacked rows
AND the MSCKF-MSCKF CI lists from the prior, applyCI runs per entry (each overwriting the covariance, Q6), applyUpdate
runs on the post-CI covariance.  Checked against tests/golden/multi_uav_n8_k20.npz (oracle/ref_np.multi_uav_update)
with the covariance owned by the State and with the covariance resident on the device."""
import os
import subprocess
import sys

import numpy as np
import pytest

from helpers import GOLDEN_DIR, rel, with_lab
from oracle import ref_np
from x_multi_agent_amd import synth

sys.path.insert(0, GOLDEN_DIR)
import make_golden_multi_uav  # noqa: E402  (the fixture's generator doubles as the headline-size case builder)

pytestmark = pytest.mark.gpu
PKG = os.path.join(os.path.dirname(__file__), "..", "x_multi_agent_amd")


def _run(tmp_path, g, resident, n_short=0):
    exe = os.path.join(PKG, "xk_multi_uav_example")
    if not os.path.exists(exe):
        from x_multi_agent_amd import build
        build.build_host()
    N, A = int(g["n_poses_max"]), int(g["n_agents"])
    off, obs = g["own_trk_off"], g["own_obs"]
    K = len(off) - 1 - n_short                      # (the last n_short tracks of the list are the short ones)
    mt, ma = g["match_track"], g["match_agent"]
    parts = [np.array([N, -K if n_short else K, A, len(mt), float(g["sigma_img"]), float(g["ci_msckf_w"])] + ([n_short] if n_short else []))]
    for a in range(A):
        parts += [g[f"a{a}_C_q_G"].ravel(), g[f"a{a}_G_p_C"].ravel(), np.asfortranarray(g[f"a{a}_P"]).ravel(order="F")]
    parts += [np.diff(off).astype(float), obs.ravel()]
    for i in range(len(mt)):
        r = g[f"recv{i}"]
        parts += [np.array([mt[i], ma[i], len(r)], float), r.ravel()]
    fin, fout = str(tmp_path / "in.bin"), str(tmp_path / f"out{int(resident)}.bin")
    np.concatenate(parts).astype("<f8").tofile(fin)
    env = dict(os.environ, LD_LIBRARY_PATH=PKG + ":/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", ""))
    if "XK_CAQR_RESIDENT_POISON" in env:            # (a test hook: the lab build of the library)
        env = with_lab(env)
    r = subprocess.run([exe, fin, fout, str(int(resident))], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    out = np.fromfile(fout, dtype="<f8")
    n = 15 + 6 * N
    at = n * n
    return dict(P=out[:at].reshape(n, n, order="F"), p_array=out[at:at + 3 * N], q_array=out[at + 3 * N:at + 7 * N],
                core=out[at + 7 * N:at + 7 * N + 16], n_ci=int(out[at + 7 * N + 16]), inlier=out[at + 7 * N + 17:at + 7 * N + 17 + K])


@pytest.mark.parametrize("resident", [False, True])
def test_ci_then_update_in_the_reference_order(tmp_path, resident):
    g = np.load(os.path.join(GOLDEN_DIR, "multi_uav_n8_k20.npz"))
    got = _run(tmp_path, g, resident)
    assert got["n_ci"] == int(g["exp_n_ci"]) >= 2                  # two entries: the second applyCI discards the first one's P
    assert np.array_equal(got["inlier"].astype(int), g["exp_inlier"].astype(int))
    assert rel(got["P"], g["exp_P"]) <= 1e-8
    # not what an update without the CI entries gives (the lists really were consumed)
    assert rel(got["P"], g["exp_P_no_ci"]) > 1e-3
    assert rel(got["p_array"], g["exp_p_array"]) <= 1e-9 and rel(got["q_array"], g["exp_q_array"]) <= 1e-9
    assert rel(got["core"], g["exp_core"]) <= 1e-7


@pytest.mark.parametrize("resident", [False, True])
def test_ci_then_update_at_the_headline_size(tmp_path, resident):
    """The same composed update at N = 30, K = 400 with three CI entries (one track matched by two agents, a joint gate and
    an own gate that fail), in resident mode the order that matters: buildMsckfCiLists downloads the prior, the build +
    compression is QUEUED, xk_apply_ci_resident rewrites the covariance three times, xk_apply_update consumes the [T_H | z]
    that was linearised at the prior.  Expected values: oracle/ref_np.multi_uav_update on the same inputs."""
    g = make_golden_multi_uav.case(30, 400, 3, 0x5EED5301, 0.2, table=[(0, [1, 2]), (1, [1]), (2, [1]), (3, [2]), (7, [2]), (11, [1])])
    got = _run(tmp_path, g, resident)
    assert got["n_ci"] == int(g["exp_n_ci"]) >= 3, g["_summary"]
    assert np.array_equal(got["inlier"].astype(int), g["exp_inlier"].astype(int))
    assert rel(got["P"], g["exp_P"]) <= 1e-8
    assert rel(got["P"], g["exp_P_no_ci"]) > 1e-4
    assert rel(got["p_array"], g["exp_p_array"]) <= 1e-9 and rel(got["q_array"], g["exp_q_array"]) <= 1e-9
    assert rel(got["core"], g["exp_core"]) <= 1e-7


@pytest.mark.parametrize("resident", [False, True])
def test_headline_ci_then_update_survives_a_single_launch_that_gives_up(tmp_path, resident):
    """the queued single-launch CAQR gives up (XK_CAQR_RESIDENT_POISON) while CI entries are waiting.  The
    retry has to happen BEFORE the first applyCI replaces the covariance the rows were linearised at (settle_async)."""
    g = make_golden_multi_uav.case(30, 400, 3, 0x5EED5301, 0.2, table=[(0, [1, 2]), (1, [1]), (2, [1]), (3, [2]), (7, [2])])
    os.environ["XK_CAQR_RESIDENT_POISON"] = "1"
    try:
        got = _run(tmp_path, g, resident)
    finally:
        os.environ.pop("XK_CAQR_RESIDENT_POISON", None)
    assert got["n_ci"] == int(g["exp_n_ci"]) and rel(got["P"], g["exp_P"]) <= 1e-8
    assert rel(got["core"], g["exp_core"]) <= 1e-7


def _short_case(N=8, K=20, n_short=5, seed=0x5EED5401, w=0.2):
    """Regular tracks 0..K-n_short-1, then n_short SHORT tracks (the last 3..5 observations of the remaining ones): the
    MULTI_UAV build takes only the CI entries of the short tracks (updater.cpp:52-66), applies them, and runs the regular
    update on what they left -- state AND covariance."""
    scs = [synth.make_scenario(N, K, 0, seed=seed)] + [synth.make_scenario(N, K, 0, seed=seed + a, agent_offset=0.04 * a) for a in (1, 2)]
    for a in (1, 2):
        scs[a] = synth.make_scenario(N, K, 0, seed=seed + a, agent_offset=0.04 * a, landmarks=scs[0]["landmarks_true"])
    tr = [synth.tracks_as_list(s_) for s_ in scs]
    Kr = K - n_short
    rng = np.random.default_rng(seed)
    Ls = [3, 4, 5, 3, 4][:n_short]
    own = [t.copy() for t in tr[0][:Kr]] + [tr[0][Kr + i][-Ls[i]:].copy() for i in range(n_short)]
    # matches: regular tracks 0 (agents 1, 2) and 4 (agent 1); short tracks 0 (agent 1), 1 (agent 2, joint gate fails), 3 (agents 1, 2)
    table = [(0, [1, 2]), (4, [1]), (Kr + 0, [1]), (Kr + 1, [2]), (Kr + 3, [1, 2])]
    recv = {}
    for t, ags in table:
        for a in ags:
            r = tr[a][t].copy()
            if t >= Kr:
                r = r[-(Ls[t - Kr] + 1):]                                  # the sender tracked it one frame longer
            recv[(t, a)] = r
    recv[(Kr + 1, 2)] = recv[(Kr + 1, 2)] + 0.05 * rng.standard_normal(recv[(Kr + 1, 2)].shape)
    mk = lambda t, a: dict(obs=recv[(t, a)], q_list=scs[a]["C_q_G"], p_list=scs[a]["G_p_C"], P=scs[a]["P"], n_poses_max=N)
    q0, p0, P0, sig = scs[0]["C_q_G"], scs[0]["G_p_C"], scs[0]["P"], scs[0]["sigma_img"]
    st = dict(p=np.zeros(3), v=np.zeros(3), q=np.array([0, 0, 0, 1.0]), b_w=np.zeros(3), b_a=np.zeros(3),
              p_array=p0.ravel().copy(), q_array=q0.ravel().copy(), f_array=np.zeros(0))
    # short-track block: every entry from the state and prior at its start, applied one after the other
    entries = []
    for t, ags in table:
        if t < Kr:
            continue
        o = ref_np.msckf_ci_track(own[t], q0, p0, P0, N, sig, [mk(t, a) for a in ags], w)
        if o["ci"] is not None:
            entries.append(o["ci"])
    P1 = P0
    for c in entries:
        P1, corr = ref_np.apply_ci(c["P_j"], c["H"], c["res"], c["S"])
        st = ref_np.state_correct(st, corr)
    n_short_ci = len(entries)
    # regular update from what the short block left
    q1, p1 = st["q_array"].reshape(N, 4), st["p_array"].reshape(N, 3)
    out = ref_np.multi_uav_update(own[:Kr], {t: [mk(t, a) for a in ags] for t, ags in table if t < Kr}, q1, p1, P1, N, sig, w)
    for c in out["corrections"]:
        st = ref_np.state_correct(st, c)
    d = dict(n_poses_max=N, sigma_img=sig, ci_msckf_w=w, n_agents=3, match_track=np.array([t for t, ags in table for a in ags]),
             match_agent=np.array([a for t, ags in table for a in ags]), exp_P=out["P"], exp_n_ci_regular=out["n_ci"],
             exp_n_ci_short=n_short_ci, exp_p_array=st["p_array"], exp_q_array=st["q_array"],
             exp_core=np.concatenate([st["p"], st["v"], st["q"], st["b_w"], st["b_a"]]), exp_P_no_short=None)
    d["exp_P_no_short"] = ref_np.multi_uav_update(own[:Kr], {t: [mk(t, a) for a in ags] for t, ags in table if t < Kr}, q0, p0, P0, N, sig, w)["P"]
    for a, s_ in enumerate(scs):
        d[f"a{a}_C_q_G"], d[f"a{a}_G_p_C"], d[f"a{a}_P"] = s_["C_q_G"], s_["G_p_C"], s_["P"]
    d["own_trk_off"] = np.concatenate([[0], np.cumsum([len(t) for t in own])]).astype(np.int32)
    d["own_obs"] = np.concatenate(own)
    for i, (t, a) in enumerate((t, a) for t, ags in table for a in ags):
        d[f"recv{i}"] = recv[(t, a)]
    return d, n_short


@pytest.mark.parametrize("resident", [False, True])
def test_short_track_ci_entries_before_the_regular_update(tmp_path, resident):
    g, n_short = _short_case()
    assert g["exp_n_ci_short"] >= 2 and g["exp_n_ci_regular"] >= 1
    got = _run(tmp_path, g, resident, n_short=n_short)
    assert got["n_ci"] == g["exp_n_ci_regular"]                    # (ciEntriesOfLastUpdate: the regular block's)
    assert rel(got["P"], g["exp_P"]) <= 1e-8
    assert rel(got["P"], g["exp_P_no_short"]) > 1e-4               # the short block's entries really were applied
    assert rel(got["p_array"], g["exp_p_array"]) <= 1e-9 and rel(got["q_array"], g["exp_q_array"]) <= 1e-9
    assert rel(got["core"], g["exp_core"]) <= 1e-7


def _collab_case():
    z = np.load(os.path.join(GOLDEN_DIR, "ci_two_agents.npz"))
    a = {k[2:]: z[k] for k in z.files if k.startswith("a_")}
    b = {k[2:]: z[k] for k in z.files if k.startswith("b_")}
    b_bad = dict(b)
    b_bad["slam_feat"] = b["slam_feat"].copy()
    b_bad["slam_feat"][3 * 1 + 2] *= 3.0                            # feature 1 of that snapshot sits somewhere else: gated out
    N, M = int(z["n_poses_max"]), len(a["slam_anchor_idxs"])
    matches = [(0, 0, 0), (1, 1, 1), (2, 2, 0), (3, 3, 0)]         # (current feature, received feature, other snapshot)
    others = [b, b_bad]
    sl, w = float(z["sigma_landmark"]), float(z["ci_slam_w"])
    entries, inl = [], []
    for cur, rcv, o in matches:
        ob = others[o]
        r = ref_np.multi_slam_match(a["C_q_G"], a["G_p_C"], a["slam_feat"], int(a["slam_anchor_idxs"][cur]), cur, a["P"], N,
                                    ob["C_q_G"], ob["G_p_C"], ob["slam_feat"], int(ob["slam_anchor_idxs"][rcv]), rcv, ob["P"], N, sl, w)
        inl.append(bool(r["inlier"]))
        if r["inlier"]:
            entries.append(r)
    st = dict(p=np.zeros(3), v=np.zeros(3), q=np.array([0, 0, 0, 1.0]), b_w=np.zeros(3), b_a=np.zeros(3),
              p_array=a["G_p_C"].ravel().copy(), q_array=a["C_q_G"].ravel().copy(), f_array=a["slam_feat"].copy())
    P, first = a["P"], None
    for e in entries:                                               # updater.cpp:27-35: every P_j from the SAME prior (Q6)
        P, corr = ref_np.apply_ci(e["P_j"], e["H"], e["res"], e["S"])
        first = P if first is None else first
        st = ref_np.state_correct(st, corr)
    return dict(N=N, M=M, a=a, others=others, matches=matches, sl=sl, w=w, inl=inl, exp_P=P, exp_P_first=first, st=st)


@pytest.mark.parametrize("resident", [False, True])
def test_collaborative_update_through_the_cpp_route(tmp_path, resident):
    """Ekf::processOthersMeasurement -> Updater::collaborativeUpdate -> VioUpdater::constructSlamCIUpdate -> applyCI
    (ekf.cpp:143-176, updater.cpp:22-36, vio_updater.cpp:81-123, multi_slam_update.cpp:61-246) with four SlamMatches, one of
    them gated out, against oracle/ref_np: all entries from the same prior, the covariance after the loop is the LAST
    entry's (Q6), the state has taken every entry's correction."""
    c = _collab_case()
    assert c["inl"] == [True, False, True, True]
    N, M, a = c["N"], c["M"], c["a"]
    parts = [np.array([N, M, len(c["others"]), len(c["matches"]), c["sl"], c["w"]])]
    for s_ in [a] + c["others"]:
        parts += [s_["C_q_G"].ravel(), s_["G_p_C"].ravel(), s_["slam_feat"].ravel(), s_["slam_anchor_idxs"].astype(float),
                  np.asfortranarray(s_["P"]).ravel(order="F")]
    parts += [np.array(m, float) for m in c["matches"]]
    fin, fout = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    np.concatenate(parts).astype("<f8").tofile(fin)
    exe = os.path.join(PKG, "xk_collaborative_example")
    env = dict(os.environ, LD_LIBRARY_PATH=PKG + ":/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", ""))
    r = subprocess.run([exe, fin, fout, str(int(resident))], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    out = np.fromfile(fout, dtype="<f8")
    n = 15 + 6 * N + 3 * M
    P = out[:n * n].reshape(n, n, order="F")
    at = n * n
    p_array, q_array, f_array = out[at:at + 3 * N], out[at + 3 * N:at + 7 * N], out[at + 7 * N:at + 7 * N + 3 * M]
    core = out[at + 7 * N + 3 * M:at + 7 * N + 3 * M + 16]
    assert rel(P, c["exp_P"]) <= 1e-9
    assert rel(P, c["exp_P_first"]) > 1e-6                          # not the first entry's posterior: each applyCI overwrote
    st = c["st"]
    assert rel(p_array, st["p_array"]) <= 1e-10 and rel(q_array, st["q_array"]) <= 1e-10 and rel(f_array, st["f_array"]) <= 1e-10
    assert rel(core, np.concatenate([st["p"], st["v"], st["q"], st["b_w"], st["b_a"]])) <= 1e-8
