"""StateManager::manage restatement (oracle/ref_np.py, state_manager.cpp:31-149): golden regression and the
structural properties the reference code implies."""
import os

import numpy as np
import pytest

from helpers import GOLDEN_DIR
from oracle import ref_np
from x_multi_agent_amd import synth


def _run(seq):
    sm, st = seq["init"]["sm"], {k: v for k, v in seq["init"].items() if k != "sm"}
    out = []
    for step in seq["steps"]:
        st.update(p=step["p"], q=step["q"], q_ic=step["q_ic"], p_ic=step["p_ic"])
        sm, st = ref_np.state_manage(sm, st, step["del"])
        out.append((dict(sm, anchor_idxs=list(sm["anchor_idxs"])), {k: np.array(v, copy=True) for k, v in st.items()}))
    return out


@pytest.mark.parametrize("name", list(synth.MANAGE_SEQUENCES))
def test_golden_sequences(name):
    g = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    for i, (sm, st) in enumerate(_run(synth.make_manage_sequence(**synth.MANAGE_SEQUENCES[name]))):
        assert np.allclose(st["cov"], g[f"s{i}_cov"], rtol=0, atol=1e-15 * max(1.0, np.abs(g[f"s{i}_cov"]).max()))
        assert np.array_equal(st["q_array"], g[f"s{i}_q_array"]) and np.array_equal(st["p_array"], g[f"s{i}_p_array"])
        assert np.allclose(st["f_array"], g[f"s{i}_f_array"], rtol=1e-14, atol=0)
        assert list(g[f"s{i}_sm"]) == [sm["n_poses"], sm["n_features"], int(sm["filled_before"])] + sm["anchor_idxs"]


def test_window_fill_then_slide_structure():
    seq = synth.make_manage_sequence(**synth.MANAGE_SEQUENCES["manage_empty_n4_m0"])
    N = seq["N"]
    res = _run(seq)
    P0 = seq["init"]["cov"]
    # first call on a never-filled window: the Jacobian of the untouched slots is ZERO (state_manager.cpp:276-279),
    # so only core + pose 0 survive
    sm, st = res[0]
    live = list(range(15)) + list(range(15, 18)) + list(range(15 + 3 * N, 18 + 3 * N))
    dead = [i for i in range(P0.shape[0]) if i not in live]
    assert np.abs(st["cov"][dead]).max() == 0.0 and np.abs(st["cov"][:, dead]).max() == 0.0
    assert np.array_equal(st["cov"][:15, :15], P0[:15, :15])
    # new camera position error = imu position error - R [p_ic]x dtheta: its covariance with the core block
    step = seq["steps"][0]
    Jp = np.zeros((3, 15)); Jp[:, 0:3] = np.eye(3); Jp[:, 6:9] = -ref_np.quat_to_rot(step["q"]) @ ref_np.skew(step["p_ic"])
    assert np.allclose(st["cov"][15:18, :15], Jp @ P0[:15, :15], rtol=1e-13, atol=1e-18)
    assert np.allclose(st["cov"][15:18, 15:18], Jp @ P0[:15, :15] @ Jp.T, rtol=1e-13, atol=1e-18)
    # window occupancy and the 'filled before' latch
    assert [r[0]["n_poses"] for r in res] == [1, 2, 3, 4, 4, 4, 4]
    assert [r[0]["filled_before"] for r in res] == [False, False, False, True, True, True, True]
    # a slide moves pose i+1 into slot i: after step 4 slot 0 holds what slot 1 held after step 3
    q3, q4 = res[3][1]["q_array"], res[4][1]["q_array"]
    assert np.array_equal(q4[:4 * (N - 1)], q3[4:])
    # every posterior stays symmetric positive semi-definite
    for _, st in res:
        c = st["cov"]
        assert np.abs(c - c.T).max() <= 1e-15 * np.abs(c).max()
        assert np.linalg.eigvalsh(0.5 * (c + c.T)).min() >= -1e-12 * np.abs(c).max()


def test_feature_removal_is_a_shift_and_reanchoring_preserves_the_landmark():
    seq = synth.make_manage_sequence(**synth.MANAGE_SEQUENCES["manage_feats_n5_m4"])
    N, M = seq["N"], seq["M"]
    sm0, st0 = seq["init"]["sm"], {k: v for k, v in seq["init"].items() if k != "sm"}

    def landmark(st, sm, j):
        a = sm["anchor_idxs"][j]
        al, be, rho = st["f_array"][3 * j:3 * j + 3]
        return ref_np.quat_to_rot(st["q_array"][4 * a:4 * a + 4]) @ np.array([al, be, 1.0]) / rho + st["p_array"][3 * a:3 * a + 3]

    res = _run(seq)
    # step 0, 1 fill the window (3 -> 5 poses); step 1 deletes feature 1: features behind it move up
    (sm_a, st_a), (sm_b, st_b) = res[0], res[1]
    assert sm_a["n_features"] == 4 and sm_b["n_features"] == 3
    assert np.array_equal(st_b["f_array"][3:9], st_a["f_array"][6:12]) and np.all(st_b["f_array"][9:] == 0)
    f0 = 15 + 6 * N
    assert np.all(st_b["cov"][-3:] == 0) and np.all(st_b["cov"][:, -3:] == 0)
    # step 2 is the first slide: features anchored in pose 0 are re-expressed in the newest pose, same 3-D point
    (sm_c, st_c) = res[2]
    st_before = dict(st_b)
    for j in range(sm_b["n_features"]):
        if sm_b["anchor_idxs"][j] == 0:
            assert sm_c["anchor_idxs"][j] == N - 2          # re-anchored to N-1, then shifted by the slide
            # the anchor pose it now refers to is the pose that sat in slot N-1 before the slide
            a = N - 1
            R = ref_np.quat_to_rot(st_before["q_array"][4 * a:4 * a + 4])
            al, be, rho = st_c["f_array"][3 * j:3 * j + 3]
            new_pt = R @ np.array([al, be, 1.0]) / rho + st_before["p_array"][3 * a:3 * a + 3]
            assert np.allclose(new_pt, landmark(st_before, sm_b, j), rtol=1e-12, atol=1e-12)
        else:
            assert sm_c["anchor_idxs"][j] == sm_b["anchor_idxs"][j] - 1


def test_propagation_blocks_equal_the_full_congruence():
    rng = np.random.default_rng(3)
    n = 15 + 6 * 5
    A = rng.normal(size=(n, n))
    P = A @ A.T
    F = np.eye(15) + 0.05 * rng.normal(size=(15, 15))
    B = rng.normal(size=(15, 15))
    Q = B @ B.T
    J = np.eye(n)
    J[:15, :15] = F
    full = J @ P @ J.T
    full[:15, :15] += Q
    got = ref_np.propagate_covariance_matrices(P, F, Q)
    assert np.allclose(got, full, rtol=1e-13, atol=1e-13 * np.abs(full).max())
    assert np.array_equal(got[15:, 15:], P[15:, 15:])
