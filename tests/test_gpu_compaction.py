"""Gated-out tracks cost the single-launch compression nothing: the launch compacts the stack itself (xk_pipe_rowplan) -- the
reference appends inliers only (src/x/vio/msckf_update.cpp:463-479).  Shapes whose NOMINAL row count is past what the tiles
hold but whose accepted rows fit take the single launch; shapes whose accepted rows do not fit are found out by the launch
itself, served by the multi-launch schedule, and not tried again at that size."""
import numpy as np
import pytest

from helpers import rel
from x_multi_agent_amd import synth

pytestmark = pytest.mark.gpu


def _run(xk, sc, resident=1, slam_split=1):
    N, K = sc["n_poses_max"], len(sc["trk_off"]) - 1
    M = len(sc["slam_anchor_idxs"]) if "slam_anchor_idxs" in sc else 0
    eng = xk.Engine(N, M, K)
    eng.set_option("caqr_resident", resident)
    eng.set_option("slam_split", slam_split)
    eng.stage(sc)
    r = eng.visual_update_staged(sc["sigma_img"])
    P = eng.download_P()
    st = eng.caqr_status()
    return eng, r, P, st


def test_more_tracks_than_the_nominal_capacity(xk, oracle_c):
    """N = 30, K = 480: 27 360 nominal rows against 23 552 register rows -- round 3's capacity cliff (K = 413) -- of which ~22 500
    pass the gate: one launch, same posterior as the multi-launch schedule and the C oracle."""
    sc = synth.make_scenario(30, 480, 0, seed=5101)
    ref = oracle_c.visual_update(sc)
    assert 2 * 30 - 3 and int(ref["inlier"].sum()) * 57 <= 23552 < 480 * 57
    eng, r, P, st = _run(xk, sc)
    assert st["schedule"] == 2 and st["giveups"] == 0, st
    assert np.array_equal(r["inlier"], ref["inlier"]) and rel(P, ref["P"]) <= 1e-8
    eng.close()
    eng2, r2, P2, st2 = _run(xk, sc, resident=0)
    assert st2["schedule"] == 0 and rel(P, P2) <= 1e-11
    eng2.close()


def test_accepted_rows_that_do_not_fit_are_found_out_by_the_launch(xk, oracle_c):
    """BASELINE config 2's shape with 260 tracks: 14 920 nominal rows (within a quarter of the wide geometry's 12 160), ~12 600
    accepted: the launch gives up at once (reason 9), the multi-launch schedule redoes the update -- same posterior -- and that
    size is not tried again; the fast path stays armed for stacks that fit."""
    sc = synth.make_scenario(30, 260, 50, seed=5102)
    ref = oracle_c.visual_update(sc)
    # (round 6: with the split compression -- the default where SLAM features keep the update out of the launch -- this stack is a narrow
    #  one, 14 820 nominal rows in 181 columns, and fits: served by the single launch.  The capacity cliff of the WIDE geometry is what this
    #  test is about: "slam_split" 0 compresses the whole stack)
    eng, r, P, st = _run(xk, sc)
    assert st["schedule"] == 2 and st["giveups"] == 0 and rel(P, ref["P"]) <= 1e-8, st
    eng.close()
    eng, r, P, st = _run(xk, sc, slam_split=0)
    assert st["schedule"] == 0 and st["giveups"] == 1 and st["last_reason"] == 9 and st["armed"], st
    assert np.array_equal(r["inlier"], ref["inlier"]) and rel(P, ref["P"]) <= 1e-8
    eng.stage(sc)
    eng.visual_update_staged(sc["sigma_img"])
    st = eng.caqr_status()
    assert st["giveups"] == 1 and st["schedule"] == 0, st            # not tried again
    small = synth.make_scenario(30, 150, 50, seed=5104)              # (same handle, a stack that fits: single launch)
    eng.stage(small)
    r3 = eng.visual_update_staged(small["sigma_img"])
    assert eng.caqr_status()["schedule"] == 2
    assert rel(eng.download_P(), oracle_c.visual_update(small)["P"]) <= 1e-8
    eng.close()


@pytest.mark.parametrize("frac", [0.0, 0.5, 0.9])
def test_heavily_gated_stacks(xk, oracle_c, frac):
    """Outlier observations on a growing share of the tracks: the compacted stack shrinks, the lighter tile step takes over,
    tiles past the last accepted row hold nothing."""
    sc = synth.make_scenario(30, 300, 0, seed=5105)
    rng = np.random.default_rng(7)
    K = 300
    bad = rng.permutation(K)[:int(frac * K)]
    off = sc["trk_off"]
    obs = sc["obs_xy"].copy()
    for k in bad:                                                    # a gross error in the middle of the track: the chi-square gate throws it out
        obs[(off[k] + off[k + 1]) // 2] += 0.05
    sc = dict(sc, obs_xy=obs)
    ref = oracle_c.visual_update(sc)
    assert ref["inlier"].sum() <= K - len(bad) + 2
    eng, r, P, st = _run(xk, sc)
    if int(ref["inlier"].sum()) * 57 >= 512:
        assert st["schedule"] == 2, st
    assert np.array_equal(r["inlier"], ref["inlier"]) and rel(P, ref["P"]) <= 1e-8
    eng.close()
