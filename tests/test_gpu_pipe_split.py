"""Round 5: the single launch with TWO first-level groups per XCD (XkPipeNarrow2: 152 tiles of 128 rows, 10 + 9 strips merged by 6
workgroups each, 16 roots at the last level -- csrc/xk_caqr_pipe.hip.h).  A first-level lane holds 11 rows instead of 24, so the
chain a panel waits for is shorter; it holds 19 456 accepted rows instead of 23 552, so the host takes it when the acceptance
ratio the LAST single launch reported says this update's rows will fit, and a launch that finds more gives up at once and is
redone with 184 tiles.  Checked here: the geometry itself against the C oracle over the shapes that stress the row plan
(forced with the lab option "pipe_split" = 3), the adaptive choice, and the overflow path."""
import numpy as np
import pytest

from helpers import rel
from x_multi_agent_amd import synth

pytestmark = pytest.mark.gpu

SHAPES = {
    "headline": lambda: synth.make_config(4),
    "cfg1": lambda: synth.make_config(1),
    "ragged": lambda: synth.make_scenario(30, 300, 0, seed=901, track_len=(2, 30)),
    "mostly_rejected": lambda: synth.make_scenario(20, 200, 0, seed=902, outlier_frac=0.7),
    "partial_window": lambda: synth.make_scenario(30, 120, 0, seed=903, n_poses=17),
    "just_enough_rows": lambda: synth.make_scenario(12, 26, 0, seed=904),
    "n31_full_width": lambda: synth.make_scenario(31, 250, 0, seed=915),                 # 187 columns: the last panel is short
    "narrow_with_slam": lambda: synth.make_scenario(20, 200, 10, seed=912),
    "large_prior": lambda: synth.make_scenario(30, 150, 0, seed=4107, prior_scale=100.0),
}


@pytest.mark.parametrize("name", sorted(SHAPES))
def test_two_first_level_groups_against_the_oracle(xk, oracle_c, name):
    sc = SHAPES[name]()
    ref = oracle_c.visual_update(sc)
    N, K = sc["n_poses_max"], len(sc["trk_off"]) - 1
    M = len(sc["slam_anchor_idxs"]) if "slam_anchor_idxs" in sc else 0
    eng = xk.LabEngine(N, M, max(K, 1))
    eng.set_option("pipe_split", 3)                       # the 152-tile geometry whatever the host expects (every shape here fits it)
    for kal in (1, 0):                                    # Kalman update inside the launch / behind it
        eng.set_option("pipe_kalman", kal)
        eng.stage(sc)
        r = eng.visual_update_staged(sc["sigma_img"])
        P = eng.download_P()
        eng.stage(sc)
        t = eng.bench_staged(sc["sigma_img"], 0, 1)
        st = eng.caqr_status()
        assert st["schedule"] == 2 and st["giveups"] == 0 and t["n_leaf"] == 152, (st, t["n_leaf"])
        assert np.array_equal(r["inlier"], ref["inlier"]) and np.array_equal(r["inlier_slam"], ref["inlier_slam"])
        assert rel(P, ref["P"]) <= 1e-8 and rel(r["correction"], ref["correction"]) <= 1e-6, (kal, rel(P, ref["P"]))
    eng.close()


def test_geometry_follows_the_acceptance_ratio(xk, oracle_c):
    """Release library, headline scenario (18 639 of 22 800 rows pass): the first update knows nothing and takes 184 tiles, reports
    its acceptance ratio; from the second on the 152-tile geometry -- same posterior to rounding, and bit-identical among
    themselves."""
    sc = synth.make_config(4)
    ref = oracle_c.visual_update(sc)
    N, K = sc["n_poses_max"], len(sc["trk_off"]) - 1
    eng = xk.Engine(N, 0, K)
    leaves, Ps = [], []
    for i in range(4):
        eng.stage(sc)
        r = eng.visual_update_staged(sc["sigma_img"])
        Ps.append(eng.download_P())
        assert np.array_equal(r["inlier"], ref["inlier"]) and rel(Ps[-1], ref["P"]) <= 1e-8
        eng.stage(sc)
        leaves.append(eng.bench_staged(sc["sigma_img"], 0, 1)["n_leaf"])
    # (bench_staged is an update too: the one after the first visual_update_staged already knows the ratio)
    assert leaves == [152, 152, 152, 152], leaves
    assert np.array_equal(Ps[1], Ps[2]) and np.array_equal(Ps[2], Ps[3])
    assert rel(Ps[0], Ps[1]) <= 1e-12
    assert eng.caqr_status()["giveups"] == 0
    # the nominal-rows scenario (22 173 rows pass) does not fit 19 456: 184 tiles
    sn = synth.make_config(4, err_scale=0.3, outlier_frac=0.0)
    refn = oracle_c.visual_update(sn)
    e2 = xk.Engine(N, 0, K)
    for i in range(3):
        e2.stage(sn)
        r = e2.visual_update_staged(sn["sigma_img"])
        assert rel(e2.download_P(), refn["P"]) <= 1e-8
        e2.stage(sn)
        assert e2.bench_staged(sn["sigma_img"], 0, 1)["n_leaf"] == 184
    assert e2.caqr_status()["giveups"] == 0
    e2.close()
    eng.close()


def test_more_rows_than_expected_costs_one_retry_with_184_tiles(xk, oracle_c):
    """The acceptance ratio comes from the LAST update.  A heavily gated update (half the tracks rejected) followed by one in which
    nearly every track passes: the 152-tile launch finds more rows than it holds, gives up at once (reason 9), the update is redone
    with 184 tiles -- correct result, one give-up on the record, fast path still armed, and the 152-tile geometry left alone for the
    next updates."""
    lo = synth.make_scenario(30, 400, 0, seed=7711, outlier_frac=0.5)
    hi = synth.make_config(4, err_scale=0.3, outlier_frac=0.0)
    ref_lo, ref_hi = oracle_c.visual_update(lo), oracle_c.visual_update(hi)
    assert 57 * int(ref_hi["inlier"].sum()) > 19456 > 57 * int(ref_lo["inlier"].sum()) * 1.1
    eng = xk.Engine(30, 0, 400)
    eng.stage(lo)
    r = eng.visual_update_staged(lo["sigma_img"])
    assert np.array_equal(r["inlier"], ref_lo["inlier"]) and rel(eng.download_P(), ref_lo["P"]) <= 1e-8
    eng.stage(hi)
    r = eng.visual_update_staged(hi["sigma_img"])                      # predicted to fit, does not
    assert np.array_equal(r["inlier"], ref_hi["inlier"]) and rel(eng.download_P(), ref_hi["P"]) <= 1e-8
    st = eng.caqr_status()
    assert st["giveups"] == 1 and st["last_reason"] == 9 and st["armed"] and st["schedule"] == 2, st
    for _ in range(3):                                                 # back-off: 184 tiles, no further give-up
        eng.stage(hi)
        assert eng.bench_staged(hi["sigma_img"], 0, 1)["n_leaf"] == 184
    assert eng.caqr_status()["giveups"] == 1
    eng.close()
