"""Tall systems (windows of 34..64 poses, 128-row slots; BASELINE config 3): the multi-launch schedule for the first panels and ONE
or TWO launches of xk_caqr_pipe<XkPipeTail4 / XkPipeTail> for the last <= 192 / 96 columns, every row in registers (round 6;
VioUpdater::applyQRDecomposition, vio_updater.cpp:487-512) -- against the C oracle, against the multi-launch schedule to the last panel,
over the plans (two launches of 192 columns, one launch of 192, one of 96), a launch that gives up, repeated updates bit for bit."""
import numpy as np
import pytest

from helpers import rel
from x_multi_agent_amd import synth

pytestmark = pytest.mark.gpu


def _update(eng, sc):
    eng.stage(sc)
    r = eng.visual_update_staged(sc["sigma_img"])
    return r, eng.download_P()


SHAPES = {
    # (scenario, launches of the tail the default plan takes: 2 = the stack does not fit one 192-column launch)
    "cfg3": (lambda: synth.make_config(3), 2),
    "n34_fits_one_launch": (lambda: synth.make_scenario(34, 300, 0, seed=3401), 1),
    "n40_ragged": (lambda: synth.make_scenario(40, 500, 0, seed=3402, track_len=(2, 40)), 1),
    "n64_half_rejected": (lambda: synth.make_scenario(64, 420, 0, seed=3403, outlier_frac=0.5), 2),
    "n48_partial_window": (lambda: synth.make_scenario(48, 260, 0, seed=3404, n_poses=37), 1),
    "n36_few_tracks": (lambda: synth.make_scenario(36, 45, 0, seed=3405), 1),
    "n40_with_slam_features": (lambda: synth.make_scenario(40, 180, 8, seed=3406), 1),      # SLAM rows packed 128 to a slot behind the tracks
}


@pytest.mark.parametrize("name", sorted(SHAPES))
def test_tail_launch_against_the_oracle_and_the_multi_launch_schedule(xk, oracle_c, name):
    make, _ = SHAPES[name]
    sc = make()
    ref = oracle_c.visual_update(sc)
    N, K = sc["n_poses_max"], len(sc["trk_off"]) - 1
    M = len(sc["slam_anchor_idxs"]) if "slam_anchor_idxs" in sc else 0
    out = {}
    for tail in (1, 2, 0):
        eng = xk.Engine(N, M, K)
        eng.set_option("caqr_tail", tail)
        r, P = _update(eng, sc)
        st = eng.caqr_status()
        assert st["giveups"] == 0, st
        assert (st["schedule"] == 3) == (tail != 0), (tail, st)
        assert np.array_equal(r["inlier"], ref["inlier"]), tail
        assert rel(P, ref["P"]) <= 1e-8 and rel(r["correction"], ref["correction"]) <= 1e-6, (tail, rel(P, ref["P"]))
        out[tail] = (r, P)
        eng.close()
    for tail in (1, 2):
        assert rel(out[tail][1], out[0][1]) <= 1e-11 and rel(out[tail][0]["correction"], out[0][0]["correction"]) <= 1e-9, tail


def test_compressed_system_keeps_the_gram_matrix(xk):
    """[T_H | z] of the schedule with the tail: T^T T = H^T H of the stacked rows (rows that passed the gates), to rounding."""
    sc = synth.make_config(3)
    N, K = sc["n_poses_max"], len(sc["trk_off"]) - 1
    G = {}
    for tail in (1, 0):
        eng = xk.Engine(N, 0, K)
        eng.set_option("caqr_tail", tail)
        eng.stage(sc)
        eng.msckf_build(sc["sigma_img"])
        T, z = eng.qr_compress()
        assert (eng.caqr_status()["schedule"] == 3) == bool(tail)
        Ta = np.column_stack([T[:, 15:], z])
        G[tail] = Ta.T @ Ta
        assert np.allclose(np.tril(T[:6 * N, 15:15 + 6 * N], -1), 0.0)
        eng.close()
    # (T^T T and T^T z; NOT z^T z: H has the filter's unobservable directions in its null space, the rows of T that belong to them are
    #  ~ 0 and their entries of z -- projections of the residual on directions that depend on the factorisation's tree -- are free)
    na = 6 * N
    assert rel(G[1][:, :na], G[0][:, :na]) <= 5e-13, rel(G[1][:, :na], G[0][:, :na])


def test_tail_that_gives_up_is_redone_and_comes_back(xk, oracle_c):
    """Lab hook "caqr_poison": the abort word is up before the tail launch -- every workgroup leaves at its first spin.  The update has to
    come back right (rows rebuilt, the multi-launch schedule to the last panel), the tail stays off for "caqr_rearm" clean updates and
    is then taken again."""
    sc = synth.make_scenario(40, 240, 0, seed=3410)
    ref = oracle_c.visual_update(sc)
    N, K = sc["n_poses_max"], len(sc["trk_off"]) - 1
    eng = xk.LabEngine(N, 0, K)
    eng.set_option("caqr_rearm", 2)
    sched = []
    for i in range(6):
        eng.set_option("caqr_poison", int(i == 1))
        r, P = _update(eng, sc)
        eng.set_option("caqr_poison", 0)
        assert np.array_equal(r["inlier"], ref["inlier"]), i
        assert rel(P, ref["P"]) <= 1e-8 and rel(r["correction"], ref["correction"]) <= 1e-6, i
        sched.append(eng.caqr_status()["schedule"])
    st = eng.caqr_status()
    # 0 tail; 1 gives up -> redone without it; 2, 3 without; 4 re-armed
    assert sched[0] == 3 and sched[1] == 0 and sched[2] == 0 and sched[3] == 0 and sched[4] == 3 and sched[5] == 3, sched
    assert st["giveups"] == 1 and st["last_reason"] == 7, st
    eng.close()


def test_repeated_updates_with_the_tail_are_bit_identical(xk):
    sc = synth.make_config(3)
    N, K = sc["n_poses_max"], len(sc["trk_off"]) - 1
    eng = xk.Engine(N, 0, K)
    r0, P0 = _update(eng, sc)
    assert eng.caqr_status()["schedule"] == 3
    for _ in range(6):
        r, P = _update(eng, sc)
        assert np.array_equal(P, P0) and np.array_equal(r["correction"], r0["correction"])
    eng.close()


def test_more_accepted_rows_than_the_tail_holds(xk, oracle_c):
    """BASELINE config 3's shape with (nearly) every track passing the gates: 77 600 nominal rows are within a quarter of what two
    launches hold (64 768), so the tail is queued -- and finds ~75 000 rows: it gives up at once (reason 9), the update is redone by the
    multi-launch schedule to the last panel, and the tail stays off for the next updates instead of giving up every time."""
    sc = synth.make_config(3, err_scale=0.3, outlier_frac=0.0)
    ref = oracle_c.visual_update(sc)
    assert int(ref["inlier"].sum()) * 97 > 64768
    N, K = sc["n_poses_max"], len(sc["trk_off"]) - 1
    eng = xk.Engine(N, 0, K)
    r, P = _update(eng, sc)
    st = eng.caqr_status()
    assert st["giveups"] == 1 and st["last_reason"] == 9 and st["schedule"] == 0, st
    assert np.array_equal(r["inlier"], ref["inlier"]) and rel(P, ref["P"]) <= 1e-8
    r2, P2 = _update(eng, sc)
    st = eng.caqr_status()
    assert st["giveups"] == 1 and st["schedule"] == 0, st
    assert np.array_equal(P2, P)
    eng.close()
