"""The register-resident single-launch QR compression (csrc/xk_caqr_pipe.hip.h, the default whenever the stack is
MSCKF rows only and fits 184 fat tiles of 128 rows) against the multi-launch CAQR schedule on the same inputs -- same R up
to rounding, hence the same posterior -- over shapes that stress its bookkeeping: ragged tracks (fat tiles cut tracks at
arbitrary rows), many rejected tracks (zero rows inside fat tiles), few rows (most fat tiles short or empty), the
headline size (tiles full), a partially filled window; and against the C oracle."""
import os

import numpy as np
import pytest

from helpers import rel
from x_multi_agent_amd import synth

pytestmark = pytest.mark.gpu


def _run(xk, sc, resident, ms_tracks=None):
    try:
        N, K = sc["n_poses_max"], len(sc["trk_off"]) - 1
        M = len(sc["slam_anchor_idxs"]) if "slam_anchor_idxs" in sc else 0
        eng = xk.Engine(N, M, max(K, 1))
        eng.set_option("caqr_resident", int(bool(resident)))     # (an operational switch of the release library)
        eng.stage(sc)
        if ms_tracks is not None:
            eng.stage_msckf_slam(ms_tracks)
        r = eng.visual_update_staged(sc["sigma_img"])
        P = eng.download_P()
        eng.stage(sc)
        if ms_tracks is not None:
            eng.stage_msckf_slam(ms_tracks)
        t = eng.bench_staged(sc["sigma_img"], 0, 1)
        eng.close()
        return r, P, t
    finally:
        pass


CASES = {
    "headline": lambda: synth.make_config(4),
    "headline_nominal_rows": lambda: synth.make_config(4, err_scale=0.3, outlier_frac=0.0),    # bench.py's value_at_nominal_rows: ~97 % of the tracks pass
    "cfg1": lambda: synth.make_config(1),
    "ragged": lambda: synth.make_scenario(30, 300, 0, seed=901, track_len=(2, 30)),
    "mostly_rejected": lambda: synth.make_scenario(20, 200, 0, seed=902, outlier_frac=0.7),
    "partial_window": lambda: synth.make_scenario(30, 120, 0, seed=903, n_poses=17),
    "just_enough_rows": lambda: synth.make_scenario(12, 26, 0, seed=904),       # 546 rows: 3 rows per fat tile
    "stress_prior": lambda: synth.make_scenario(24, 350, 0, seed=905, prior_kind="stress", prior_scale=0.01),
    # SLAM rows and systems wider than 192 columns: the wide geometry (2 lanes per column, 152 tiles of 80 rows)
    "cfg2": lambda: synth.make_config(2),
    "slam_heavy": lambda: synth.make_scenario(20, 150, 40, seed=908),
    "slam_ragged": lambda: synth.make_scenario(30, 120, 25, seed=909, track_len=(2, 30)),
    "wide_no_slam_rows": lambda: synth.make_scenario(33, 150, 0, seed=910),          # 199 columns: wide because of the window alone
    "widest": lambda: synth.make_scenario(33, 120, 61, seed=914),                     # 382 columns: two short of the wide geometry's 384
    # SLAM rows in a system that still fits 192 columns: the narrow geometry with all three row kinds in its row map
    "narrow_with_slam": lambda: synth.make_scenario(20, 200, 10, seed=912),
}
NLEAF = {"cfg2": 152, "slam_heavy": 152, "slam_ragged": 152, "wide_no_slam_rows": 152, "widest": 152}


@pytest.mark.parametrize("name", sorted(CASES))
def test_resident_equals_multi_launch(xk, oracle_c, name):
    sc = CASES[name]()
    ra, Pa, ta = _run(xk, sc, True)
    rb, Pb, tb = _run(xk, sc, False)
    # (narrow systems: 184 tiles, or 152 with two first-level groups per XCD once the handle knows the acceptance ratio -- round 5)
    assert ta["n_levels"] == 1 and ta["n_leaf"] in ((NLEAF[name],) if name in NLEAF else (184, 152)), "the single-launch path did not run"
    assert tb["n_levels"] > 1
    assert np.array_equal(ra["inlier"], rb["inlier"])
    assert rel(Pa, Pb) <= 1e-11 and rel(ra["correction"], rb["correction"]) <= 1e-9
    ref = oracle_c.visual_update(sc)
    assert np.array_equal(ra["inlier"], ref["inlier"])
    assert rel(Pa, ref["P"]) <= 1e-8


def test_msckf_slam_rows_go_through_the_single_launch_too(xk, oracle_c):
    """All three row kinds of vio_updater.cpp:406-422 in one stack -- MSCKF tracks, MSCKF-SLAM tracks (the landmark becomes a
    persistent feature this frame), SLAM rows -- single launch against the multi-launch schedule."""
    sc = synth.make_scenario(16, 60, 6, seed=913)
    tr = synth.tracks_as_list(sc)
    sc2 = dict(sc)
    sc2["trk_off"] = sc["trk_off"][:51].copy()
    sc2["obs_xy"] = sc["obs_xy"][:sc["trk_off"][50]].copy()
    ra, Pa, ta = _run(xk, sc2, True, ms_tracks=tr[50:56])
    rb, Pb, tb = _run(xk, sc2, False, ms_tracks=tr[50:56])
    assert ta["n_levels"] == 1 and ta["n_leaf"] in (184, 152) and tb["n_levels"] > 1
    assert np.array_equal(ra["inlier"], rb["inlier"])
    assert rel(Pa, Pb) <= 1e-11 and rel(ra["correction"], rb["correction"]) <= 1e-9


def test_resident_path_steps_aside_when_it_does_not_apply(xk):
    """Windows of more than 33 poses (128-row slots): the multi-launch schedule runs (since round 6 for the first panels only: n_levels counts
    its launches and the tail's).  Few rows are no reason any more (round 6): 130 rows in 49 columns take the single launch like any stack that
    needs compressing."""
    for sc, single in ((synth.make_config(3), False), (synth.make_scenario(40, 420, 0, seed=906), False), (synth.make_scenario(8, 10, 0, seed=907), True)):
        N, K = sc["n_poses_max"], len(sc["trk_off"]) - 1
        M = len(sc["slam_anchor_idxs"]) if "slam_anchor_idxs" in sc else 0
        eng = xk.Engine(N, M, K)
        eng.stage(sc)
        t = eng.bench_staged(sc["sigma_img"], 0, 1)
        assert (t["n_levels"] == 1) == single, t["n_levels"]
        eng.close()


def test_resident_launch_that_gives_up_is_redone_by_the_multi_launch_schedule(xk, oracle_c):
    """Every spin of the single launch is bounded and looks at an abort word; a launch that gives up (workgroups not
    co-resident, uneven XCD placement) must cost one retry, not a wrong answer.  XK_CAQR_RESIDENT_POISON raises the abort
    word before the launch: the update has to come back correct, through the multi-launch schedule, and the handle has to
    keep working (with that schedule) afterwards."""
    sc = synth.make_config(4)
    ref = oracle_c.visual_update(sc)
    N, K = sc["n_poses_max"], len(sc["trk_off"]) - 1
    eng = xk.LabEngine(N, 0, K)                      # (test hook: the lab build of the library)
    eng.set_option("caqr_poison", 1)
    eng.stage(sc)
    r = eng.visual_update_staged(sc["sigma_img"])
    eng.set_option("caqr_poison", 0)
    assert np.array_equal(r["inlier"], ref["inlier"])
    assert rel(eng.download_P(), ref["P"]) <= 1e-8
    assert rel(r["correction"], ref["correction"]) <= 1e-6
    st = eng.caqr_status()
    assert st["giveups"] == 1 and st["last_reason"] == 7 and not st["armed"] and st["schedule"] == 0, st
    eng.stage(sc)                                   # the same handle again: multi-launch for the next XK_CAQR_REARM updates
    t = eng.bench_staged(sc["sigma_img"], 0, 1)
    assert t["n_leaf"] != 184
    eng.stage(sc)
    r2 = eng.visual_update_staged(sc["sigma_img"])
    assert rel(eng.download_P(), ref["P"]) <= 1e-8 and rel(r2["correction"], ref["correction"]) <= 1e-6
    eng.close()


def test_fast_path_is_rearmed_after_clean_updates(xk, oracle_c):
    """A launch that gave up must not cost the fast path for the life of the handle: after XK_CAQR_REARM clean multi-launch
    updates the single launch is tried again (with its sync words cleared -- the launch that gave up left them mid-count) and,
    the GPU being free again, stays; a second give-up doubles the distance."""
    sc = synth.make_config(4)
    ref = oracle_c.visual_update(sc)
    N, K = sc["n_poses_max"], len(sc["trk_off"]) - 1
    eng = xk.LabEngine(N, 0, K)
    eng.set_option("caqr_rearm", 3)
    sched = []
    for i in range(8):
        eng.set_option("caqr_poison", int(i in (0, 6)))
        eng.stage(sc)
        r = eng.visual_update_staged(sc["sigma_img"])
        eng.set_option("caqr_poison", 0)
        assert np.array_equal(r["inlier"], ref["inlier"])
        assert rel(eng.download_P(), ref["P"]) <= 1e-8 and rel(r["correction"], ref["correction"]) <= 1e-6, i
        sched.append(eng.caqr_status()["schedule"])
    # update 0 gives up -> multi-launch; 1, 2, 3 multi-launch; 4 re-armed: single launch again; 5 likewise; 6 gives up again
    assert sched[:4] == [0, 0, 0, 0] and sched[4] == 2 and sched[5] == 2 and sched[6] == 0, sched
    st = eng.caqr_status()
    assert st["giveups"] == 2 and not st["armed"]
    eng.close()


def test_tracks_staged_in_place_give_the_same_update(xk, oracle_c):
    """xk_stage_tracks_begin / _end (the CSR lists are built IN the engine's pinned staging memory) against xk_stage_tracks on
    the same scenario, and the misuse the header rules out: a last offset that disagrees with n_obs, _end without _begin."""
    import ctypes as C
    sc = synth.make_scenario(12, 60, 0, seed=911, track_len=(2, 12))
    N, K = sc["n_poses_max"], len(sc["trk_off"]) - 1
    eng = xk.Engine(N, 0, K)
    eng.stage(sc)
    eng.visual_update_staged(sc["sigma_img"])                      # (the first update of a handle does not know the acceptance ratio yet and
    eng.stage(sc)                                                  #  may take another geometry than the following ones: compare the 2nd and 3rd)
    ra = eng.visual_update_staged(sc["sigma_img"])
    Pa = eng.download_P()
    eng.stage(sc)                                                  # window, SLAM (none) and prior again, then the tracks in place
    off = np.asarray(sc["trk_off"], np.int32)
    obs = np.ascontiguousarray(sc["obs_xy"], np.float64).ravel()
    po, px = C.POINTER(C.c_int)(), C.POINTER(C.c_double)()
    assert eng.L.xk_stage_tracks_begin(eng.h, C.c_int(K), C.c_int(int(off[-1])), C.byref(po), C.byref(px)) == 0
    C.memmove(po, off.ctypes.data, off.nbytes)
    C.memmove(px, obs.ctypes.data, obs.nbytes)
    assert eng.L.xk_stage_tracks_end(eng.h) == 0
    rb = eng.visual_update_staged(sc["sigma_img"])
    assert np.array_equal(ra["inlier"], rb["inlier"]) and np.array_equal(ra["correction"], rb["correction"])
    assert np.array_equal(Pa, eng.download_P())
    ref = oracle_c.visual_update(sc)
    assert rel(Pa, ref["P"]) <= 1e-8
    assert eng.L.xk_stage_tracks_end(eng.h) != 0                   # nothing begun
    assert eng.L.xk_stage_tracks_begin(eng.h, C.c_int(K), C.c_int(int(off[-1]) + 1), C.byref(po), C.byref(px)) == 0
    C.memmove(po, off.ctypes.data, off.nbytes)
    assert eng.L.xk_stage_tracks_end(eng.h) != 0                   # trk_off[K] != n_obs
    eng.close()
