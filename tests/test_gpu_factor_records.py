"""Round 5: H0 never visits HBM on the single-launch path.  The per-feature kernel leaves the FACTOR RECORD of
a track -- the three reflectors of its Hf, r' = Q^T res, and per column the two non-zero Jacobian entries with w = T^T V^T J[:, c]
(csrc/xk_feature.hip.h: XkFeatArgs::Hc, a tenth of the 64-row tile) -- and xk_caqr_pipe's tile workgroups form the rows
A^T [J | res] (msckf_update.cpp:423-432, 468-479) from the records they stage in LDS.  Checked here: the posterior against the C
oracle on the shapes that stress the row plan (ragged tracks: many records per tile; rejected tracks between accepted ones;
SLAM rows, which stay tiles, next to records), equality to rounding with the tiles-in-HBM path (lab option "caqr_hlite" = 0), and
the multi-launch schedule behind a launch that gave up -- its first tile pass (xk_caqr_tile, panel 0) forms its rows from the same
records; with 128-row slots (windows of 34..64 poses, BASELINE config 3) that is the only schedule and the records the only form."""
import numpy as np
import pytest

from helpers import rel
from x_multi_agent_amd import synth

pytestmark = pytest.mark.gpu

SHAPES = {
    "headline": lambda: synth.make_config(4),
    "cfg1": lambda: synth.make_config(1),
    "ragged": lambda: synth.make_scenario(30, 300, 0, seed=901, track_len=(2, 30)),
    "very_short_tracks": lambda: synth.make_scenario(30, 400, 0, seed=931, track_len=(2, 4)),     # 1..5 rows per track: dozens of records per tile
    "mostly_rejected": lambda: synth.make_scenario(20, 200, 0, seed=902, outlier_frac=0.7),
    "partial_window": lambda: synth.make_scenario(30, 120, 0, seed=903, n_poses=17),
    "just_enough_rows": lambda: synth.make_scenario(12, 26, 0, seed=904),
    "n31_full_width": lambda: synth.make_scenario(31, 250, 0, seed=915),
    "narrow_with_slam": lambda: synth.make_scenario(20, 200, 10, seed=912),
    "n33_longest_records": lambda: synth.make_scenario(33, 150, 0, seed=932),                     # 2 L = 66 rows: the record's last row
}


def _run(eng, sc):
    eng.stage(sc)
    r = eng.visual_update_staged(sc["sigma_img"])
    return r, eng.download_P()


@pytest.mark.parametrize("name", sorted(SHAPES))
def test_records_against_oracle_and_tiles(xk, oracle_c, name):
    sc = SHAPES[name]()
    ref = oracle_c.visual_update(sc)
    N, K = sc["n_poses_max"], len(sc["trk_off"]) - 1
    M = len(sc["slam_anchor_idxs"]) if "slam_anchor_idxs" in sc else 0
    out = {}
    for hl in (1, 0):
        eng = xk.LabEngine(N, M, max(K, 1))
        eng.set_option("caqr_hlite", hl)
        for geom in (3, 0):                               # 152 tiles (two first-level groups) / 184 tiles
            eng.set_option("pipe_split", geom)
            r, P = _run(eng, sc)
            st = eng.caqr_status()
            assert st["schedule"] == 2 and st["giveups"] == 0, st
            assert np.array_equal(r["inlier"], ref["inlier"]) and np.array_equal(r["inlier_slam"], ref["inlier_slam"])
            assert rel(P, ref["P"]) <= 1e-8 and rel(r["correction"], ref["correction"]) <= 1e-6, (hl, geom, rel(P, ref["P"]))
            out[(hl, geom)] = (P, r["correction"])
        eng.close()
    for geom in (3, 0):
        assert rel(out[(1, geom)][0], out[(0, geom)][0]) <= 1e-12, rel(out[(1, geom)][0], out[(0, geom)][0])
        assert rel(out[(1, geom)][1], out[(0, geom)][1]) <= 1e-9


@pytest.mark.parametrize("name", ["headline", "ragged", "narrow_with_slam"])
def test_gave_up_launch_is_redone_from_the_records(xk, oracle_c, name):
    """The abort word raised before the launch (lab hook "caqr_poison"): every workgroup gives up, the host redoes the update with
    the multi-launch schedule -- whose first tile pass forms its rows from the records the per-feature kernel left."""
    sc = SHAPES[name]()
    ref = oracle_c.visual_update(sc)
    N, K = sc["n_poses_max"], len(sc["trk_off"]) - 1
    M = len(sc["slam_anchor_idxs"]) if "slam_anchor_idxs" in sc else 0
    eng = xk.LabEngine(N, M, max(K, 1))
    r0, P0 = _run(eng, sc)                                # records, single launch
    assert eng.caqr_status()["schedule"] == 2
    eng.set_option("caqr_poison", 1)
    r1, P1 = _run(eng, sc)                                # records written, launch gives up, multi-launch forms its rows from them
    st = eng.caqr_status()
    assert st["giveups"] == 1 and st["schedule"] != 2, st
    eng.set_option("caqr_poison", 0)
    assert np.array_equal(r1["inlier"], ref["inlier"])
    assert rel(P1, ref["P"]) <= 1e-8 and rel(P1, P0) <= 1e-12 and rel(r1["correction"], r0["correction"]) <= 1e-9
    r2, P2 = _run(eng, sc)                                # fast path off for a while: tiles written directly, multi-launch
    assert rel(P2, P1) <= 1e-12
    eng.close()


@pytest.mark.parametrize("name,mk", [
    ("cfg3_like_n40", lambda: synth.make_scenario(40, 300, 0, seed=941)),
    ("n50_ragged", lambda: synth.make_scenario(50, 200, 0, seed=942, track_len=(2, 50))),
    ("n64_longest_records", lambda: synth.make_scenario(64, 120, 0, seed=943)),          # 2 L = 128 rows: the record's last rows
    ("n36_mostly_rejected", lambda: synth.make_scenario(36, 200, 0, seed=944, outlier_frac=0.7)),
])
def test_records_next_to_128_row_slots(xk, oracle_c, name, mk):
    """Windows of 34..64 poses: the multi-launch schedule, whose first tile pass forms the rows of every track from its record
    (xk_linalg.hip.h: XkCaqrArgs::Hc) -- against the C oracle and against tiles written by the per-feature kernel ("caqr_hlite" 0)."""
    sc = mk()
    ref = oracle_c.visual_update(sc)
    N, K = sc["n_poses_max"], len(sc["trk_off"]) - 1
    out = {}
    for hl in (1, 0):
        eng = xk.LabEngine(N, 0, K)
        eng.set_option("caqr_hlite", hl)
        r, P = _run(eng, sc)
        assert eng.caqr_status()["schedule"] == 3    # (round 6: the multi-launch schedule for the first panels, the last <= 192 columns in single launches)
        assert np.array_equal(r["inlier"], ref["inlier"])
        assert rel(P, ref["P"]) <= 1e-8 and rel(r["correction"], ref["correction"]) <= 1e-6, (hl, rel(P, ref["P"]))
        out[hl] = (P, r["correction"])
        eng.close()
    assert rel(out[1][0], out[0][0]) <= 1e-12 and rel(out[1][1], out[0][1]) <= 1e-9
