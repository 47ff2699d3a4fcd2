"""SPLIT compression of systems with SLAM features (round 6; xk_handle::d_R2).  The rows of MSCKF tracks are zero in the features' columns
(msckf_update.cpp:412-416) and every feature brings two rows of its own (slam_update.cpp), so only the tracks' rows are compressed
(VioUpdater::applyQRDecomposition, vio_updater.cpp:487-512), in the 6 N pose columns + the residual; the features' 2 M rows join the
compressed system as built: T = [R1 | 0 | z1 ; H_slam | res_slam], T^T T = H^T H and T^T z = H^T res exactly as for the R of the whole
stack.  Taken where the update cannot ride inside the launch (n > 206: BASELINE config 2).  Against the C oracle and against the
compression of the whole stack ("slam_split" 0), over shapes, histories with changing inputs on one handle, the iterated update, the
MULTI_UAV order's two-call form, a launch that gives up."""
import ctypes as C

import numpy as np
import pytest

from helpers import rel
from x_multi_agent_amd import synth

pytestmark = pytest.mark.gpu

SHAPES = {
    "cfg2": lambda: synth.make_config(2),
    "slam_heavy": lambda: synth.make_scenario(20, 150, 40, seed=7101),                       # n = 255: narrow launch on 121 columns
    "slam_ragged": lambda: synth.make_scenario(30, 120, 25, seed=7102, track_len=(2, 30)),
    "widest": lambda: synth.make_scenario(33, 120, 61, seed=7103),                            # 6 N + 1 = 199 > 192: the wide geometry on the pose columns
    "partial_window": lambda: synth.make_scenario(30, 140, 30, seed=7104, n_poses=19),
    "many_rejected": lambda: synth.make_scenario(28, 200, 36, seed=7105, outlier_frac=0.5),
    "few_features": lambda: synth.make_scenario(32, 160, 3, seed=7106),                       # n = 216
}


def _dims(sc):
    N, K = sc["n_poses_max"], len(sc["trk_off"]) - 1
    M = len(sc["slam_anchor_idxs"]) if "slam_anchor_idxs" in sc else 0
    return N, M, K


def _update(eng, sc, ms_tracks=None):
    eng.stage(sc)
    if ms_tracks is not None:
        eng.stage_msckf_slam(ms_tracks)
    r = eng.visual_update_staged(sc["sigma_img"])
    return r, eng.download_P()


@pytest.mark.parametrize("name", sorted(SHAPES))
def test_split_against_the_oracle_and_the_whole_stack(xk, oracle_c, name):
    sc = SHAPES[name]()
    ref = oracle_c.visual_update(sc)
    N, M, K = _dims(sc)
    out = {}
    for split in (1, 0):
        eng = xk.Engine(N, M, K)
        eng.set_option("slam_split", split)
        for rep in range(3):                      # (the first launch of a handle runs the 184-tile geometry, the following ones may not)
            r, P = _update(eng, sc)
            st = eng.caqr_status()
            assert st["schedule"] == 2 and st["giveups"] == 0, (split, rep, st)
            assert np.array_equal(r["inlier"], ref["inlier"]) and np.array_equal(r["inlier_slam"], ref["inlier_slam"]), (split, rep)
            assert rel(P, ref["P"]) <= 1e-8 and rel(r["correction"], ref["correction"]) <= 1e-6, (split, rep, rel(P, ref["P"]))
        eng.stage(sc)
        t = eng.bench_staged(sc["sigma_img"], 0, 1)
        out[split] = (P, r["correction"], t)
        eng.close()
    assert rel(out[1][0], out[0][0]) <= 1e-10 and rel(out[1][1], out[0][1]) <= 1e-8
    # the split launch factors 6 N + 1 columns: a narrow geometry (184 / 152 tiles) wherever those fit 192
    if 6 * N + 1 <= 192:
        assert out[1][2]["n_leaf"] in (184, 152) and out[0][2]["n_leaf"] == 152, (out[1][2]["n_leaf"], out[0][2]["n_leaf"])


def test_changing_inputs_on_one_handle(xk, oracle_c):
    """Three different scenarios of BASELINE config 2's shape taken in turn on ONE handle (different windows, tracks, features, priors):
    whatever a launch finds in the hand-off slabs of the launches before it must not reach its result.  (A replay of identical inputs
    cannot see that: stale values and fresh ones are the same numbers.)"""
    N, K, M = synth.CONFIGS[2]
    scs = [synth.make_scenario(N, K, M, seed=7200 + i, outlier_frac=f) for i, f in enumerate((0.05, 0.4, 0.0))]
    refs = [oracle_c.visual_update(s) for s in scs]
    eng = xk.Engine(N, M, K)
    for it in range(12):
        i = (it * 2 + it // 3) % 3
        eng.upload_P(scs[i]["P"])
        r, P = _update(eng, scs[i])
        assert np.array_equal(r["inlier"], refs[i]["inlier"]) and np.array_equal(r["inlier_slam"], refs[i]["inlier_slam"]), it
        assert rel(P, refs[i]["P"]) <= 1e-8 and rel(r["correction"], refs[i]["correction"]) <= 1e-6, (it, i, rel(P, refs[i]["P"]))
    assert eng.caqr_status()["giveups"] == 0
    eng.close()


def test_tracks_that_become_features_ride_along(xk, oracle_c):
    """MSCKF-SLAM tracks (msckf_slam_update.cpp:64-267: their rows touch pose columns only) between the tracks and the SLAM rows."""
    sc = synth.make_scenario(30, 150, 24, seed=7301)
    tr = synth.tracks_as_list(sc)
    sc2 = dict(sc)
    sc2["trk_off"] = sc["trk_off"][:141].copy()
    sc2["obs_xy"] = sc["obs_xy"][:sc["trk_off"][140]].copy()
    ms = tr[140:150]
    N, M, _ = _dims(sc)
    out = {}
    for split in (1, 0):
        eng = xk.Engine(N, M, 150)
        eng.set_option("slam_split", split)
        r, P = _update(eng, sc2, ms)
        assert eng.caqr_status()["schedule"] == 2
        out[split] = (P, r["correction"], r["inlier"].copy())
        eng.close()
    assert np.array_equal(out[1][2], out[0][2])
    assert rel(out[1][0], out[0][0]) <= 1e-10 and rel(out[1][1], out[0][1]) <= 1e-8


def test_iterated_update_and_the_two_call_form(xk, oracle_c):
    """A pass with correction_total != 0 queued whole (xk_build_compress_update_pass_async -> xk_apply_update), and the MULTI_UAV order's
    xk_build_compress_async ... xk_apply_update, at n = 345: the compressed system they apply is the split one."""
    sc = synth.make_config(2)
    N, M, K = _dims(sc)
    rng = np.random.default_rng(5)
    ct = 1e-3 * rng.standard_normal(15 + 6 * N + 3 * M)
    res = {}
    for split in (1, 0):
        eng = xk.Engine(N, M, K)
        eng.set_option("slam_split", split)
        eng.stage(sc)
        eng.build_compress_update_pass_async(sc["sigma_img"], ct, True)
        c1 = eng.apply_update(ct, True)
        P1 = eng.download_P()
        eng.upload_P(sc["P"])
        eng.stage(sc)
        assert eng.L.xk_build_compress_async(eng.h, C.c_double(sc["sigma_img"])) == 0
        c2 = eng.apply_update(None, True)
        P2 = eng.download_P()
        res[split] = (c1, P1, c2, P2)
        eng.close()
    ref = oracle_c.visual_update(sc)
    assert rel(res[1][3], ref["P"]) <= 1e-8 and rel(res[1][2], ref["correction"]) <= 1e-6
    for a, b in zip(res[1], res[0]):
        assert rel(a, b) <= 1e-8, rel(a, b)


def test_reference_shaped_compression_keeps_the_whole_stack(xk):
    """xk_qr_compress hands out the reference's T_H: upper triangular, 6 N + 3 M rows -- never the split form."""
    sc = synth.make_config(2)
    N, M, K = _dims(sc)
    eng = xk.Engine(N, M, K)
    eng.stage(sc)
    eng.msckf_build(sc["sigma_img"])
    T, z = eng.qr_compress()
    na = 6 * N + 3 * M
    assert np.allclose(np.tril(T[:na, 15:15 + na], -1), 0.0)
    assert np.abs(np.diag(T[:na, 15:15 + na])[6 * N:6 * N + 2 * M]).max() > 0      # rows of the features' columns are there
    corr = eng.apply_update(None, True)                                              # ... and xk_apply_update applies THAT system
    ref_eng = xk.Engine(N, M, K)
    r, P = _update(ref_eng, sc)
    assert rel(eng.download_P(), P) <= 1e-10 and rel(corr, r["correction"]) <= 1e-8
    eng.close(); ref_eng.close()


def test_split_launch_that_gives_up(xk, oracle_c):
    sc = synth.make_config(2)
    ref = oracle_c.visual_update(sc)
    N, M, K = _dims(sc)
    eng = xk.LabEngine(N, M, K)
    r0, P0 = _update(eng, sc)
    eng.set_option("caqr_poison", 1)
    r1, P1 = _update(eng, sc)
    eng.set_option("caqr_poison", 0)
    st = eng.caqr_status()
    assert st["giveups"] == 1 and st["schedule"] == 0, st
    assert np.array_equal(r1["inlier"], ref["inlier"]) and rel(P1, ref["P"]) <= 1e-8 and rel(P1, P0) <= 1e-10
    r2, P2 = _update(eng, sc)                              # (multi-launch schedule on the whole stack while the fast path is off)
    assert rel(P2, ref["P"]) <= 1e-8
    eng.close()


@pytest.mark.parametrize("N,M", [(30, 50), (20, 10), (12, 30), (40, 12)])
def test_slam_rows_alone_are_not_compressed(xk, oracle_c, N, M):
    """No track ended this frame: the stack is the features' 2 M rows against n > 3 M columns.  The reference compresses only when rows >
    columns (vio_updater.cpp:487); neither does this -- no QR launch at all (xk_caqr_status: schedule 4), the rows go to the update as built.
    Against the C oracle, against the compression the option switches back on, through the queued and the two-call form (at n <= 206 the latter
    used to DEFER the compression behind xk_apply_update: nothing to defer here)."""
    sc = synth.make_scenario(N, 0, M, seed=7400 + N)
    ref = oracle_c.visual_update(sc)
    out = {}
    for split in (1, 0):
        eng = xk.Engine(N, M, 1)
        eng.set_option("slam_split", split)
        for rep in range(2):
            r, P = _update(eng, sc)
            assert (eng.caqr_status()["schedule"] == 4) == bool(split), (split, eng.caqr_status())
            assert np.array_equal(r["inlier_slam"], ref["inlier_slam"])
            assert rel(P, ref["P"]) <= 1e-8 and rel(r["correction"], ref["correction"]) <= 1e-6, (split, rep, rel(P, ref["P"]))
        out[split] = (P, r["correction"])
        if split:
            eng.upload_P(sc["P"]); eng.stage(sc)
            assert eng.L.xk_build_compress_update_async(eng.h, C.c_double(sc["sigma_img"])) == 0
            c1 = eng.apply_update(None, True)
            assert rel(eng.download_P(), P) <= 1e-12 and rel(c1, r["correction"]) <= 1e-10
            eng.upload_P(sc["P"]); eng.stage(sc)
            assert eng.L.xk_build_compress_async(eng.h, C.c_double(sc["sigma_img"])) == 0
            c2 = eng.apply_update(None, True)
            assert rel(eng.download_P(), P) <= 1e-12 and rel(c2, r["correction"]) <= 1e-10
            assert eng.caqr_status()["schedule"] == 4
        eng.close()
    assert rel(out[1][0], out[0][0]) <= 1e-10 and rel(out[1][1], out[0][1]) <= 1e-8


def test_frames_with_and_without_tracks_on_one_handle(xk, oracle_c):
    """A filter's life: frames whose stack is SLAM rows only between frames that also carry tracks, on one handle (n = 345)."""
    N, K, M = synth.CONFIGS[2]
    full = synth.make_config(2)
    only = synth.make_scenario(N, 0, M, seed=7500)
    refs = {"full": oracle_c.visual_update(full), "only": oracle_c.visual_update(only)}
    eng = xk.Engine(N, M, K)
    for it, which in enumerate(("only", "full", "only", "only", "full", "full", "only")):
        sc = full if which == "full" else only
        eng.upload_P(sc["P"])
        r, P = _update(eng, sc)
        assert eng.caqr_status()["schedule"] == (2 if which == "full" else 4), (it, eng.caqr_status())
        assert rel(P, refs[which]["P"]) <= 1e-8 and rel(r["correction"], refs[which]["correction"]) <= 1e-6, (it, which)
    assert eng.caqr_status()["giveups"] == 0
    eng.close()


SMALL = {
    # a handful of tracks ended this frame: nominal rows <= n -> not compressed (vio_updater.cpp:487), schedule 4
    "three_full_tracks": lambda: synth.make_scenario(30, 3, 0, seed=7601),
    "eight_short_tracks": lambda: synth.make_scenario(30, 8, 0, seed=7602, track_len=(4, 12)),
    "short_tracks_and_features": lambda: synth.make_scenario(20, 5, 10, seed=7603, track_len=(3, 10)),
    "cfg2_window_six_tracks": lambda: synth.make_scenario(30, 6, 50, seed=7604, track_len=(4, 14)),
    "small_window": lambda: synth.make_scenario(12, 4, 0, seed=7605),
    "tall_window_two_tracks": lambda: synth.make_scenario(40, 2, 0, seed=7606),                     # 128-row slots
    "half_rejected": lambda: synth.make_scenario(24, 10, 4, seed=7607, track_len=(3, 8), outlier_frac=0.5),
    "one_track": lambda: synth.make_scenario(16, 1, 0, seed=7608),
}


@pytest.mark.parametrize("name", sorted(SMALL))
def test_small_stacks_are_not_compressed(xk, oracle_c, name):
    sc = SMALL[name]()
    ref = oracle_c.visual_update(sc)
    N, M, K = _dims(sc)
    n = 15 + 6 * N + 3 * M
    assert sum(2 * (sc["trk_off"][k + 1] - sc["trk_off"][k]) - 3 for k in range(K)) + 2 * M <= n
    out = {}
    for split in (1, 0):
        eng = xk.Engine(N, M, K)
        eng.set_option("slam_split", split)
        for rep in range(2):
            r, P = _update(eng, sc)
            st = eng.caqr_status()
            assert (st["schedule"] == 4) == bool(split) and st["giveups"] == 0, (split, st)
            assert np.array_equal(r["inlier"], ref["inlier"]) and np.array_equal(r["inlier_slam"], ref["inlier_slam"])
            assert rel(P, ref["P"]) <= 1e-8 and rel(r["correction"], ref["correction"]) <= 1e-6, (split, rep, rel(P, ref["P"]))
        out[split] = (P, r["correction"])
        eng.close()
    assert rel(out[1][0], out[0][0]) <= 1e-10 and rel(out[1][1], out[0][1]) <= 1e-8


def test_small_stack_with_tracks_that_become_features(xk):
    """MSCKF-SLAM tracks' slots sit between the tracks' and the SLAM rows: the uncompressed stack keeps that order (vio_updater.cpp:406-422)."""
    sc = synth.make_scenario(30, 6, 8, seed=7610, track_len=(4, 10))
    tr = synth.tracks_as_list(sc)
    sc2 = dict(sc)
    sc2["trk_off"] = sc["trk_off"][:5].copy()
    sc2["obs_xy"] = sc["obs_xy"][:sc["trk_off"][4]].copy()
    ms = tr[4:6]
    N, M, _ = _dims(sc)
    out = {}
    for split in (1, 0):
        eng = xk.Engine(N, M, 6)
        eng.set_option("slam_split", split)
        r, P = _update(eng, sc2, ms)
        assert (eng.caqr_status()["schedule"] == 4) == bool(split)
        out[split] = (P, r["correction"], r["inlier"].copy())
        eng.close()
    assert np.array_equal(out[1][2], out[0][2])
    assert rel(out[1][0], out[0][0]) <= 1e-10 and rel(out[1][1], out[0][1]) <= 1e-8


def test_small_and_large_frames_on_one_handle(xk, oracle_c):
    """Frames with 400 tracks, 3 tracks, 20 short tracks (398 rows > n: compressed, by the single launch like every stack that needs it since
    round 6) in turn on one handle, each against the oracle; and the reference-shaped call still compresses."""
    N = 30
    big = synth.make_config(4)
    small = synth.make_scenario(N, 3, 0, seed=7601)
    mid = synth.make_scenario(N, 20, 0, seed=4320, track_len=(4, 20))
    refs = [oracle_c.visual_update(s) for s in (big, small, mid)]
    eng = xk.Engine(N, 0, 400)
    for it, i in enumerate((1, 0, 1, 2, 0, 1, 1, 2)):
        sc = (big, small, mid)[i]
        eng.upload_P(sc["P"])
        r, P = _update(eng, sc)
        assert eng.caqr_status()["schedule"] == (2, 4, 2)[i], (it, eng.caqr_status())
        assert np.array_equal(r["inlier"], refs[i]["inlier"])
        assert rel(P, refs[i]["P"]) <= 1e-8 and rel(r["correction"], refs[i]["correction"]) <= 1e-6, (it, i)
    eng.upload_P(small["P"]); eng.stage(small)
    eng.msckf_build(small["sigma_img"])
    T, z = eng.qr_compress()
    assert np.allclose(np.tril(T[:180, 15:195], -1), 0.0) and eng.caqr_status()["schedule"] != 4
    corr = eng.apply_update(None, True)
    assert rel(eng.download_P(), refs[1]["P"]) <= 1e-8 and rel(corr, refs[1]["correction"]) <= 1e-6
    eng.close()
