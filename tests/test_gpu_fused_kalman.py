"""The Kalman update INSIDE the compression launch (xk_pipe_kalman, csrc/xk_caqr_pipe.hip.h): Updater::applyUpdate
(src/x/ekf/updater.cpp:117-141, correction_total = 0, cov_update) applied block by block as the panels of the QR finish, on
the one workgroup of the launch the merge tree does not need.  Against the same update with the separate Kalman launches
(xk_set_option "pipe_kalman" 0) and against the C oracle, over every window size the narrow geometry serves -- the state
dimension walks across the 192-column tile border (n = 189 ... 201) -- with SLAM features, ragged tracks and a launch that gives
up."""
import ctypes as C

import numpy as np
import pytest

from helpers import rel
from x_multi_agent_amd import synth

pytestmark = pytest.mark.gpu

SHAPES = {
    "headline_n195": lambda: synth.make_config(4),
    "cfg1_n75": lambda: synth.make_config(1),
    "n29_k120_n189": lambda: synth.make_scenario(29, 120, 0, seed=4101),            # n = 189: nothing past the 192-column border
    "n31_k90_n201": lambda: synth.make_scenario(31, 90, 0, seed=4102),              # n = 201: nine state columns past it
    "n16_k400_n111": lambda: synth.make_scenario(16, 400, 0, seed=4103),
    "slam_n24_m8_n183": lambda: synth.make_scenario(24, 150, 8, seed=4104),         # persistent features in the narrow geometry
    "slam_n28_m7_n204": lambda: synth.make_scenario(28, 100, 7, seed=4105),
    "ragged_n30": lambda: synth.make_scenario(30, 200, 0, seed=4106, track_len=(2, 30)),
    "large_prior_n30": lambda: synth.make_scenario(30, 150, 0, seed=4107, prior_scale=100.0),
    "rows_176_n8": lambda: synth.make_scenario(8, 60, 0, seed=4108),                # na = 48: the last panel holds the residual column only
}


def _update(xk, sc, kalman):
    N, K = sc["n_poses_max"], len(sc["trk_off"]) - 1
    M = len(sc["slam_anchor_idxs"]) if "slam_anchor_idxs" in sc else 0
    eng = xk.LabEngine(N, M, max(K, 1))
    eng.set_option("pipe_kalman", kalman)
    eng.stage(sc)
    r = eng.visual_update_staged(sc["sigma_img"])
    P = eng.download_P()
    eng.stage(sc)
    t = eng.bench_staged(sc["sigma_img"], 0, 1)
    st = eng.caqr_status()
    eng.close()
    return r, P, t, st


@pytest.mark.parametrize("name", sorted(SHAPES))
def test_update_inside_the_launch_matches_separate_launches_and_oracle(xk, oracle_c, name):
    sc = SHAPES[name]()
    ref = oracle_c.visual_update(sc)
    ra, Pa, ta, sa = _update(xk, sc, 1)
    rb, Pb, tb, sb = _update(xk, sc, 0)
    assert sa["schedule"] == 2 and sb["schedule"] == 2, (sa, sb)            # both took the single launch
    assert ta["stages"]["xk_kalman_update"]["launches"] == 0 and tb["stages"]["xk_kalman_update"]["launches"] > 0
    assert np.array_equal(ra["inlier"], ref["inlier"]) and np.array_equal(rb["inlier"], ref["inlier"])
    assert rel(Pa, Pb) <= 1e-11, rel(Pa, Pb)                                # block-sequential == batch, to rounding
    assert rel(Pa, ref["P"]) <= 1e-8 and rel(ra["correction"], ref["correction"]) <= 1e-6
    assert rel(ra["correction"], rb["correction"]) <= 1e-8
    assert np.array_equal(Pa, Pa.T)                                         # symmetric bit for bit (updater.cpp:133)
    w = np.linalg.eigvalsh(Pa)
    assert w.min() > -1e-12 * w.max()


def test_wide_systems_keep_the_separate_launches(xk, oracle_c):
    """BASELINE config 2 (n = 345): [P | d] does not fit one CU's registers -- the single launch compresses, the Kalman launches
    follow, whatever the option says."""
    sc = synth.make_config(2)
    r, P, t, st = _update(xk, sc, 1)
    assert st["schedule"] == 2 and t["stages"]["xk_kalman_update"]["launches"] > 0
    assert rel(P, oracle_c.visual_update(sc)["P"]) <= 1e-8


def test_launch_that_gives_up_with_the_update_inside(xk, oracle_c):
    """Abort word raised before the launch: the Kalman role gives up with everybody else, the host redoes rows, compression and
    update with the multi-launch schedule -- through xk_visual_update_staged and through the queued form the C++ mirror uses
    (xk_build_compress_update_async -> xk_apply_update, which must not sit out the marker's time-out)."""
    import time
    sc = synth.make_config(4)
    ref = oracle_c.visual_update(sc)
    N, K = sc["n_poses_max"], len(sc["trk_off"]) - 1
    eng = xk.LabEngine(N, 0, K)
    eng.set_option("caqr_poison", 1)
    eng.stage(sc)
    r = eng.visual_update_staged(sc["sigma_img"])
    assert rel(eng.download_P(), ref["P"]) <= 1e-8 and eng.caqr_status()["giveups"] == 1
    eng.close()
    eng = xk.LabEngine(N, 0, K)
    eng.stage(sc)
    eng.visual_update_staged(sc["sigma_img"])                          # (warm: first-use costs out of the timing below)
    for poison in (0, 1):
        eng.set_option("caqr_poison", poison)
        eng.stage(sc)
        t0 = time.perf_counter()
        assert eng.L.xk_build_compress_update_async(eng.h, C.c_double(sc["sigma_img"])) == 0
        corr = eng.apply_update(None, True)
        dt = time.perf_counter() - t0
        assert rel(eng.download_P(), ref["P"]) <= 1e-8 and rel(corr, ref["correction"]) <= 1e-6, poison
        assert dt < 0.05, dt                                            # (the marker's fallback is a one-second spin)
    assert eng.caqr_status()["giveups"] == 1
    # what the queued form refuses: it queued applyUpdate(correction_total = 0, cov_update = true) and nothing else
    eng.set_option("caqr_poison", 0)
    eng.stage(sc)
    assert eng.L.xk_build_compress_update_async(eng.h, C.c_double(sc["sigma_img"])) == 0
    with pytest.raises(RuntimeError):
        eng.apply_update(np.ones(eng.n), True)
    eng.close()


@pytest.mark.parametrize("name", ["headline_n195", "cfg1_n75", "n31_k90_n201", "slam_n24_m8_n183", "ragged_n30"])
def test_iterated_pass_inside_the_launch(xk, name):
    """One pass of the iterated update (updater.cpp:99-110, iekf_iter > 1) queued whole: correction_total != 0 enters the Kalman
    role as the start value of its correction column (d = -ct: the block recurrence d += K_k (z_k - T_k d) then ends at
    K (res + H ct) - ct, updater.cpp:126-128), cov_update = 0 sends the role's posterior to a scratch matrix and leaves the
    prior.  Against the separate Kalman launches and against NumPy on the compressed system the device itself produced."""
    sc = SHAPES[name]()
    N, K = sc["n_poses_max"], len(sc["trk_off"]) - 1
    M = len(sc["slam_anchor_idxs"]) if "slam_anchor_idxs" in sc else 0
    rng = np.random.default_rng(77)
    eng = xk.LabEngine(N, M, max(K, 1))
    n = eng.n
    ct = 1e-3 * rng.standard_normal(n)
    # the compressed system and the prior, as the device has them
    eng.stage(sc)
    P0 = eng.download_P()
    eng.msckf_build(sc["sigma_img"])
    T, z = eng.qr_compress()
    S = T @ P0 @ T.T + sc["sigma_img"] ** 2 * np.eye(n)
    Kg = np.linalg.solve(S, T @ P0).T
    want_corr = Kg @ (z + T @ ct) - ct
    want_P = (np.eye(n) - Kg @ T) @ P0
    want_P = 0.5 * (want_P + want_P.T)
    for kalman in (1, 0):
        eng.set_option("pipe_kalman", kalman)
        for cov in (False, True):
            eng.stage(sc)
            eng.build_compress_update_pass_async(sc["sigma_img"], ct, cov)
            with pytest.raises(RuntimeError):
                eng.apply_update(2.0 * ct, cov)                         # not the pass that was queued: refused, still collectable
            with pytest.raises(RuntimeError):
                eng.apply_update(ct, not cov)
            corr = eng.apply_update(ct, cov)
            P = eng.download_P()
            assert eng.caqr_status()["schedule"] == 2 and eng.caqr_status()["giveups"] == 0
            assert rel(corr, want_corr) <= 1e-7, (kalman, cov, rel(corr, want_corr))
            assert rel(P, want_P if cov else P0) <= (1e-8 if cov else 0.0), (kalman, cov)
    # zero correction_total through the pass entry point == the one-pass entry point
    eng.set_option("pipe_kalman", 1)
    eng.stage(sc)
    eng.build_compress_update_pass_async(sc["sigma_img"], np.zeros(n), True)
    c0 = eng.apply_update(None, True)
    Pa = eng.download_P()
    eng.stage(sc)
    assert eng.L.xk_build_compress_update_async(eng.h, C.c_double(sc["sigma_img"])) == 0
    c1 = eng.apply_update(np.zeros(n), True)
    assert np.array_equal(c0, c1) and np.array_equal(Pa, eng.download_P())
    eng.close()


@pytest.mark.parametrize("poison", [0, 1])
def test_two_call_form_defers_the_compression_behind_what_rewrites_the_covariance(xk, poison):
    """MULTI_UAV order (updater.cpp:84-97): constructUpdate, then applyCI entries that REPLACE the covariance, then applyUpdate on
    the [T_H | z] that was linearised and gated at the prior.  xk_build_compress_async queues the rows only where the single launch
    can take the Kalman update along; xk_apply_update queues the compression with the Kalman role on the covariance of that
    moment -- and, if that launch gives up, redoes the compression WITHOUT rebuilding the rows (they belong to the old prior)."""
    sc = synth.make_config(4)
    N, K = sc["n_poses_max"], len(sc["trk_off"]) - 1
    eng = xk.LabEngine(N, 0, K)
    n = eng.n
    eng.stage(sc)
    P0 = eng.download_P()
    flags = eng.msckf_build(sc["sigma_img"])
    T, z = eng.qr_compress()                                            # gated and linearised at P0
    rng = np.random.default_rng(3)
    A = np.eye(n) + 0.05 * rng.standard_normal((n, n)) / np.sqrt(n)
    P1 = A @ P0 @ A.T                                                    # what the CI entries leave: another SPD matrix
    P1 = 0.5 * (P1 + P1.T)
    S = T @ P1 @ T.T + sc["sigma_img"] ** 2 * np.eye(n)
    Kg = np.linalg.solve(S, T @ P1).T
    want_corr = Kg @ z
    want_P = (np.eye(n) - Kg @ T) @ P1
    want_P = 0.5 * (want_P + want_P.T)
    eng.set_option("caqr_poison", poison)
    eng.stage(sc)
    assert eng.L.xk_build_compress_async(eng.h, C.c_double(sc["sigma_img"])) == 0
    eng.upload_P(P1)                                                     # (xk_apply_ci_resident ends with the same pointer swap)
    corr = eng.apply_update(None, True)
    P = eng.download_P()
    st = eng.caqr_status()
    assert st["giveups"] == poison and (st["schedule"] == 2) == (poison == 0), st
    assert rel(corr, want_corr) <= 1e-7 and rel(P, want_P) <= 1e-8, (rel(corr, want_corr), rel(P, want_P))
    eng.set_option("caqr_poison", 0)
    eng.close()
