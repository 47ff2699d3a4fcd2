"""The single launch picks its narrow geometry (184 tiles, or 152 with two first-level groups per XCD) by the acceptance ratio the LAST
launch reported -- a prediction.  Here the tracker's outlier rate MOVES across frames, 60 % -> 98 % -> 70 % of the tracks
passing the gate, through the C++ frame loop (Ekf::processImu x7 -> setMeasurement -> processUpdateMeasurement, covariance resident):
every frame's posterior must be right whichever geometry served it, a mis-prediction must cost a give-up (xk_caqr_status) and a redo,
not a wrong answer, and the handle must find its way back to the 152-tile geometry."""
import os
import subprocess

import numpy as np
import pytest

from helpers import rel
from x_multi_agent_amd import synth
from test_gpu_frame_loop import PKG, oracle_frame

pytestmark = pytest.mark.gpu


def _corrupt(sc, frac, seed):
    """A copy of the scenario in which `frac` of the tracks are mismatches: every observation moved by ~ 25 sigma."""
    rng = np.random.default_rng(seed)
    out = dict(sc)
    obs = sc["obs_xy"].copy()
    off = sc["trk_off"]
    K = len(off) - 1
    bad = rng.permutation(K)[:int(round(frac * K))]
    for k in bad:
        obs[off[k]:off[k + 1]] += rng.normal(0.0, 25.0 * sc["sigma_img"], size=(off[k + 1] - off[k], 2))
    out["obs_xy"] = obs
    return out


def test_acceptance_ratio_swings_across_frames(tmp_path):
    base = synth.make_config(4, err_scale=0.3, outlier_frac=0.0)              # ~ 97 % of the tracks pass
    sets = [_corrupt(base, 0.38, 1), base, _corrupt(base, 0.28, 2)]           # ~ 60 %, ~ 97 %, ~ 70 %
    N, K = base["n_poses_max"], len(base["trk_off"]) - 1
    n = 15 + 6 * N
    # frames: 3 x set 0 (the handle learns 60 %), 3 x set 1 (mis-prediction: the 152-tile launch overflows), then set 2 for long enough
    # that the 152-tile geometry comes back (it stays off for 64 updates after an overflow), then set 1 and set 0 again
    sched = [0] * 3 + [1] * 3 + [2] * 70 + [1] * 2 + [0] * 2
    frames = len(sched)
    parts = [np.array([N, K, frames, 7, 1, base["sigma_img"]], float), base["C_q_G"].ravel(), base["G_p_C"].ravel(),
             np.diff(base["trk_off"]).astype(float), sets[0]["obs_xy"].ravel(), np.asfortranarray(base["P"]).ravel(order="F"),
             np.array([2.0]), sets[1]["obs_xy"].ravel(), sets[2]["obs_xy"].ravel(), np.array(sched, float)]
    fin, fout = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    np.concatenate(parts).astype("<f8").tofile(fin)
    env = dict(os.environ, LD_LIBRARY_PATH=PKG + ":/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", ""))
    r = subprocess.run([os.path.join(PKG, "xk_frame_loop_example"), fin, fout], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    out = np.fromfile(fout, dtype="<f8")
    at = n * n + 7 * N + 16 + frames
    per = out[at:at + 5 * frames].reshape(frames, 5)                          # set, inliers, give-ups so far, schedule, rel dev vs the set's first frame
    Pf = out[at + 5 * frames:].reshape(3, n, n).transpose(0, 2, 1)            # (column-major on file)
    refs = [oracle_frame(s, 7) for s in sets]
    rates = [rf["inliers"] / K for rf in refs]
    assert 0.5 < rates[0] < 0.68 and rates[1] > 0.94 and 0.6 < rates[2] < 0.8, rates
    for s in range(3):
        assert rel(Pf[s], refs[s]["P"]) <= 1e-8, (s, rel(Pf[s], refs[s]["P"]))
    for f in range(frames):
        s = int(per[f, 0])
        assert int(per[f, 1]) == refs[s]["inliers"], (f, per[f])
        assert per[f, 4] <= 1e-11, (f, per[f])                                # the geometries differ in rounding only (1e-15)
        assert int(per[f, 3]) == 2, (f, per[f])                               # every frame ends up served by a single launch
    giveups = per[:, 2].astype(int)
    # two mis-predictions, both at the first frame of a run of set 1: queued on 60 % / 70 % with 152 tiles, 97 % found -- a give-up
    # (reason 9) and the update redone with 184 tiles, each time; the second one shows that the 152-tile geometry HAD come back after
    # its 64 updates off (frames 4..67 of set 2 ran on 184 tiles, 68..75 on 152 again)
    assert giveups[2] == 0 and giveups[3] == 1 and giveups[75] == 1 and giveups[76] == 2 and giveups[-1] == 2, giveups.tolist()
    print(f"moving acceptance: rates {[round(x, 3) for x in rates]}, give-ups {giveups[-1]}, worst rel dP vs the oracle "
          f"{max(rel(Pf[s], refs[s]['P']) for s in range(3)):.2e}, worst frame-to-frame deviation {per[:, 4].max():.2e}")
