"""The Kalman-stage products behind the wide single launch and the multi-launch schedule use what is known about the compressed
system (XkGemmArgs::tri_a / tri_b / sym_cols, csrc/xk_linalg.hip.h): T = R is upper trapezoidal, only the upper triangle of S and
of its Schur complement is read, the lower tiles of P+ are mirror images.  XK_GEMM_STRUCT=0 runs the same products as general
ones: same gate verdicts, posterior equal to rounding (the K ranges of the waves differ, 1e-13 measured), and the structured posterior exactly symmetric (updater.cpp:131-133)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from helpers import with_lab

pytestmark = pytest.mark.gpu
ROOT = os.path.join(os.path.dirname(__file__), "..")

CHILD = r'''
import json, sys
sys.path.insert(0, ".")
import numpy as np
from x_multi_agent_amd import engine, synth
out = {}
for name, sc, opt in (("cfg2", synth.make_config(2), {}), ("n40", synth.make_scenario(40, 120, 0, seed=5151), {}),
                      ("headline_separate", synth.make_config(4), {"pipe_kalman": 0}), ("few_rows", synth.make_scenario(30, 4, 3, seed=5152), {})):
    N, K = sc["n_poses_max"], len(sc["trk_off"]) - 1
    M = len(sc["slam_anchor_idxs"]) if "slam_anchor_idxs" in sc else 0
    eng = engine.Engine(N, M, max(K, 1))
    for k, v in opt.items():
        eng.set_option(k, v)
    eng.stage(sc)
    r = eng.visual_update_staged(sc["sigma_img"])
    np.save(sys.argv[1] + "_" + name + ".npy", eng.download_P())
    out[name] = dict(inliers=int(r["inlier"].sum()), corr=[float(x) for x in r["correction"]])
    eng.close()
print(json.dumps(out))
'''


def _run(tmp_path, tag, struct):
    env = with_lab(dict(os.environ, XK_GEMM_STRUCT=str(struct)))      # (an A/B switch: the lab build reads it)
    r = subprocess.run([sys.executable, "-c", CHILD, str(tmp_path / tag)], capture_output=True, text=True, env=env, cwd=ROOT, timeout=600)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    return json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])


def test_structured_products_match_the_general_ones(tmp_path):
    a, b = _run(tmp_path, "s", 1), _run(tmp_path, "g", 0)
    for name in a:
        assert a[name]["inliers"] == b[name]["inliers"], name
        Ps, Pg = np.load(str(tmp_path / ("s_" + name + ".npy"))), np.load(str(tmp_path / ("g_" + name + ".npy")))
        assert np.array_equal(Ps, Ps.T), name + ": the mirrored posterior is exactly symmetric"
        assert np.linalg.norm(Ps - Pg) <= 1e-11 * np.linalg.norm(Pg), (name, np.linalg.norm(Ps - Pg) / np.linalg.norm(Pg))
        ca, cb = np.array(a[name]["corr"]), np.array(b[name]["corr"])
        assert np.linalg.norm(ca - cb) <= 1e-9 * max(np.linalg.norm(cb), 1e-300), name
