"""The single-launch CAQR needs its 256 workgroups co-resident, one per CU.  What the host may rely on when they are not
(include/xk.h, xk_caqr_status): the update is still correct (the multi-launch schedule redoes it), the give-up is a bounded
retry -- spins give up after 2 ms, the whole update stays under 5 ms -- and the handle reports what happened.

Two tests: (1) a launch in which one workgroup never shows up (XK_CAQR_TEST_STALL: the worst case, every other workgroup
runs into the bound of its spin); (2) two agents in two processes on ONE GPU, fast path armed in both, against the same
two processes with the multi-launch schedule.  On this stack two processes' queues are arbitrated by the driver in slices of
80-90 ms -- a few updates of EITHER schedule wait that long for the GPU -- so what (2) asserts is what the fast path is
answerable for: same results, no slower median, and no more long updates than the multi-launch schedule sees."""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest

from x_multi_agent_amd import synth

pytestmark = pytest.mark.gpu

WORKER = r'''
import json, sys, time
import numpy as np
sys.path.insert(0, sys.argv[1])
from x_multi_agent_amd import engine, synth
ref = np.load(sys.argv[2])
sc = synth.make_config(4)
N, K = sc["n_poses_max"], len(sc["trk_off"]) - 1
eng = engine.Engine(N, 0, K)
if __import__("os").environ.get("TEST_MULTI_LAUNCH") == "1": eng.set_option("caqr_resident", 0)   # (operational switch of the release library)
eng.stage(sc); eng.visual_update_staged(sc["sigma_img"])          # warm-up (module load, first launches)
open(sys.argv[3], "w").write("ready")
while not all(__import__("os").path.exists(p) for p in sys.argv[4:]): time.sleep(0.001)
worst, relP, sched, times = 0.0, 0.0, [], []
for i in range(150):
    eng.stage(sc)
    t0 = time.perf_counter()
    r = eng.visual_update_staged(sc["sigma_img"])
    times.append(time.perf_counter() - t0)
    worst = max(worst, times[-1])
    if i % 10 == 0:
        P = eng.download_P()
        relP = max(relP, float(np.linalg.norm(P - ref["P"]) / np.linalg.norm(ref["P"])))
        assert np.array_equal(r["inlier"], ref["inlier"])
    sched.append(eng.caqr_status()["schedule"])
st = eng.caqr_status()
ts = sorted(times)
print(json.dumps(dict(median_ms=ts[len(ts)//2]*1e3, p99_ms=ts[-2]*1e3, slow=[(i, round(t*1e3,2)) for i, t in enumerate(times) if t > 2e-3], worst_ms=worst * 1e3, relP=relP, giveups=st["giveups"], fast=sum(1 for x in sched if x != 0), n=len(sched))))
'''


def test_missing_workgroup_costs_one_bounded_retry(xk, oracle_c):
    import time
    from helpers import rel
    sc = synth.make_config(4)
    ref = oracle_c.visual_update(sc)
    N, K = sc["n_poses_max"], len(sc["trk_off"]) - 1
    eng = xk.LabEngine(N, 0, K)
    eng.stage(sc); eng.visual_update_staged(sc["sigma_img"])
    eng.set_option("caqr_test_stall", 1)
    eng.stage(sc)
    t0 = time.perf_counter()
    r = eng.visual_update_staged(sc["sigma_img"])
    dt = time.perf_counter() - t0
    eng.set_option("caqr_test_stall", 0)
    st = eng.caqr_status()
    assert st["giveups"] == 1 and st["schedule"] == 0 and not st["armed"], st
    assert np.array_equal(r["inlier"], ref["inlier"]) and rel(eng.download_P(), ref["P"]) <= 1e-8
    assert 1e-3 < dt < 5e-3, dt                      # the spins' 2 ms bound + the multi-launch redo
    eng.close()


def _two(env):
    sc = synth.make_config(4)
    from oracle import c_oracle
    ref = c_oracle.visual_update(sc)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with tempfile.TemporaryDirectory() as td:
        np.savez(os.path.join(td, "ref.npz"), P=ref["P"], inlier=ref["inlier"])
        flags = [os.path.join(td, f"ready{i}") for i in range(2)]
        procs = [subprocess.Popen([sys.executable, "-c", WORKER, root, os.path.join(td, "ref.npz"), flags[i]] + flags,
                                  stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=dict(os.environ, **env)) for i in range(2)]
        outs = [p.communicate(timeout=300) for p in procs]
    res = []
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se[-2000:]
        res.append(json.loads(so.strip().splitlines()[-1]))
    return res


def test_two_processes_share_one_gpu():
    fast = _two({})
    slow = _two({"TEST_MULTI_LAUNCH": "1"})
    print("two tenants, fast path armed:", json.dumps(fast))
    print("two tenants, multi-launch:   ", json.dumps(slow))
    for r in fast + slow:
        assert r["relP"] <= 1e-8, (fast, slow)
    for r in fast:
        assert r["fast"] + 2 * r["giveups"] >= 0.5 * r["n"] or r["giveups"] > 0, fast   # the fast path ran (or said why not)
    assert max(r["median_ms"] for r in fast) <= 1.2 * max(r["median_ms"] for r in slow) + 0.1
    # long updates = the driver's arbitration between the two processes; the fast path must not add to them
    assert sum(len(r["slow"]) for r in fast) <= sum(len(r["slow"]) for r in slow) + 6, (fast, slow)
