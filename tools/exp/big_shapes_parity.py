import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import os as _os; _os.environ.setdefault("XK_LIB_PATH", _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "..", "..", "x_multi_agent_amd", "lab", "libxk.so"))   # the lab build: env switches, hooks, probes (include/xk_lab.h)
from x_multi_agent_amd import engine, synth
from oracle import c_oracle
from helpers import rel
# shapes that take the less common schedules: arity-40 overlapped (401..800 tiles), 128-row tiles with the fused
# launch, column-split tile steps (C1 > 192), plain three-level sequence (> 800 tiles at arity 40 -> 21..40 groups)
cases = [(30, 640, 0, {}), (30, 800, 0, {}), (30, 401, 4, {}), (40, 300, 0, {}), (50, 420, 0, {}), (64, 120, 0, {}),
         (30, 300, 20, {}), (34, 60, 30, {}), (30, 900, 0, {}), (12, 1000, 0, {"track_len": (2, 12)})]
for N, K, M, kw in cases:
    t0 = time.time()
    sc = synth.make_scenario(N, K, M, seed=7000 + N + K + M, **kw)
    ref = c_oracle.visual_update(sc)
    eng = engine.Engine(N, M, K)
    got = eng.visual_update(sc); again = eng.visual_update(sc); eng.close()
    ok = np.array_equal(got["inlier"], ref["inlier"]) and np.array_equal(got["inlier_slam"], ref["inlier_slam"])
    print(N, K, M, kw, "ok" if ok else "MASK DIFFERS", f"rel dP {rel(got['P'], ref['P']):.2e} corr {rel(got['correction'], ref['correction']):.2e} repeat {rel(again['P'], got['P']):.1e} ({time.time()-t0:.0f} s)", flush=True)
