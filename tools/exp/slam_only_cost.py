"""Round 6: cost of an update whose stack is SLAM rows only (no MSCKF track ended this frame: the common frame of a filter with persistent
features) -- stage times and replay rate.   python tools/exp/slam_only_cost.py"""
import sys, time
sys.path.insert(0, '.')
import numpy as np
from x_multi_agent_amd import engine, synth
from oracle import c_oracle
for (N, M) in ((30, 50), (20, 10), (12, 30)):
    sc = synth.make_scenario(N, 0, M, seed=4242)
    ref = c_oracle.visual_update(sc)
    eng = engine.Engine(N, M, 1)
    eng.stage(sc)
    got = eng.visual_update_staged(sc["sigma_img"])
    relP = np.linalg.norm(eng.download_P() - ref["P"]) / np.linalg.norm(ref["P"])
    eng.stage(sc)
    t = eng.bench_staged(sc["sigma_img"], 3, 20)
    eng.stage(sc); eng.run_steps(sc["sigma_img"], 20)
    t0 = time.perf_counter(); eng.run_steps(sc["sigma_img"], 200); dt = (time.perf_counter() - t0) / 200
    print(f"N={N} M={M} n={15+6*N+3*M}: rel dP {relP:.1e} masks {bool(np.array_equal(got['inlier_slam'], ref['inlier_slam']))}",
          {k: round(v["ms"], 4) for k, v in t["stages"].items() if v["ms"] > 0}, "launches", t["n_levels"], f"replay {1e3*dt:.4f} ms", eng.caqr_status(), flush=True)
    eng.close()
