"""Round 6: cost of SMALL updates -- the frames of a real filter, where a handful of tracks end -- stage times and replay rate.
    python tools/exp/small_stack_cost.py"""
import sys, time
sys.path.insert(0, '.')
import numpy as np
from x_multi_agent_amd import engine, synth
from oracle import c_oracle
for (N, K, M, tl) in ((30, 3, 0, None), (30, 8, 0, (4, 12)), (30, 10, 0, None), (30, 20, 0, (4, 20)), (20, 5, 10, (3, 10)), (30, 6, 50, (4, 14)), (12, 4, 0, None)):
    kw = dict(seed=4300 + K)
    if tl: kw["track_len"] = tl
    sc = synth.make_scenario(N, K, M, **kw)
    rows = int(sum(2 * (sc["trk_off"][k + 1] - sc["trk_off"][k]) - 3 for k in range(K))) + 2 * M
    ref = c_oracle.visual_update(sc)
    eng = engine.Engine(N, M, K)
    eng.stage(sc)
    got = eng.visual_update_staged(sc["sigma_img"])
    relP = np.linalg.norm(eng.download_P() - ref["P"]) / np.linalg.norm(ref["P"])
    eng.stage(sc)
    t = eng.bench_staged(sc["sigma_img"], 3, 20)
    eng.stage(sc); eng.run_steps(sc["sigma_img"], 20)
    t0 = time.perf_counter(); eng.run_steps(sc["sigma_img"], 200); dt = (time.perf_counter() - t0) / 200
    print(f"N={N} K={K} M={M} n={15+6*N+3*M} nominal rows {rows}: rel dP {relP:.1e} masks {bool(np.array_equal(got['inlier'], ref['inlier']))}",
          {k: round(v["ms"], 4) for k, v in t["stages"].items() if v["ms"] > 0}, "launches", t["n_levels"], f"replay {1e3*dt:.4f} ms", eng.caqr_status()["schedule"], flush=True)
    eng.close()
