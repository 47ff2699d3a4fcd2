"""Round 6: stacks between n and 512 nominal rows (compressed by the multi-launch schedule today): would the single launch serve them?
    XK_PIPE_MIN_ROWS=1 python tools/exp/mid_stack_ab.py"""
import sys, time, os
sys.path.insert(0, '.')
import numpy as np
from x_multi_agent_amd import engine, synth
from oracle import c_oracle
for (N, K, M, tl) in ((30, 20, 0, (4, 20)), (30, 5, 0, None), (30, 12, 0, (8, 20)), (20, 30, 4, (3, 9)), (12, 10, 0, None), (30, 9, 0, None), (16, 6, 0, None)):
    kw = dict(seed=4400 + K)
    if tl: kw["track_len"] = tl
    sc = synth.make_scenario(N, K, M, **kw)
    rows = int(sum(2 * (sc["trk_off"][k + 1] - sc["trk_off"][k]) - 3 for k in range(K))) + 2 * M
    ref = c_oracle.visual_update(sc)
    eng = engine.LabEngine(N, M, K)
    res = []
    for rep in range(3):
        eng.stage(sc); eng.upload_P(sc["P"])
        got = eng.visual_update_staged(sc["sigma_img"])
        res.append(np.linalg.norm(eng.download_P() - ref["P"]) / np.linalg.norm(ref["P"]))
    eng.stage(sc)
    t = eng.bench_staged(sc["sigma_img"], 3, 20)
    print(f"N={N} K={K} M={M} n={15+6*N+3*M} rows {rows}: rel dP {max(res):.1e} masks {bool(np.array_equal(got['inlier'], ref['inlier']))}",
          {k: round(v["ms"], 4) for k, v in t["stages"].items() if v["ms"] > 0}, "total", round(t["total_ms"], 4), "launches", t["n_levels"], eng.caqr_status(), flush=True)
    eng.close()
