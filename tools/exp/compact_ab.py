"""Device-side compaction of gated-out rows: stage times at the headline, config 2 and shapes past the old capacity cliff."""
import sys, time
sys.path.insert(0, '.')
import numpy as np
import os as _os; _os.environ.setdefault("XK_LIB_PATH", _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "..", "..", "x_multi_agent_amd", "lab", "libxk.so"))   # the lab build: env switches, hooks, probes (include/xk_lab.h)
from x_multi_agent_amd import engine, synth
from oracle import c_oracle
cases = [("cfg4", lambda: synth.make_config(4)), ("cfg2", lambda: synth.make_config(2)), ("cfg1", lambda: synth.make_config(1)),
         ("N30 K480", lambda: synth.make_scenario(30, 480, 0, seed=5101)), ("N30 K260 M50", lambda: synth.make_scenario(30, 260, 50, seed=5102)),
         ("N30 K560 (overflow)", lambda: synth.make_scenario(30, 560, 0, seed=5103))]
for name, mk in cases:
    sc = mk()
    N, K = sc["n_poses_max"], len(sc["trk_off"]) - 1
    M = len(sc["slam_anchor_idxs"]) if "slam_anchor_idxs" in sc else 0
    eng = engine.Engine(N, M, K)
    eng.stage(sc)
    r = eng.visual_update_staged(sc["sigma_img"])
    P = eng.download_P()
    st0 = eng.caqr_status()
    eng.stage(sc)
    t = eng.bench_staged(sc["sigma_img"], 5, 40)
    eng.stage(sc); eng.run_steps(sc["sigma_img"], 200)
    t0 = time.perf_counter(); eng.run_steps(sc["sigma_img"], 300); dt = (time.perf_counter() - t0) / 300
    ref = c_oracle.visual_update(sc) if K <= 480 else None
    relP = np.linalg.norm(P - ref["P"]) / np.linalg.norm(ref["P"]) if ref else float("nan")
    print(f"{name:22s} rows {t['rows_stacked']:6d} relP {relP:.1e} inl {int(r['inlier'].sum())}/{K} stages "
          f"{({k: round(v['ms'], 4) for k, v in t['stages'].items() if v['ms'] > 0})} replay {1e3*dt:.4f} ms = {1/dt:.0f} upd/s first {st0} now {eng.caqr_status()}", flush=True)
    eng.close()
