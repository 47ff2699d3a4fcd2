import os, sys, time
sys.path.insert(0, '.')
import numpy as np
import os as _os; _os.environ.setdefault("XK_LIB_PATH", _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "..", "..", "x_multi_agent_amd", "lab", "libxk.so"))   # the lab build: env switches, hooks, probes (include/xk_lab.h)
from x_multi_agent_amd import engine, synth
from oracle import c_oracle
for cfg in (1, 4):
    sc = synth.make_config(cfg)
    N, K, M = synth.CONFIGS[cfg]
    ref = c_oracle.visual_update(sc)
    for kal in (0, 1):
        eng = engine.Engine(N, M, K)
        eng.set_option("pipe_kalman", kal)
        eng.stage(sc)
        r = eng.visual_update_staged(sc["sigma_img"])
        P = eng.download_P()
        relP = np.linalg.norm(P - ref["P"]) / np.linalg.norm(ref["P"])
        relc = np.linalg.norm(r["correction"] - ref["correction"]) / np.linalg.norm(ref["correction"])
        asym = np.abs(P - P.T).max()
        eng.stage(sc)
        t = eng.bench_staged(sc["sigma_img"], 5, 50)
        eng.stage(sc); eng.run_steps(sc["sigma_img"], 300)
        t0 = time.perf_counter(); eng.run_steps(sc["sigma_img"], 500); dt = (time.perf_counter() - t0) / 500
        print(f"cfg{cfg} kal={kal} relP {relP:.2e} relcorr {relc:.2e} asym {asym:.1e} inl_ok {np.array_equal(r['inlier'], ref['inlier'])} "
              f"stages {({k: round(v['ms'], 4) for k, v in t['stages'].items() if v['ms'] > 0})} total {t['total_ms']:.4f} replay {1e3*dt:.4f} ms = {1/dt:.0f} upd/s {eng.caqr_status()}", flush=True)
        eng.close()
