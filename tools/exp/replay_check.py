import sys, time
sys.path.insert(0, '.')
import os as _os; _os.environ.setdefault("XK_LIB_PATH", _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "..", "..", "x_multi_agent_amd", "lab", "libxk.so"))   # the lab build: env switches, hooks, probes (include/xk_lab.h)
from x_multi_agent_amd import engine, synth
for cfg in (1, 1, 4):
    sc = synth.make_config(cfg)
    N, K, M = synth.CONFIGS[cfg]
    eng = engine.Engine(N, M, K)
    eng.stage(sc)
    for rep in range(3):
        eng.run_steps(sc["sigma_img"], 300)
        t0 = time.perf_counter(); eng.run_steps(sc["sigma_img"], 500); dt = (time.perf_counter() - t0) / 500
        print(cfg, rep, f"{1e3*dt:.4f} ms", eng.caqr_status(), flush=True)
    eng.close()
