"""Stage times of a BASELINE config with the single-launch CAQR and with the multi-launch schedule: python tools/exp/cfg_stages.py 2"""
import os, sys
sys.path.insert(0, '.')
import os as _os; _os.environ.setdefault("XK_LIB_PATH", _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "..", "..", "x_multi_agent_amd", "lab", "libxk.so"))   # the lab build: env switches, hooks, probes (include/xk_lab.h)
from x_multi_agent_amd import engine, synth
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
N, K, M = synth.CONFIGS[cfg]
sc = synth.make_config(cfg)
for mode, env in (("single launch", {}), ("multi-launch", {"XK_CAQR_RESIDENT": "0"})):
    os.environ.pop("XK_CAQR_RESIDENT", None); os.environ.update(env)
    eng = engine.Engine(N, M, K)
    eng.stage(sc)
    t = eng.bench_staged(sc["sigma_img"], 5, 40)
    import time
    eng.stage(sc); eng.run_steps(sc["sigma_img"], 5)
    t0 = time.perf_counter(); eng.run_steps(sc["sigma_img"], 200); dt = (time.perf_counter() - t0) / 200
    print(f"config {cfg} {mode:14s}", {k: round(v["ms"], 4) for k, v in t["stages"].items() if v["ms"] > 0}, "total", round(t["total_ms"], 4),
          "leaves", t["n_leaf"], "rows", t["rows_stacked"], f"replay {1e3*dt:.4f} ms = {1/dt:.0f} updates/s", eng.caqr_status())
    eng.close()
