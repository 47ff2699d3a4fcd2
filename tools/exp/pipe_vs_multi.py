"""QR-stage time of the single launch (xk_caqr_pipe) against the multi-launch schedule over shapes of its window, and the
replay rate of BASELINE config 1."""
import os, sys, time
sys.path.insert(0, '.')
import os as _os; _os.environ.setdefault("XK_LIB_PATH", _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "..", "..", "x_multi_agent_amd", "lab", "libxk.so"))   # the lab build: env switches, hooks, probes (include/xk_lab.h)
from x_multi_agent_amd import engine, synth
shapes = [(12, 26, 0), (10, 50, 0), (20, 60, 0), (30, 100, 0), (30, 200, 0), (24, 350, 0), (16, 400, 0), (30, 400, 0), (30, 410, 0), (20, 150, 40), (30, 200, 50), (33, 150, 0)]
for N, K, M in shapes:
    sc = synth.make_config(1) if (N, K, M) == (10, 50, 0) else synth.make_config(2) if (N, K, M) == (30, 200, 50) else synth.make_config(4) if (N, K, M) == (30, 400, 0) else synth.make_scenario(N, K, M, seed=77 + N + K)
    row = []
    for mode, env in (("single", {}), ("multi", {"XK_CAQR_RESIDENT": "0"})):
        os.environ.pop("XK_CAQR_RESIDENT", None); os.environ.update(env)
        eng = engine.Engine(N, M, K)
        eng.stage(sc)
        t = eng.bench_staged(sc["sigma_img"], 5, 40)
        qr = sum(v["ms"] for k, v in t["stages"].items() if "caqr" in k)
        eng.stage(sc); eng.run_steps(sc["sigma_img"], 300)       # (long warm-up: the clocks sag while the host builds the next scenario)
        t0 = time.perf_counter(); eng.run_steps(sc["sigma_img"], 200); dt = (time.perf_counter() - t0) / 200
        row.append(f"{mode} QR {qr:.3f} ms ({t['n_leaf']} leaves), {1/dt:.0f} upd/s")
        eng.close()
    print(f"N={N:3d} K={K:4d} M={M:3d} C1={6*N+3*M+1:4d} rows={t['rows_stacked']:6d}: " + " | ".join(row), flush=True)
