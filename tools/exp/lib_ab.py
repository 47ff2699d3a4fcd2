"""One library build, one config: device-only replay rate + stage split (for A/B of compile-time variants on one box).
    XK_AB_LIB=tools/exp/bin/libxk_X.so python tools/exp/lib_ab.py [config=4] [steps=400] [pipe_split=1]"""
import os, sys, time
sys.path.insert(0, '.')
from x_multi_agent_amd import engine, synth
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 4
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 400
mode = int(sys.argv[3]) if len(sys.argv) > 3 else 1
N, K, M = synth.CONFIGS[cfg]
sc = synth.make_config(cfg)
lib = os.environ.get("XK_AB_LIB", engine.LAB_LIB_PATH)
best = None
for rep in range(3):
    eng = engine.Engine(N, M, K, lib_path=lib)
    eng.set_option("pipe_split", mode)
    if os.environ.get("XK_AB_HLITE") is not None:
        eng.set_option("caqr_hlite", int(os.environ["XK_AB_HLITE"]))    # 0: tiles in HBM (as before round 5), 1: factor records
    eng.stage(sc)
    eng.run_steps(sc["sigma_img"], 50)
    t0 = time.perf_counter()
    eng.run_steps(sc["sigma_img"], steps)
    dt = time.perf_counter() - t0
    t = eng.bench_staged(sc["sigma_img"], 2, 20)
    stages = ", ".join("%s: %.4f" % (k, v["ms"]) for k, v in t["stages"].items() if v["launches"])
    line = "%-28s cfg %d split %d: %8.1f updates/s  %.4f ms  leaves %d  {%s} giveups %d" % (
        os.path.basename(lib), cfg, mode, steps / dt, 1e3 * dt / steps, t["n_leaf"], stages, eng.caqr_status()["giveups"])
    if best is None or dt < best[0]:
        best = (dt, line)
    eng.close()
print(best[1], flush=True)
