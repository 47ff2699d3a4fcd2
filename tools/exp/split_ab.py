"""A/B of the two narrow geometries of the single launch on one box: 184 tiles with one first-level group per XCD ("pipe_split" 0)
against 152 tiles with two ("pipe_split" 3 = always, 1 = the adaptive default), device-only replay of `steps` updates + the per-stage HIP-event split.
    python tools/exp/split_ab.py [config=4] [steps=500] [nominal]"""
import os, sys, time
sys.path.insert(0, '.')
import numpy as np
from x_multi_agent_amd import engine, synth
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 4
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 500
kw = dict(err_scale=0.3, outlier_frac=0.0) if len(sys.argv) > 3 else {}
N, K, M = synth.CONFIGS[cfg]
sc = synth.make_config(cfg, **kw)
for rep in range(2):
    for mode in (0, 3, 1):
        eng = engine.Engine(N, M, K, lib_path=os.environ.get("XK_AB_LIB", engine.LAB_LIB_PATH))    # (a lab build: "pipe_split" is a lab option)
        eng.set_option("pipe_split", mode)
        eng.stage(sc)
        eng.run_steps(sc["sigma_img"], 50)
        t0 = time.perf_counter()
        eng.run_steps(sc["sigma_img"], steps)
        dt = time.perf_counter() - t0
        t = eng.bench_staged(sc["sigma_img"], 2, 20)
        stages = ", ".join("%s: %.4f" % (k, v["ms"]) for k, v in t["stages"].items() if v["launches"])
        print("pipe_split=%d: %8.1f updates/s  %.4f ms  leaves %d  rows %d  stages {%s}  status %s"
              % (mode, steps / dt, 1e3 * dt / steps, t["n_leaf"], t["rows_stacked"], stages, eng.caqr_status()), flush=True)
        eng.close()
