"""Timing probes (builds that compute WRONG results on purpose): wall time of `steps` queued updates whatever their status."""
import os, sys, time
sys.path.insert(0, '.')
from x_multi_agent_amd import engine, synth
cfg, steps, mode = 4, 400, int(sys.argv[1]) if len(sys.argv) > 1 else 3
N, K, M = synth.CONFIGS[cfg]
sc = synth.make_config(cfg)
lib = os.environ.get("XK_AB_LIB", engine.LAB_LIB_PATH)
best = 1e9
for rep in range(3):
    eng = engine.Engine(N, M, K, lib_path=lib)
    eng.set_option("pipe_split", mode)
    eng.stage(sc)
    for n in (50, steps):
        t0 = time.perf_counter()
        try:
            eng.run_steps(sc["sigma_img"], n)
        except engine.XkError as e:
            err = str(e)[:60]
        dt = time.perf_counter() - t0
    best = min(best, dt)
    eng.close()
print("%-24s split %d: %.4f ms per update (timing probe)" % (os.path.basename(lib), mode, 1e3 * best / steps), flush=True)
