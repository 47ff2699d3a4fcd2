"""A/B of the two narrow geometries of the single launch on one box: 184 tiles with one first-level group per XCD ("pipe_split" 0)
against 152 tiles with two ("pipe_split" 2), device-only replay of `steps` updates + the per-stage HIP-event split.
    python tools/exp/split_ab.py [config=4] [steps=500] [nominal]"""
import sys, time
sys.path.insert(0, '.')
import numpy as np
from x_multi_agent_amd import engine, synth
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 4
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 500
kw = dict(err_scale=0.3, outlier_frac=0.0) if len(sys.argv) > 3 else {}
N, K, M = synth.CONFIGS[cfg]
sc = synth.make_config(cfg, **kw)
for rep in range(2):
    for mode in (0, 2, 1):
        eng = engine.LabEngine(N, M, K)
        eng.set_option("pipe_split", mode)
        eng.stage(sc)
        eng.run_steps(sc["sigma_img"], 50)
        t0 = time.perf_counter()
        eng.run_steps(sc["sigma_img"], steps)
        dt = time.perf_counter() - t0
        t = eng.bench_staged(sc["sigma_img"], 2, 20)
        print(f"pipe_split={mode}: {steps / dt:8.1f} updates/s  {1e3 * dt / steps:.4f} ms  leaves {t['n_leaf']}  rows {t['rows_stacked']}  "
              f"stages {{{', '.join(f'{k}: {v['ms']:.4f}' for k, v in t['stages'].items() if v['launches'])}}}  status {eng.caqr_status()}", flush=True)
        eng.close()
