"""Round 6: tall systems (BASELINE config 3) with and without the single-launch TAIL (xk_caqr_pipe<XkPipeTail>: the last <= 96 columns in one
launch, every row in registers): parity against the C oracle, stage times, replay rate.   python tools/exp/tail_ab.py [config] [steps]"""
import os, sys, time
sys.path.insert(0, '.')
import numpy as np
from x_multi_agent_amd import engine, synth
from oracle import c_oracle
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
N, K, M = synth.CONFIGS[cfg]
sc = synth.make_config(cfg)
ref = c_oracle.visual_update(sc)
for tail in (1, 2, 0):
    eng = engine.Engine(N, M, K)
    eng.set_option("caqr_tail", tail)
    got = eng.visual_update(sc)
    relP = np.linalg.norm(got["P"] - ref["P"]) / np.linalg.norm(ref["P"])
    relc = np.linalg.norm(got["correction"] - ref["correction"]) / np.linalg.norm(ref["correction"])
    same = bool(np.array_equal(got["inlier"], ref["inlier"]))
    st0 = eng.caqr_status()
    eng.stage(sc)
    t = eng.bench_staged(sc["sigma_img"], 3, 20)
    eng.stage(sc); eng.run_steps(sc["sigma_img"], 5)
    t0 = time.perf_counter(); eng.run_steps(sc["sigma_img"], steps); dt = (time.perf_counter() - t0) / steps
    print(f"config {cfg} tail={tail}: rel dP {relP:.2e} rel dcorr {relc:.2e} masks {same} | first update {st0} |",
          {k: round(v["ms"], 4) for k, v in t["stages"].items() if v["ms"] > 0}, "launches", t["n_levels"],
          f"| replay {1e3*dt:.4f} ms = {1/dt:.1f} updates/s", eng.caqr_status(), flush=True)
    eng.close()
