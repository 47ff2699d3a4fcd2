"""Round 6: SPLIT compression of systems with SLAM features (xk_handle::d_R2: only the tracks' rows are compressed, in the pose columns;
the features' own 2 M rows join the compressed system as built) against the compression of the whole stack: parity against the C oracle,
stage times, replay rate.   python tools/exp/slam_split_ab.py [config] [steps]"""
import os, sys, time
sys.path.insert(0, '.')
import numpy as np
from x_multi_agent_amd import engine, synth
from oracle import c_oracle
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
N, K, M = synth.CONFIGS[cfg]
sc = synth.make_config(cfg)
ref = c_oracle.visual_update(sc)
for split in (1, 0, 1):
    eng = engine.Engine(N, M, K)
    eng.set_option("slam_split", split)
    eng.stage(sc)
    got = eng.visual_update_staged(sc["sigma_img"])
    P = eng.download_P()
    relP = np.linalg.norm(P - ref["P"]) / np.linalg.norm(ref["P"])
    relc = np.linalg.norm(got["correction"] - ref["correction"]) / np.linalg.norm(ref["correction"])
    same = bool(np.array_equal(got["inlier"], ref["inlier"]) and np.array_equal(got["inlier_slam"], ref["inlier_slam"]))
    eng.stage(sc)
    t = eng.bench_staged(sc["sigma_img"], 3, 30)
    eng.stage(sc); eng.run_steps(sc["sigma_img"], 10)
    t0 = time.perf_counter(); eng.run_steps(sc["sigma_img"], steps); dt = (time.perf_counter() - t0) / steps
    print(f"config {cfg} slam_split={split}: rel dP {relP:.2e} rel dcorr {relc:.2e} masks {same} |",
          {k: round(v["ms"], 4) for k, v in t["stages"].items() if v["ms"] > 0}, "leaves", t["n_leaf"],
          f"| replay {1e3*dt:.4f} ms = {1/dt:.1f} updates/s", eng.caqr_status(), flush=True)
    eng.close()
