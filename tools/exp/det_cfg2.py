"""Round 6 debugging aid: repeated updates of BASELINE config 2 on one handle (upload_P + xk_visual_update_staged): is every posterior
bit-identical to the second update's, and if not what else differs (gate results, correction)."""
import sys, os
sys.path.insert(0, '.')
import numpy as np
from x_multi_agent_amd import engine, synth
from oracle import c_oracle
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
N, K, M = synth.CONFIGS[cfg]
sc = synth.make_config(cfg)
ref = c_oracle.visual_update(sc)
variants = (("adaptive", {}, None), ("152 first, then adaptive", {"pipe_split": 3}, ("pipe_split", 1)), ("184 twice, then adaptive", {"pipe_split": 0}, ("pipe_split", 1)))
if os.environ.get("XK_DET_STRICT") or os.environ.get("XK_DET_ONE"): variants = variants[:1]
for name, opts, sw in variants:
    eng = engine.Engine(N, M, K, lib_path=engine.STRICT_LIB_PATH) if os.environ.get("XK_DET_STRICT") else engine.LabEngine(N, M, K)
    for k, v in opts.items(): eng.set_option(k, v)
    eng.stage(sc)
    out = []
    for i in range(24):
        if sw and i == 2: eng.set_option(*sw)
        eng.upload_P(sc["P"])
        r = eng.visual_update_staged(sc["sigma_img"])
        P = eng.download_P()
        d = np.linalg.norm(P - ref["P"]) / np.linalg.norm(ref["P"])
        dc = np.linalg.norm(r["correction"] - ref["correction"]) / np.linalg.norm(ref["correction"])
        m = int(np.array_equal(r["inlier"], ref["inlier"]))
        out.append(f"{d:.0e}/{dc:.0e}/{m}")
    print(name, " ".join(out), eng.caqr_status(), flush=True)
    eng.close()
