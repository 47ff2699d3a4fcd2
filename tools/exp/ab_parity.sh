#!/bin/bash
# parity of an alternative build of libxk.so against the C oracle on a few shapes: bash tools/exp/ab_parity.sh lib.so
XK_LIB_PATH=$1 python - <<'PY'
import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from x_multi_agent_amd import engine, synth
from oracle import c_oracle
def rel(a, b): return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)
c_oracle.build()
for name, (N, K, M), sc in (("cfg4", synth.CONFIGS[4], synth.make_config(4)), ("cfg2", synth.CONFIGS[2], synth.make_config(2)),
                 ("large_prior", (30, 200, 0), synth.make_scenario(30, 200, 0, seed=904, prior_scale=100.0)),
                 ("huge_prior", (30, 200, 0), synth.make_scenario(30, 200, 0, seed=904, prior_scale=1e4))):
    ref = c_oracle.visual_update(sc)
    eng = engine.Engine(N, M, K)
    got = eng.visual_update(sc)
    print(f"{name:12s} rel dP {rel(got['P'], ref['P']):.2e}  rel dcorr {rel(got['correction'], ref['correction']):.2e}  masks {np.array_equal(got['inlier'], ref['inlier'])}  schedule {eng.caqr_status()['schedule']}")
    eng.close()
PY
