import sys
sys.path.insert(0, '.')
import numpy as np, torch
import os as _os; _os.environ.setdefault("XK_LIB_PATH", _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "..", "..", "x_multi_agent_amd", "lab", "libxk.so"))   # the lab build: env switches, hooks, probes (include/xk_lab.h)
from x_multi_agent_amd import engine, fleet, synth
N, K, M = synth.CONFIGS[4]
scs = [fleet.shared_scenario(synth, 4, r) for r in range(8)]
eng = engine.Engine(N, M, K)
eng.stage(scs[0]); eng.run_steps(scs[0]["sigma_img"], 2)
dyn = np.zeros(16); dyn[9] = 1
pays = np.stack([fleet.pack_payload_host(r, 0.0, dyn, scs[r]["C_q_G"], scs[r]["G_p_C"], None, None, scs[r]["P"], N, M) for r in range(8)])
trks = np.stack([fleet.pack_tracks(scs[r], 2, N).ravel() for r in range(8)])
dev = torch.from_numpy(pays).cuda(); tdev = torch.from_numpy(trks).cuda()
for rep in range(3):
    eng.stage(scs[0]); torch.cuda.synchronize()
    fused, _ = fleet.ci_round_device(eng, scs[0], 0, 8, dev, tdev, 2, 0.05)
torch.cuda.synchronize()
