"""Cost of one CI round on one agent as a function of the number of other agents (single GPU, no transport):
the device -> host copy of the gathered payloads, the unpack, and fleet.ci_round."""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
import os as _os; _os.environ.setdefault("XK_LIB_PATH", _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "..", "..", "x_multi_agent_amd", "lab", "libxk.so"))   # the lab build: env switches, hooks, probes (include/xk_lab.h)
from x_multi_agent_amd import engine, fleet, synth
N, K, M = synth.CONFIGS[4]
scs = [fleet.shared_scenario(synth, 4, r) for r in range(8)]
eng = engine.Engine(N, M, K)
eng.stage(scs[0])
eng.run_steps(scs[0]["sigma_img"], 3)
lay = fleet.payload_layout(N, M)
dyn = np.zeros(16); dyn[9] = 1
pays = np.stack([fleet.pack_payload_host(r, 0.0, dyn, scs[r]["C_q_G"], scs[r]["G_p_C"], None, None, scs[r]["P"], N, M) for r in range(8)])
trks = np.stack([fleet.pack_tracks(scs[r], 2, N).ravel() for r in range(8)])
for world in (2, 4, 8):
    dev = torch.from_numpy(pays[:world].copy()).cuda()
    tdev = torch.from_numpy(trks[:world].copy()).cuda()
    torch.cuda.synchronize()
    ts = []
    for rep in range(5):
        t0 = time.perf_counter()
        allp = dev.cpu().numpy(); allt = tdev.cpu().numpy()
        t1 = time.perf_counter()
        others = []
        for r in range(1, world):
            u = fleet.unpack_payload(allp[r], N, M); u["tracks"] = fleet.unpack_tracks(allt[r], N); others.append(u)
        t2 = time.perf_counter()
        fused, _ = fleet.ci_round(eng, scs[0], others, 2, 0.05)
        t3 = time.perf_counter()
        ts.append((t1 - t0, t2 - t1, t3 - t2))
    a = np.median(np.array(ts), axis=0) * 1e3
    print(f"world {world}: d2h {a[0]:.2f} ms, unpack {a[1]:.2f} ms, ci_round {a[2]:.2f} ms (fused {fused}) -> {a.sum():.2f} ms per round")
print("device-resident round (payloads stay in HBM, agents batched):")
for world in (2, 4, 8):
    dev = torch.from_numpy(pays[:world].copy()).cuda()
    tdev = torch.from_numpy(trks[:world].copy()).cuda()
    torch.cuda.synchronize()
    ts = []
    for rep in range(20):
        eng.stage(scs[0]); torch.cuda.synchronize()
        t0 = time.perf_counter()
        fused, _ = fleet.ci_round_device(eng, scs[0], 0, world, dev, tdev, 2, 0.05)
        ts.append(time.perf_counter() - t0)
    print(f"world {world}: {np.median(ts) * 1e3:.3f} ms per round, min {min(ts) * 1e3:.3f} (fused {fused})")
