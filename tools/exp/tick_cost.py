"""Where a request/response tick (config 5) spends its time on one agent: synthetic descriptors (bench overhead), keyframe insert,
VLAD, search, copy, CI round."""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
import os as _os; _os.environ.setdefault("XK_LIB_PATH", _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "..", "..", "x_multi_agent_amd", "lab", "libxk.so"))   # the lab build: env switches, hooks, probes (include/xk_lab.h)
from x_multi_agent_amd import engine, fleet, synth, place
N, K, M = synth.CONFIGS[5]
sc = fleet.shared_scenario(synth, 5, 0); sc1 = fleet.shared_scenario(synth, 5, 1)
eng = engine.Engine(N, M, K); eng.stage(sc); eng.run_steps(sc["sigma_img"], 3); eng.snapshot_P()
pay_n = eng.payload_doubles(); trk_n = 2 * (1 + 2 * N)
kdb = place.Database(eng, place.load_vocabulary("visual"), 0.6, payload_doubles=pay_n, tracks_doubles=trk_n, max_desc=256)
scene = synth.make_descriptors(96, 32, seed=0x5EED)
pay = torch.zeros(pay_n, dtype=torch.float64, device="cuda"); trk = torch.from_numpy(fleet.pack_tracks(sc, 2, N).ravel()).cuda()
resp = torch.zeros(2 + pay_n + trk_n, dtype=torch.float64, device="cuda")
dyn = np.zeros(16); dyn[9] = 1
e1 = engine.Engine(N, M, K); e1.stage(sc1); p1 = torch.zeros(pay_n, dtype=torch.float64, device="cuda"); e1.pack_payload_into(1, 0.0, dyn, p1.data_ptr())
t1 = torch.from_numpy(fleet.pack_tracks(sc1, 2, N).ravel()).cuda()
T = {k: [] for k in ("synth descriptors (bench)", "pack payload", "add_keyframe", "compute_vlad", "find_candidate", "copy_keyframe", "ci_round (2 agents)", "restore")}
for tick in range(30):
    t = time.perf_counter(); d1 = synth.observe_descriptors(scene, 4, seed=7919 + tick); d2 = synth.observe_descriptors(scene, 4, seed=104729 + tick); T["synth descriptors (bench)"].append(time.perf_counter() - t)
    t = time.perf_counter(); eng.pack_payload_into(0, float(tick), dyn, pay.data_ptr()); T["pack payload"].append(time.perf_counter() - t)
    t = time.perf_counter(); kdb.add_keyframe(d1, pay.data_ptr(), trk.data_ptr(), tag=tick); T["add_keyframe"].append(time.perf_counter() - t)
    t = time.perf_counter(); v = kdb.compute_vlad(d2); T["compute_vlad"].append(time.perf_counter() - t)
    t = time.perf_counter(); idx, score, tag = kdb.find_candidate(1, v); T["find_candidate"].append(time.perf_counter() - t)
    t = time.perf_counter()
    if idx >= 0: kdb.copy_keyframe(idx, resp.data_ptr() + 16, resp.data_ptr() + 8 * (2 + pay_n))
    torch.cuda.synchronize(); T["copy_keyframe"].append(time.perf_counter() - t)
    allp = torch.stack([pay, p1]); allt = torch.stack([trk, t1])
    t = time.perf_counter(); fused, _ = fleet.ci_round_device(eng, sc, 0, 2, allp, allt, 2, 0.05); T["ci_round (2 agents)"].append(time.perf_counter() - t)
    t = time.perf_counter(); eng.snapshot_P(restore=True); T["restore"].append(time.perf_counter() - t)
for k, v in T.items(): print(f"{k:28s} {1e3*np.median(v[5:]):.3f} ms")
print("candidate", idx, "fused", fused)
