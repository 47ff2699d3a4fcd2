"""Soak of the register-resident single-launch CAQR: random shapes inside its window (MSCKF tracks only, 512 <= rows <=
23 808, <= 192 columns) against the C oracle, several updates per handle, plus repeated headline updates on one handle
(the double-buffered sync words alternate; a launch that gives up would show as an error string).
    python tools/exp/resident_soak.py [n_cases] [seed]"""
import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from x_multi_agent_amd import engine, synth
from oracle import c_oracle
from helpers import rel
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 80
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
worst, bad, took, t0 = 0.0, [], 0, time.time()
for it in range(n_cases):
    N = int(rng.integers(6, 32))
    K = int(rng.integers(12, 401))
    kw = dict(seed=int(rng.integers(1, 1 << 30)), outlier_frac=float(rng.choice([0.0, 0.05, 0.3, 0.8])),
              prior_scale=float(rng.choice([0.1, 1.0, 30.0])))
    if rng.random() < 0.5:
        kw["track_len"] = (2, N)
    if rng.random() < 0.25:
        kw["n_poses"] = int(rng.integers(max(3, N // 2), N))
        if "track_len" in kw:
            kw["track_len"] = (2, kw["n_poses"])
    try:
        sc = synth.make_scenario(N, K, 0, **kw)
    except Exception:
        continue
    ref = c_oracle.visual_update(sc)
    eng = engine.Engine(N, 0, K)
    for rep in range(3):
        eng.stage(sc)
        got = eng.visual_update_staged(sc["sigma_img"])
        P = eng.download_P()
        ok = np.array_equal(got["inlier"], ref["inlier"])
        rp = rel(P, ref["P"])
        worst = max(worst, rp)
        if not ok or not (rp <= 1e-8):
            bad.append((it, rep, N, K, kw, ok, rp))
    eng.stage(sc)
    t = eng.bench_staged(sc["sigma_img"], 0, 1)
    took += int(t["n_levels"] == 1)
    eng.close()
sc = synth.make_config(4)
ref = c_oracle.visual_update(sc)
eng = engine.Engine(30, 0, 400)
for rep in range(200):
    eng.stage(sc)
    got = eng.visual_update_staged(sc["sigma_img"])
    if rep % 50 == 0:
        rp = rel(eng.download_P(), ref["P"])
        worst = max(worst, rp)
        if not (rp <= 1e-8): bad.append(("headline", rep, rp))
eng.stage(sc)
t = eng.bench_staged(sc["sigma_img"], 0, 1)
print(f"{n_cases} random cases ({took} took the resident path), 200 headline updates on one handle (still resident: {t['n_levels'] == 1}); "
      f"worst rel dP {worst:.2e}; failures: {bad if bad else 'none'}; {time.time() - t0:.0f} s")
