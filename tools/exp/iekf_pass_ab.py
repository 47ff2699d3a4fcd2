"""One IEKF pass (correction_total != 0) at a config: queued whole (xk_build_compress_update_pass_async: the Kalman role inside the
single launch) against the two-call form (xk_build_compress_async, then xk_apply_update queues the separate Kalman launches).
    python tools/exp/iekf_pass_ab.py [config=4] [passes=300]"""
import ctypes as C, sys, time
sys.path.insert(0, '.')
import numpy as np
from x_multi_agent_amd import engine, synth
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 4
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
N, K, M = synth.CONFIGS[cfg]
sc = synth.make_config(cfg)
eng = engine.Engine(N, M, K)
eng.stage(sc)
eng.snapshot_P(0) if hasattr(eng, "snapshot_P") else None
ct = 1e-3 * np.random.default_rng(5).standard_normal(eng.n)
out = {}
eng.close()
eng = engine.LabEngine(N, M, K)       # (the two-call form of today defers the compression and fuses as well: "pipe_kalman" 0 = as before round 5)
eng.stage(sc)
for form in ("pass", "two-call"):
    eng.set_option("pipe_kalman", 1 if form == "pass" else 0)
    for cov in (0, 1):
        ts = []
        for rep in range(reps + 20):
            eng.stage(sc) if rep == 0 else None
            t0 = time.perf_counter()
            if form == "pass":
                eng.build_compress_update_pass_async(sc["sigma_img"], ct, cov)
            else:
                assert eng.L.xk_build_compress_async(eng.h, C.c_double(sc["sigma_img"])) == 0
            eng.apply_update(ct, cov)
            ts.append(time.perf_counter() - t0)
        out[(form, cov)] = 1e3 * float(np.median(ts[20:]))
for k, v in out.items():
    print("config %d  %-8s cov_update %d: %.4f ms per pass (host clock, median of %d)" % (cfg, k[0], k[1], v, reps), flush=True)
eng.close()
