"""Per-block phase times of the Kalman role inside xk_caqr_pipe (XK_CAQR_PERSIST_DBG=1): python tools/exp/kal_trace.py [cfg]"""
import ctypes as C, os, sys
sys.path.insert(0, '.')
os.environ.setdefault("XK_CAQR_PERSIST_DBG", "1")
import numpy as np
import os as _os; _os.environ.setdefault("XK_LIB_PATH", _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "..", "..", "x_multi_agent_amd", "lab", "libxk.so"))   # the lab build: env switches, hooks, probes (include/xk_lab.h)
from x_multi_agent_amd import engine, synth
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 4
sc = synth.make_config(cfg)
N, K, M = synth.CONFIGS[cfg]
eng = engine.Engine(N, M, K)
eng.stage(sc)
t = eng.bench_staged(sc["sigma_img"], 3, 20)
print({k: round(v["ms"], 4) for k, v in t["stages"].items() if v["ms"] > 0}, "total", round(t["total_ms"], 4))
NW = 4096
out = (C.c_longlong * NW)()
eng.L.xk_debug_persist_stamps(eng.h, out, C.c_int(NW))
w = np.array(list(out), dtype=np.int64)
T = w[:512].reshape(32, 16); L = w[1024:1536].reshape(32, 16); F = w[2048:2560].reshape(32, 16); R = w[2560:3072].reshape(32, 16)
npan = (6 * N + 3 * M + 1 + 15) // 16
us = lambda x: x / 100.0
t0 = T[0, 0]
names = ["wait R", "T->LDS", "W loop", "part+S part", "bar1", "W12 (w11)", "S sum+chol", "bar3", "X", "update", "bar5"]
print("factor wave (wave 0): block | start after tile panel 0 began | last level of the panel out at | " + " | ".join(names))
for k in range(npan):
    f = F[k]
    if f[0] == 0: continue
    d = [us(f[i + 1] - f[i]) for i in range(10)]
    print(f"{k:3d} start {us(f[0]-t0):7.2f} R-out {us(L[k,8]-t0):7.2f}  " + " ".join(f"{x:6.2f}" for x in d) + f"  | block total {us(f[10]-f[0]):6.2f}")
print("tile wave 5: W loop end .. | same columns from stamp 2")
for k in range(npan):
    r = R[k]
    if r[2] == 0: continue
    print(f"{k:3d} " + " ".join(f"{us(r[i+1]-r[i]):6.2f}" for i in range(2, 10)))
print("end of the role, us after tile panel 0 began: last block done", us(F[31,1]-t0), "correction + marker out", us(F[31,2]-t0), "wave 0's stores issued", us(F[31,0]-t0), "thread 0 leaves the role", us(F[31,3]-t0))
print(f"kalman role done {us(F[31,0]-t0):.2f} us after tile panel 0 began; last level done {us(w[1539]-t0):.2f}")
eng.close()
