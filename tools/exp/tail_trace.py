"""Round 6: per-panel phase stamps of the TAIL launch of a tall system (xk_caqr_pipe<XkPipeTail>, XK_CAQR_PERSIST_DBG=1, lab build)."""
import ctypes as C, os, sys
sys.path.insert(0, '.')
os.environ.setdefault("XK_CAQR_PERSIST_DBG", "1")
import numpy as np
from x_multi_agent_amd import engine, synth
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
sc = synth.make_config(cfg)
N, K, MS = synth.CONFIGS[cfg]
eng = engine.LabEngine(N, MS, K)
eng.stage(sc)
t = eng.bench_staged(sc["sigma_img"], 3, 10)
print({k: round(v["ms"], 4) for k, v in t["stages"].items() if v["ms"] > 0}, "launches", t["n_levels"], eng.caqr_status())
NW = 2048
out = (C.c_longlong * NW)()
eng.L.xk_debug_persist_stamps(eng.h, out, C.c_int(NW))
w = np.array(list(out), dtype=np.int64)
T = w[:512].reshape(32, 16); M = w[512:1024].reshape(32, 16); L = w[1024:1536].reshape(32, 16)
C1 = 6 * N + 3 * MS + 1
ccut = 16 * ((C1 - int(os.environ.get("XK_TAIL_COLS", "192")) + 15) // 16)
npan = (C1 - ccut + 15) // 16
us = lambda x: x / 100.0
t0 = T[0, 0]
print("tile workgroup (XCD 0, slot 1), us: phase 0 | 1 | 2 | 3 | drain + arrive | wait for the strips | reload || panel total | cumulative")
for k in range(npan):
    r = T[k]
    nxt = T[k + 1, 0] if k + 1 < npan else r[5]
    ph = [r[0]] + [x for x in r[1:5] if x > 0]
    print(f"{k:3d}  " + " ".join(f"{us(ph[i+1]-ph[i]):5.2f}" for i in range(len(ph) - 1)) + f" | {us(r[5]-ph[-1]):5.2f} {us(r[6]-r[5]):5.2f} {us(r[7]-r[6]):5.2f} || {us(nxt-r[0]):6.2f} | {us(nxt-t0):7.2f}")
print("first level (XCD 0, item 0), us after the tile step of the panel started")
for k in range(npan):
    print(f"{k:3d}  " + "  ".join(f"{us(M[k,i]-T[k,0]):6.2f}" for i in range(9) if M[k, i] > 0))
print("last level (workgroup 0), us after the tile step of the panel started")
for k in range(npan):
    print(f"{k:3d}  " + "  ".join(f"{us(L[k,i]-T[k,0]):6.2f}" for i in range(9) if L[k, i] > 0))
print(f"start-up: entry -> rows gathered {us(w[1537]-w[1536]):.2f} us; tile workgroup leaves {us(w[1538]-w[1536]):.2f} us after its entry, last level {us(w[1539]-w[1536]):.2f} us")
eng.close()
