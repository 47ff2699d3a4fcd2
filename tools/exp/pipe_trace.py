"""The pipelined register-resident CAQR (xk_caqr_pipe) at a given config: parity against the multi-launch
schedule, stage times, per-panel phase spans of one tile / first-level / last-level
workgroup (XK_CAQR_PERSIST_DBG=1)."""
import ctypes as C, os, sys, subprocess
sys.path.insert(0, '.')
os.environ.setdefault("XK_CAQR_PERSIST_DBG", "1")
import numpy as np
import os as _os; _os.environ.setdefault("XK_LIB_PATH", _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "..", "..", "x_multi_agent_amd", "lab", "libxk.so"))   # the lab build: env switches, hooks, probes (include/xk_lab.h)
from x_multi_agent_amd import engine, synth

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 4
mode = sys.argv[2] if len(sys.argv) > 2 else "pipe"
if mode == "multi": os.environ["XK_CAQR_RESIDENT"] = "0"
sc = synth.make_config(cfg)
N = int(sc["n_poses_max"]); K = len(sc["trk_off"]) - 1
MS = int(len(sc['slam_anchor_idxs'])) if 'slam_anchor_idxs' in sc else 0
eng = engine.Engine(N, MS, K)
got = eng.visual_update(sc)
t = eng.bench_staged(sc["sigma_img"], 3, 20)
print(mode + ":", {k: round(v["ms"], 4) for k, v in t["stages"].items() if v["ms"] > 0}, "total", round(t["total_ms"], 4), "launches", t["n_levels"], "leaves", t["n_leaf"], flush=True)
if mode == "pipe":
    NW = 2048
    out = (C.c_longlong * NW)()
    eng.L.xk_debug_persist_stamps(eng.h, out, C.c_int(NW))
    w = np.array(list(out), dtype=np.int64)
    T = w[:512].reshape(32, 16); M = w[512:1024].reshape(32, 16); L = w[1024:1536].reshape(32, 16)
    npan = int((6 * N + 3 * MS + 1 + 15) // 16)
    t0 = T[0, 0]
    us = lambda x: x / 100.0
    print("tile workgroup (XCD 0, slot 1), us: phase 0 | 1 | 2 | 3 (steps + rows stored) | drain + arrive | wait for the strips | reload || panel total | cumulative")
    for k in range(npan):
        r = T[k]
        nxt = T[k + 1, 0] if k + 1 < npan else r[5]
        ph = [r[0]] + [x for x in r[1:5] if x > 0]
        print(f"{k:3d}  " + " ".join(f"{us(ph[i+1]-ph[i]):5.2f}" for i in range(len(ph) - 1)) + f" | {us(r[5]-ph[-1]):5.2f} {us(r[6]-r[5]):5.2f} {us(r[7]-r[6]):5.2f} || {us(nxt-r[0]):6.2f} | {us(nxt-t0):7.2f} | shader clock {(r[9]-r[8])/max(1,(r[10]-r[0]))/10:.2f} GHz")
    print("first level (XCD 0, item 0), us after the tile step of the panel started: per phase (rows seen, steps done + published) ... | strips stored + arrived")
    for k in range(npan):
        print(f"{k:3d}  " + "  ".join(f"{us(M[k,i]-T[k,0]):6.2f}" for i in range(9) if M[k, i] > 0) +
              ("   | rows landed (probe build): " + "  ".join(f"{us(M[k,i]-T[k,0]):6.2f}" for i in range(10, 14) if M[k, i] > 0) if M[k, 10] > 0 else ""))
    print("last level (workgroup 0), us after the tile step of the panel started: per phase (roots seen, steps done) ... | out")
    for k in range(npan):
        print(f"{k:3d}  " + "  ".join(f"{us(L[k,i]-T[k,0]):6.2f}" for i in range(9) if L[k, i] > 0))
    print(f"start-up: entry -> rows gathered {us(w[1537]-w[1536]):.2f} us; tile workgroup leaves {us(w[1538]-w[1536]):.2f} us after its entry, last level {us(w[1539]-w[1536]):.2f} us")
    if w[1544] > 0:
        print("gather from factor records, us after entry: start %.2f, slot range %.2f, records staged %.2f, blocks done %s" % (
            us(w[1544]-w[1536]), us(w[1545]-w[1536]), us(w[1546]-w[1536]), " ".join("%.2f" % us(x-w[1536]) for x in w[1547:1552] if x > 0)))
    np.save("/tmp/pipe_P.npy", got["P"]); np.save("/tmp/pipe_c.npy", got["correction"])
    eng.close()
    for m in ("multi",):
        r = subprocess.run([sys.executable, __file__, str(cfg), m], capture_output=True, text=True)
        print((r.stdout.strip() or r.stderr.strip()[-400:]))
else:
    P = np.load("/tmp/pipe_P.npy"); c = np.load("/tmp/pipe_c.npy")
    print(f"   pipe vs {mode}: rel dP {np.linalg.norm(P-got['P'])/np.linalg.norm(got['P']):.2e} rel dcorr {np.linalg.norm(c-got['correction'])/np.linalg.norm(got['correction']):.2e}")
