"""The register-resident single-launch CAQR (XK_CAQR_RESIDENT=1) at a given config: parity against the multi-launch
schedule, stage times, per-panel phase spans of one tile workgroup (XK_CAQR_PERSIST_DBG=1)."""
import ctypes as C, os, sys, subprocess
sys.path.insert(0, '.')
os.environ.setdefault("XK_CAQR_PERSIST_DBG", "1")
import numpy as np
from x_multi_agent_amd import engine, synth

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 4
sc = synth.make_config(cfg)
N = sc["n_poses_max"]; K = len(sc["trk_off"]) - 1
eng = engine.Engine(N, 0, K)
got = eng.visual_update(sc)
t = eng.bench_staged(sc["sigma_img"], 3, 20)
mode = "multi-launch" if os.environ.get("XK_CAQR_RESIDENT") == "0" else ("resident+blocked" if os.environ.get("XK_CAQR_BLOCKED") == "1" else "resident")
print(mode + ":", {k: round(v["ms"], 4) for k, v in t["stages"].items() if v["ms"] > 0}, "total", round(t["total_ms"], 4), "launches", t["n_levels"])
if mode != "multi-launch":
    NW = 1024
    out = (C.c_longlong * NW)()
    eng.L.xk_debug_persist_stamps(eng.h, out, C.c_int(NW))
    allw = np.array(list(out), dtype=np.int64)
    w = allw[:256].reshape(32, 8)
    w2 = allw[256:512].reshape(32, 8)
    w3 = allw[512:768].reshape(32, 8)
    npan = (6 * N + 1 + 15) // 16
    t0 = w[0, 0]
    print("panel: tile  bar1  wait  merge bar2+reload | cum us | last-level: start(after tile start) span")
    for k in range(npan):
        r = w[k]
        print(f"{k:3d}  {(r[1]-r[0])/100:5.2f} {(r[2]-r[1])/100:5.2f} {(r[3]-r[2])/100:5.2f} {(r[4]-r[3])/100:5.2f} {(r[5]-r[4])/100:5.2f} | {(r[5]-t0)/100:7.2f} | {(r[6]-r[0])/100:6.2f} {(r[7]-r[6])/100:6.2f} | steps only {(w2[k,1]-w2[k,0])/100:5.2f} | merge: load {(w2[k,2]-r[3])/100:5.2f} steps {(w2[k,3]-w2[k,2])/100:5.2f} ({(w2[k,5]-w2[k,4])/max(1,(w2[k,3]-w2[k,2]))/10:.2f} GHz) rest {(r[4]-w2[k,3])/100:5.2f}")
    w4 = allw[768:1024].reshape(32, 8)
    print("last level (workgroup 0), us after the tile step of the same panel started: rows 0..7 flagged | steps 0..7 done | rows 8..15 flagged | steps 8..15 done | strips published")
    for k in range(npan):
        print(f"{k:3d}  " + "  ".join(f"{(w4[k,i]-w[k,0])/100:6.2f}" for i in range(4)) + f"  {(w[k,7]-w[k,0])/100:6.2f}")
    print(f"start-up: entry -> census done {(allw[513]-allw[512])/100:.2f} us, -> rows gathered (panel 0 starts) {(w[0,0]-allw[513])/100:.2f} us; "
          f"tile workgroup leaves {(allw[514]-allw[512])/100:.2f} us after its entry, last-level workgroup {(allw[515]-allw[512])/100:.2f} us")
    np.save("/tmp/res_P.npy", got["P"]); np.save("/tmp/res_c.npy", got["correction"])
    eng.close()
    env = dict(os.environ, XK_CAQR_RESIDENT="0")
    print(subprocess.run([sys.executable, __file__, str(cfg)], env=env, capture_output=True, text=True).stdout.strip().splitlines()[-1])
else:
    P = np.load("/tmp/res_P.npy"); c = np.load("/tmp/res_c.npy")
    print(f"resident vs multi-launch: rel dP {np.linalg.norm(P-got['P'])/np.linalg.norm(got['P']):.2e} rel dcorr {np.linalg.norm(c-got['correction'])/np.linalg.norm(got['correction']):.2e}")
