"""One headline-size update through the single-launch CAQR (xk_caqr_persist) with its phase stamps, next to the
multi-launch schedule: parity of the two against the C oracle's compressed system, stage times, per-panel phase spans.

    XK_CAQR_PERSIST_DBG=1 python tools/exp/persist_trace.py [config]
"""
import ctypes as C, os, sys, subprocess
sys.path.insert(0, '.')
os.environ.setdefault("XK_CAQR_PERSIST_DBG", "1")
import numpy as np
from x_multi_agent_amd import engine, synth

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 4
sc = synth.make_config(cfg)
N = sc["n_poses_max"]; K = len(sc["trk_off"]) - 1
M = len(sc["slam_anchor_idxs"]) if "slam_anchor_idxs" in sc else 0
eng = engine.Engine(N, M, K)
got = eng.visual_update(sc)
t = eng.bench_staged(sc["sigma_img"], 3, 20)
print("persist:", {k: round(v["ms"], 4) for k, v in t["stages"].items() if v["ms"] > 0}, "total", round(t["total_ms"], 4), "launches", t["n_levels"])
NW = 256 + 64 * 256
out = (C.c_longlong * NW)()
rc = eng.L.xk_debug_persist_stamps(eng.h, out, C.c_int(NW))
allw = np.array(list(out), dtype=np.int64)
w = allw[:256].reshape(32, 8)
sl = allw[256:].reshape(64, 32, 8)
npan = (6 * N + 3 * M + 1 + 15) // 16
t0 = w[0, 0]
print("panel: tile  bar1  wait  merge bar2 | cum us | last-level: start(after tile start) span")
for k in range(npan):
    r = w[k]
    print(f"{k:3d}  {(r[1]-r[0])/100:5.2f} {(r[2]-r[1])/100:5.2f} {(r[3]-r[2])/100:5.2f} {(r[4]-r[3])/100:5.2f} {(r[5]-r[4])/100:5.2f} | {(r[5]-t0)/100:7.2f} | {(r[6]-r[0])/100:6.2f} {(r[7]-r[6])/100:6.2f}")
print(f"QR span (T workgroup 0): {(w[npan-1,5]-t0)/100:.1f} us; last level ends {(w[npan-1,7]-t0)/100:.1f} us")
for k in (1, 6):
    base = sl[:50, k, 0].min()
    print(f"panel {k}, XCD 0, all role-T slots (us after the first tile start): slot: tile_start tile_end bar1_exit merge_start merge_end bar2_exit")
    for s_ in range(50):
        r = (sl[s_, k, :6] - base) / 100.0
        print(f"  {s_:2d}: " + " ".join(f"{x:7.2f}" if sl[s_, k, i] else "      -" for i, x in enumerate(r)))
eng.close()
# the same update through the multi-launch schedule, in a fresh process (the switch is read once)
if os.environ.get("XK_CAQR_PERSIST") != "0":
    np.save("/tmp/persist_P.npy", got["P"]); np.save("/tmp/persist_c.npy", got["correction"])
    env = dict(os.environ, XK_CAQR_PERSIST="0")
    print(subprocess.run([sys.executable, __file__, str(cfg)], env=env, capture_output=True, text=True).stdout.strip().splitlines()[-1])
else:
    P = np.load("/tmp/persist_P.npy"); c = np.load("/tmp/persist_c.npy")
    print("multi-launch:", {k: round(v["ms"], 4) for k, v in t["stages"].items() if v["ms"] > 0}, "total", round(t["total_ms"], 4),
          f"| persist vs multi-launch: rel dP {np.linalg.norm(P-got['P'])/np.linalg.norm(got['P']):.2e} rel dcorr {np.linalg.norm(c-got['correction'])/np.linalg.norm(got['correction']):.2e}")
