"""Start of the single launch, every workgroup (XK_CAQR_PERSIST_DBG stamps): when the tiles enter, when they start panel 0, when their
phase-0 rows are counted in, when the first level sees them, how long the idle workgroups take to re-arm the slabs.
    XK_HLITE=0|1 python tools/exp/tile_starts.py"""
import ctypes as C, os, sys
sys.path.insert(0, '.')
os.environ["XK_CAQR_PERSIST_DBG"] = "1"
import numpy as np
from x_multi_agent_amd import engine, synth
sc = synth.make_config(4)
N, K, M = synth.CONFIGS[4]
eng = engine.LabEngine(N, M, K)
eng.stage(sc)
t = eng.bench_staged(sc["sigma_img"], 3, 20)
NW = 65536
out = (C.c_longlong * NW)()
eng.L.xk_debug_persist_stamps(eng.h, out, C.c_int(NW))
w = np.array(list(out), dtype=np.int64)
s0 = w[40960:40960+152]; s1 = w[40960+256:40960+256+152]; e = w[40960+512:40960+512+152]
ok = s0 > 0
t0 = e[ok].min()
print("HLITE", os.environ.get("XK_HLITE"), "pipe ms", round(t["stages"]["xk_caqr_panel0"]["ms"], 4))
print("tile entry us after the first entry: min %.1f median %.1f max %.1f" % tuple(np.percentile((e[ok]-t0)/100.0, [0, 50, 100])))
d = (s0[ok]-e[ok])/100.0
print("start-up (entry -> panel 0 starts) per tile: min %.1f median %.1f p90 %.1f max %.1f" % tuple(np.percentile(d, [0, 50, 90, 100])))
print("slowest tiles:", np.argsort(-d)[:8].tolist(), np.sort(-d)[:8].round(1).tolist())
print("panel 0 duration per tile: median %.1f max %.1f" % (np.median((s1[ok]-s0[ok])/100.0), ((s1[ok]-s0[ok])/100.0).max()))
idle = w[45056:45056 + 8 * 32 * 2].reshape(8, 32, 2)
m = idle[:, :, 0] > 0
print("idle workgroups (first / last level, Kalman): entry after first tile entry min %.1f max %.1f; re-arm takes min %.1f median %.1f max %.1f us" % (
    ((idle[:, :, 0][m] - t0) / 100.0).min(), ((idle[:, :, 0][m] - t0) / 100.0).max(),
    *np.percentile((idle[:, :, 1][m] - idle[:, :, 0][m]) / 100.0, [0, 50, 100])))
seen = w[46080:46080 + 8 * 12]
arr = w[47104:47104 + 152]
print("tiles: phase-0 rows of panel 0 counted in, us after entry: min %.1f median %.1f max %.1f (slowest tiles %s)" % (
    *np.percentile((arr[ok] - t0) / 100.0, [0, 50, 100]), np.argsort(-arr)[:6].tolist()))
print("first level: sees them, us after entry, per workgroup:", np.round((seen[seen > 0] - t0) / 100.0, 1).tolist())
eng.close()
