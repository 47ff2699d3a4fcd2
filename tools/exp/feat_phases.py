"""Phase stamps, wall-clock interval and placement of every workgroup of xk_msckf_feature
(needs the probe build: hipcc ... -DXK_FEAT_PROBE -o tools/exp/bin/libxk_featprobe.so)."""
import ctypes as C, os, sys
sys.path.insert(0, '.')
os.environ["XK_LIB_PATH"] = "tools/exp/bin/libxk_featprobe.so"
import numpy as np
from x_multi_agent_amd import engine, synth
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 4
if os.environ.get("XK_FEAT_SHAPE"):      # e.g. XK_FEAT_SHAPE=30,8,4,12: window 30, 8 tracks of 4..12 observations (round 6: the small frames)
    N, K, lo, hi = (int(x) for x in os.environ["XK_FEAT_SHAPE"].split(","))
    M = 0
    sc = synth.make_scenario(N, K, 0, seed=4308, track_len=(lo, hi))
else:
    sc = synth.make_config(cfg)
    N, K, M = synth.CONFIGS[cfg]
eng = engine.Engine(N, M, K)
eng.stage(sc)
res = eng.msckf_build(sc["sigma_img"])
n = 12 * K
out = (C.c_longlong * n)()
for rep in range(3):
    eng.L.xk_debug_feature_phases(eng.h, C.c_double(sc["sigma_img"]), out, C.c_int(n))
w = np.array(list(out), dtype=np.int64).reshape(K, 12)
names = ["tri(DLT+GN)", "jacobians", "hf_qr", "M build", "2-sided", "cholesky", "up/flag", "tile write"]
inl = np.asarray(res["inlier"]).astype(bool)
ph = np.diff(w[:, :8], axis=1) / 2.4e3
ph[~inl, 6] = 0   # rejected tracks return before the tile write
st, en, hw = w[:, 8], w[:, 9], w[:, 10]
t0 = st.min()
dur = (en - st) / 100.0
slow = dur > 1.5 * np.median(dur)
print(f"{'phase (us)':14s} {'median':>8s} {'max':>8s} {'slow WGs mean':>14s}")
for i, nme in enumerate(names[:7]):
    print(f"{nme:14s} {np.median(ph[:, i]):8.2f} {ph[:, i].max():8.2f} {ph[slow, i].mean() if slow.any() else 0:14.2f}")
print(f"kernel span {(en.max()-t0)/100.0:.1f} us; WG duration min/median/max {dur.min():.1f}/{np.median(dur):.1f}/{dur.max():.1f} us; "
      f"last start {(st.max()-t0)/100.0:.1f} us; {slow.sum()} WGs > 100 us (inlier fraction {inl[slow].mean() if slow.any() else 0:.2f})")
print("duration histogram (us):", np.histogram(dur, bins=[0, 60, 70, 80, 90, 100, 110, 120, 130, 140, 200])[0])
xcc = (hw >> 32) & 0xF
hwid = hw & 0xFFFFFFFF
key = xcc * 100000 + ((hwid >> 8) & 0xFFF)
uniq, cnt = np.unique(key, return_counts=True)
print(f"{len(uniq)} distinct CUs; WGs per CU: {dict(zip(*[x.tolist() for x in np.unique(cnt, return_counts=True)]))}")
print("slow WG ids:", np.nonzero(slow)[0].tolist()[:32], "xcc:", xcc[slow].tolist()[:32])
