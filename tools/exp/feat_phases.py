import ctypes as C, os, sys
sys.path.insert(0, '.')
os.environ["XK_LIB_PATH"] = "tools/exp/bin/libxk_featprobe.so"
import numpy as np
from x_multi_agent_amd import engine, synth
sc = synth.make_config(4)
eng = engine.Engine(30, 0, 400)
eng.stage(sc)
eng.msckf_build(sc["sigma_img"])
out = (C.c_longlong * 8)()
for rep in range(2):
    eng.L.xk_debug_feature_phases(eng.h, C.c_double(sc["sigma_img"]), out)
t = np.array(list(out), dtype=np.int64)
names = ["tri(DLT+GN)", "jacobians", "hf_qr", "M build", "2-sided", "cholesky", "up/flag", "tile write"]
d = np.diff(t)
for n, v in zip(names, d): print(f"{n:12s} {v:8d} ticks  {v/2.4e3:7.2f} us")
print("total", (t[-1]-t[0])/2.4e3, "us")
