import sys
sys.path.insert(0, '.')
import numpy as np
import os as _os; _os.environ.setdefault("XK_LIB_PATH", _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "..", "..", "x_multi_agent_amd", "lab", "libxk.so"))   # the lab build: env switches, hooks, probes (include/xk_lab.h)
from x_multi_agent_amd import engine, synth
for N, K in ((33, 400), (34, 200), (34, 400), (33, 200)):
    sc = synth.make_scenario(N, K, 0, seed=99)
    eng = engine.Engine(N, 0, K)
    eng.stage(sc)
    tm = eng.bench_staged(sc["sigma_img"], 3, 20)
    st = tm["stages"]
    print(N, K, "rows/tile", 2 * N - 3, "total", round(tm["total_ms"], 4), {k: round(v["ms"], 4) for k, v in st.items() if v["launches"]})
    eng.close()
