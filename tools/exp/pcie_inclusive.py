import sys, time; sys.path.insert(0,'.')
import numpy as np
import os as _os; _os.environ.setdefault("XK_LIB_PATH", _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "..", "..", "x_multi_agent_amd", "lab", "libxk.so"))   # the lab build: env switches, hooks, probes (include/xk_lab.h)
from x_multi_agent_amd import engine, synth
sc = synth.make_config(4)
eng = engine.Engine(30, 0, 400)
for _ in range(5): eng.visual_update(sc)
ts=[]
for _ in range(30):
    t0=time.perf_counter(); eng.visual_update(sc); ts.append(time.perf_counter()-t0)
print("xk_visual_update (host buffers in/out, every input staged per call): %.3f ms median" % (np.median(ts)*1e3))
eng.stage(sc)
ts=[]
for _ in range(30):
    eng.upload_P(sc["P"])
    t0=time.perf_counter(); r=eng.visual_update_staged(sc["sigma_img"]); ts.append(time.perf_counter()-t0)
print("xk_visual_update_staged (inputs resident, correction + flags back): %.3f ms median" % (np.median(ts)*1e3))
