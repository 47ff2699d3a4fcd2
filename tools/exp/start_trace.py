"""Round 6: the first panels of the single launch, every workgroup (XK_CAQR_PERSIST_DBG=1, lab build): where the pipeline fill goes.
Per panel 0..3: per XCD tile start; per first-level group the phase stamps (root rows of phase q out); per last-level workgroup rows in / done."""
import ctypes as C, os, sys
sys.path.insert(0, '.')
os.environ.setdefault("XK_CAQR_PERSIST_DBG", "1")
os.environ.setdefault("XK_PIPE_SPLIT", "3")
import numpy as np
from x_multi_agent_amd import engine, synth
sc = synth.make_config(4)
N, K, M = synth.CONFIGS[4]
eng = engine.LabEngine(N, M, K)
eng.stage(sc)
t = eng.bench_staged(sc["sigma_img"], 3, 10)
print({k: round(v["ms"], 4) for k, v in t["stages"].items() if v["ms"] > 0})
NW = 65536
out = (C.c_longlong * NW)()
eng.L.xk_debug_persist_stamps(eng.h, out, C.c_int(NW))
w = np.array(list(out), dtype=np.int64)
us = lambda x: x / 100.0
e = w[40960 + 512:40960 + 512 + 152]; t0 = e[e > 0].min()           # first tile entry
TS = w[12288:12288 + 8 * 32].reshape(8, 32)                          # [xcd][panel] tile (slot 0) panel start
MP = w[16640:16640 + 96 * 16 * 4].reshape(96, 16, 4)                 # [first-level wg][panel][phase] done
M1 = w[4096:4096 + 96 * 32].reshape(96, 32)                          # [wg][0..15 pending in | 16..31 phase-0 rows out]
LP = w[32768:32768 + 7 * 16 * 4 * 2].reshape(7, 16, 4, 2)            # [last-level wg][panel][phase][in, done]
Lw = w[8192:8192 + 7 * 64].reshape(7, 64)
seen = w[46080:46080 + 96]
f = lambda a: " ".join(f"{us(x - t0):6.1f}" if x > 0 else "   -  " for x in a)
print("all times: us after the first tile workgroup's entry")
print("first level sees panel-0 phase-0 rows (per workgroup):", f(seen))
for k in range(0, 4):
    print(f"== panel {k}")
    print(" tile start per XCD:", f(TS[:, k]))
    for q in range(4):
        m = MP[:, k, q]
        print(f"  first level phase {q} out: per XCD-group (slowest item) ", f([m[g * 6:(g + 1) * 6].max() for g in range(16)]))
    if k > 0: print("  pending in (slowest item per group)", f([M1[g * 6:(g + 1) * 6, k].max() for g in range(16)]))
    for q in range(4):
        print(f"  last level phase {q}: in ", f(LP[:, k, q, 0]), "| done", f(LP[:, k, q, 1]))
    print("  last level out:", f(Lw[:, 32 + k]))
eng.close()
